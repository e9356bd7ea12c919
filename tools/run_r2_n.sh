cd $GRAFT_REPO_ROOT
for t in 20 21 22 23; do
  echo "== tile $t"; S3D_GEMM_NT_TILE=$t python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm_forward_nt or epilogues" 2>&1 | tail -1
  S3D_GEMM_NT_TILE=$t python tools/gemm_bench.py 2>&1 | grep -E "fwd.*split=1|Error"
done
