cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
for t in 4 5 6 7 8; do
  echo "== tile $t tests"; S3D_GEMM_NT_TILE=$t python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm_forward_nt or epilogues" 2>&1 | tail -2
  echo "== tile $t bench"; S3D_GEMM_NT_TILE=$t python tools/gemm_bench.py 2>&1 | grep -E "fwd.*split=1|Error"
done > gpurun_out/r2/gemm_newtiles.txt 2>&1
cat gpurun_out/r2/gemm_newtiles.txt
