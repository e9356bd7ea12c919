cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
for t in 2 1 0; do echo "== S3D_GEMM_NT_TILE=$t"; ONLY= S3D_GEMM_NT_TILE=$t python tools/gemm_bench.py 2>&1 | grep -E "fwd|Error|error"; done > gpurun_out/r2/gemm_tiles.txt
cat gpurun_out/r2/gemm_tiles.txt
