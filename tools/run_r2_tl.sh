set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
export S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_tl.so
python tools/timeline_probe.py > gpurun_out/r2/tl_default.txt 2>&1
for t in 0 1 3; do S3D_GEMM_NT_TILE=$t python tools/timeline_probe.py > gpurun_out/r2/tl_tile$t.txt 2>&1; done
unset S3D_LIB_PATH
python tools/gemm_bench.py > gpurun_out/r2/gemm_bench0.txt 2>&1
python bench.py --steps 100 --warmup 10 > gpurun_out/r2/bench0.json 2> gpurun_out/r2/bench0.err
tail -3 gpurun_out/r2/tl_default.txt
