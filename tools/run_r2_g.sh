cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
export S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_tl.so
for t in 4 7 5; do ONLY=fc1 S3D_GEMM_NT_TILE=$t python tools/timeline_probe.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r2/tl_fc1_fat.txt
ONLY=fc2 S3D_GEMM_NT_TILE=6 python tools/timeline_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2/tl_fc1_fat.txt
cut -c1-400 gpurun_out/r2/tl_fc1_fat.txt
