#!/bin/bash
# in-kernel timeline of the 256x256 (FAT=3) and 128x256 (FAT=2) forward tiles at cfg-3 row counts (TL build)
cd $GRAFT_REPO_ROOT
export S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_tl.so TL_M=${TL_M:-94080} TL_D=768 TL_BK=32 S3D_GEMM_NT_TILE=2 ONLY=${ONLY:-qkv,fc1}
for fat in ${FATS:-3 2}; do echo "=== S3D_GEMM_NT_FAT=$fat"; S3D_GEMM_NT_FAT=$fat timeout 300 python tools/timeline_probe.py 2>&1 | grep -v amdgpu.ids; done
