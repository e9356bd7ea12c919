#!/bin/bash
# 256x256 forward tile vs the product 128x256 one (tuning build), same box
cd $GRAFT_REPO_ROOT
export S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_exp.so
for fat in 2 3 ${EXTRA_FAT}; do S3D_GEMM_NT_FAT=$fat timeout 300 python tools/fat_check.py 2>&1 | grep -v amdgpu.ids; done
