#!/bin/bash
cd $GRAFT_REPO_ROOT
export S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_exp.so CHECK_M=8300
for m in 188160 12608 32768; do for fat in 2 3; do echo "== M=$m FAT=$fat"; M=$m S3D_GEMM_NT_FAT=$fat timeout 300 python tools/fat_check.py 2>&1 | grep -E "TFLOP"; done; done
