#!/bin/bash
cd $GRAFT_REPO_ROOT
export S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_exp.so S3D_GEMM_NT_FAT=3 CHECK_M=9000
for cp in 0 1 2 3 4; do echo "== S3D_FAT_CP=$cp"; S3D_FAT_CP=$cp timeout 300 python tools/fat_check.py 2>&1 | grep -E "TFLOP|worst"; done
