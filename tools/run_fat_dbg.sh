#!/bin/bash
# the 256x256 forward tile with one k-loop ingredient removed (tuning build; S3D_FAT_DBG)
cd $GRAFT_REPO_ROOT
export S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_exp.so S3D_GEMM_NT_FAT=3 CHECK_M=21700 M=${M:-188160}
for d in ${DBGS:-0 1 2 3}; do echo "== S3D_FAT_DBG=$d"; S3D_FAT_DBG=$d timeout 300 python tools/fat_check.py 2>&1 | grep -E "TFLOP|check|worst"; done
