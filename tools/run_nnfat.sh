#!/bin/bash
# 256x256 dgrad tile (default) vs the 128x128 / 256x128 tiles (S3D_DGRAD_FAT=2, tuning build), same box
cd $GRAFT_REPO_ROOT
export S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_exp.so
for v in 2 -1 2 -1; do if [ $v = -1 ]; then unset S3D_DGRAD_FAT; else export S3D_DGRAD_FAT=$v; fi; timeout 300 python tools/nnfat_check.py 2>&1 | grep -v amdgpu.ids; done
