#!/bin/bash
cd $GRAFT_REPO_ROOT
export S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_exp.so
for v in -1 4 -1 4; do if [ $v = -1 ]; then unset S3D_DGRAD_FAT; else export S3D_DGRAD_FAT=$v; fi; timeout 300 python tools/nnfat_check.py 2>&1 | grep -v amdgpu.ids; done
