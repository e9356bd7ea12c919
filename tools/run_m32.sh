#!/bin/bash
cd $GRAFT_REPO_ROOT
# needs the tuning library: make -C simple3d-former_amd/csrc EXP=1
export S3D_LIB_PATH=$GRAFT_REPO_ROOT/simple3d-former_amd/libs3d_hip_exp.so
S3D_GEMM_M32=1 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
for m in 0 1; do echo "== S3D_GEMM_M32=$m"; S3D_GEMM_M32=$m M=65536 python tools/gemm_big_bench.py 2>&1 | grep "split=1"; done
