#!/bin/bash
# forward-GEMM tile variants on the cfg-3 shapes (one process per tile id: the override is read once).
# Needs the tuning library:  make -C simple3d-former_amd/csrc EXP=1  (tile ids 4 .. 28 are compiled out of the product build)
export S3D_LIB_PATH=$GRAFT_REPO_ROOT/simple3d-former_amd/libs3d_hip_exp.so
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
for t in ${TILES:--1 11 16 14 7 17 4}; do
  echo "== S3D_GEMM_NT_TILE=$t"
  S3D_GEMM_NT_TILE=$t M=${M:-65536} timeout 300 python tools/gemm_big_bench.py 2>&1 | grep -E "split=1|Error|error" 
done > gpurun_out/r2/gemm_tiles_big.txt 2>&1
cat gpurun_out/r2/gemm_tiles_big.txt
