#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2
S3D_WGRAD_FAT=3 S3D_DGRAD_FAT=3 python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad or dgrad" 2>&1 | grep -E "passed|failed|Error" | tail -3
for v in "0 0" "1 1" "3 3"; do
  set -- $v
  echo "== S3D_WGRAD_FAT=$1 S3D_DGRAD_FAT=$2"
  S3D_WGRAD_FAT=$1 S3D_DGRAD_FAT=$2 ROWS=${ROWS:-188160} timeout 300 python tools/wgrad_big_bench.py 2>&1 | grep -E "wgrad|dgrad|rror"
done > gpurun_out/r2/wgrad_big.txt 2>&1
cat gpurun_out/r2/wgrad_big.txt
