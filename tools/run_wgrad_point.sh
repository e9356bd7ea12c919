#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2
for v in "X=0" "S3D_GEMM_TILE=2" "S3D_GEMM_TILE=2 S3D_GEMM_SPLITK=32" "S3D_GEMM_TILE=2 S3D_GEMM_SPLITK=64" "S3D_GEMM_TILE=1 S3D_GEMM_SPLITK=32" "S3D_GEMM_TILE=1 S3D_GEMM_SPLITK=8"; do
  echo "== $v"
  env $v ROWS=${ROWS:-32896} timeout 300 python tools/wgrad_point_bench.py 2>&1 | grep -E "wgrad|rror"
done > gpurun_out/r2/wgrad_point.txt 2>&1
cat gpurun_out/r2/wgrad_point.txt
