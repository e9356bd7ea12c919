#!/bin/bash
# phase cost of blk_attn_bwd_kernel by elimination: probe libraries that return after the staging (1) / after the GEMM (2)
cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r4
for v in 1 2; do
  rm -rf gpurun_out/r4/prof_p$v
  S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_probe$v.so rocprofv3 --kernel-trace --stats -d gpurun_out/r4/prof_p$v -o run -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline > /dev/null 2> gpurun_out/r4/prof_p$v.err
  DB=$(find gpurun_out/r4/prof_p$v -name "*.db" | head -1)
  python tools/prof_summary.py $DB | grep -E "blk_attn_bwd" | cut -c1-120
  rm -rf gpurun_out/r4/prof_p$v
done
