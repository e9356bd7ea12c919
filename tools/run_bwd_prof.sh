#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r4
for f in 1; do
  rm -rf gpurun_out/r4/prof_bwd$f
  S3D_FUSED_BWD=$f rocprofv3 --kernel-trace --stats -d gpurun_out/r4/prof_bwd$f -o run -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline > gpurun_out/r4/prof_bwd${f}_bench.json 2> gpurun_out/r4/prof_bwd$f.err
  DB=$(find gpurun_out/r4/prof_bwd$f -name "*.db" | head -1)
  python tools/prof_summary.py $DB > gpurun_out/r4/bwd${f}_kernel_stats.txt
  rm -rf gpurun_out/r4/prof_bwd$f
  echo "== S3D_FUSED_BWD=$f"; head -14 gpurun_out/r4/bwd${f}_kernel_stats.txt | cut -c1-150
done
