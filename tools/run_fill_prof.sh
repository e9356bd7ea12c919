#!/bin/bash
# per-kernel durations with and without the optimizer shares riding on the backward launches (rocprofv3 --kernel-trace --stats) + timed A/B
cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r4
for f in 0 1; do
  rm -rf gpurun_out/r4/prof_fill$f
  S3D_ADAM_FILL=$f rocprofv3 --kernel-trace --stats -d gpurun_out/r4/prof_fill$f -o run -- python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-roofline > gpurun_out/r4/prof_fill${f}_bench.json 2> gpurun_out/r4/prof_fill$f.err
  DB=$(find gpurun_out/r4/prof_fill$f -name "*.db" | head -1)
  python tools/prof_summary.py $DB > gpurun_out/r4/fill${f}_kernel_stats.txt
  find gpurun_out/r4/prof_fill$f -type f ! -name "*.txt" -delete
  echo "== S3D_ADAM_FILL=$f"; head -16 gpurun_out/r4/fill${f}_kernel_stats.txt | cut -c1-150
done
tools/run_fill_ab.sh
