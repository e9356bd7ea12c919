#!/bin/bash
# Per-kernel launch durations of the cfg-2 step at per-GPU batch 8 / 16 / 32 / 64: separates the fixed (cold-start, latency) part of
# every launch from the part that scales with the work.
cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r4
for b in 8 16 32 64; do
  rm -rf gpurun_out/r4/prof_b$b
  rocprofv3 --kernel-trace --stats -d gpurun_out/r4/prof_b$b -o run -- python bench.py --batch $b --steps 100 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r4/prof_b${b}_bench.json 2> gpurun_out/r4/prof_b$b.err
  DB=$(find gpurun_out/r4/prof_b$b -name "*.db" | head -1)
  python tools/prof_summary.py $DB > gpurun_out/r4/b${b}_kernel_stats.txt
  rm -rf gpurun_out/r4/prof_b$b
  echo "== batch $b"; head -16 gpurun_out/r4/b${b}_kernel_stats.txt | cut -c1-150
done
