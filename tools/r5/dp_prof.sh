#!/bin/bash
# kernel table of the one-rank data-parallel step with live stand-in collectives (see dp_sweep.sh)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r5
for tag in standin143 standin286 nolive; do
  case $tag in
    standin143) A="--force-collectives --graph-collectives on --standin-gbps 143";;
    standin286) A="--force-collectives --graph-collectives on --standin-gbps 286";;
    nolive) A="--force-collectives --graph-collectives on";;
  esac
  rm -rf gpurun_out/r5/prof_$tag
  rocprofv3 --kernel-trace --stats -d gpurun_out/r5/prof_$tag -o run -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-diagnostics $A > gpurun_out/r5/prof_${tag}_bench.json 2> gpurun_out/r5/prof_$tag.err
  DB=$(find gpurun_out/r5/prof_$tag -name "*.db" | head -1)
  python tools/prof_summary.py $DB > gpurun_out/r5/dp_${tag}_kernel_stats.txt
  rm -rf gpurun_out/r5/prof_$tag
  echo "== $tag $(tail -1 gpurun_out/r5/prof_${tag}_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])")"; head -14 gpurun_out/r5/dp_${tag}_kernel_stats.txt | cut -c1-150
  grep -E "paced|adam" gpurun_out/r5/dp_${tag}_kernel_stats.txt | cut -c1-150
done
