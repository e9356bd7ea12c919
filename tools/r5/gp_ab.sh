#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_points.py -x -q 2>&1 | tail -2
for cfg in cfg4 cfg5; do
  python bench.py --config $cfg --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'])"
  rm -rf gpurun_out/r5/prof_g
  rocprofv3 --kernel-trace --stats -d gpurun_out/r5/prof_g -o run -- python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  DB=$(find gpurun_out/r5/prof_g -name "*.db" | head -1)
  python tools/prof_summary.py $DB | grep "group_proj" | cut -c1-110
  rm -rf gpurun_out/r5/prof_g
done
