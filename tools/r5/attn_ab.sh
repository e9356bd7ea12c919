#!/bin/bash
# coop attention kernels inside the cfg-4 / cfg-5 step: per-kernel time under rocprofv3 + step times; attention parity tests first
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_gpu_kernels.py -k "attention or attn" -x -q > gpurun_out/r5/attn_tests.log 2>&1; tail -2 gpurun_out/r5/attn_tests.log
for cfg in cfg4 cfg5 ${EXTRA_CFG}; do
  python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'])"
  rm -rf gpurun_out/r5/prof_at
  rocprofv3 --kernel-trace --stats -d gpurun_out/r5/prof_at -o run -- python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  DB=$(find gpurun_out/r5/prof_at -name "*.db" | head -1)
  python tools/prof_summary.py $DB | grep "attn_" | cut -c1-110
  rm -rf gpurun_out/r5/prof_at
done
