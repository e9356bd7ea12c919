#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
B="python bench.py --config cfg3 --steps 4 --warmup 1 --no-roofline --no-cpu-baseline"
for r in 1 2; do
  echo "## dropout 0.1 (default)"; $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
  echo "## no dropout"; S3D_BENCH_NO_DROPOUT=1 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done
