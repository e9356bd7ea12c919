#!/bin/bash
# stored 1-bit attention dropout mask (default) vs three hash evaluations (S3D_NO_ATTN_MASK=1): cfg-3 step, same box
cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['loss_last_step'])"; }
for i in 1 2 3; do
  S3D_NO_ATTN_MASK=1 python bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | line "hash x3"
  python bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | line "stored "
done
