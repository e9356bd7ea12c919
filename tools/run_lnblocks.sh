#!/bin/bash
cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for nb in 208 416 832 1024; do
  for c in cfg4 cfg5; do
    S3D_LN_PARTIAL_BLOCKS=$nb python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | line "nb=$nb $c"
  done
done
for nb in 208 416 768; do
    S3D_LN_PARTIAL_BLOCKS=$nb python bench.py --config cfg3 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | line "nb=$nb cfg3"
done
