#!/bin/bash
# A/B of an environment switch in ONE gpurun call:  tools/run_ab_env.sh "S3D_GEMM_COL_SUMS=0" cfg4 cfg5
cd $GRAFT_REPO_ROOT
V=$1; shift
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for c in "$@"; do
  S="--steps 30 --warmup 5"; [ $c = cfg3 ] && S="--steps 5 --warmup 2"; [ $c = cfg2 ] && S=""
  env $V python bench.py --config $c $S --no-cpu-baseline --no-roofline 2>/dev/null | line "[$V] $c"
  python bench.py --config $c $S --no-cpu-baseline --no-roofline 2>/dev/null | line "[default] $c"
done
done
