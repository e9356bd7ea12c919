#!/bin/bash
# A/B of environment switches on ONE box: every argument is "label:VAR=val,VAR=val" (or "label:" for the defaults); three interleaved
# rounds of bench.py (cfg-2 unless CFG is set).  Example: bash tools/run_env_ab.sh base: ov3:S3D_UPDATE_OVERLAP=3
cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  for spec in "$@"; do
    label=${spec%%:*}; envs=${spec#*:}
    ( IFS=,; for kv in $envs; do export "$kv"; done; unset IFS; python bench.py ${CFG:+--config $CFG} ${STEPS} --no-cpu-baseline --no-roofline 2>/dev/null | line "$label" )
  done
done
