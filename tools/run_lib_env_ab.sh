#!/bin/bash
# A/B of two library builds x environment switches on ONE box: arguments as tools/run_env_ab.sh; every spec runs with
# libs3d_hip_prev.so ("prev") and libs3d_hip.so ("new"), three interleaved rounds.
cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  for spec in "$@"; do
    label=${spec%%:*}; envs=${spec#*:}
    ( IFS=,; for kv in $envs; do export "$kv"; done; unset IFS
      S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_prev.so python bench.py ${CFG:+--config $CFG} ${STEPS} --no-cpu-baseline --no-roofline 2>/dev/null | line "prev $label"
      python bench.py ${CFG:+--config $CFG} ${STEPS} --no-cpu-baseline --no-roofline 2>/dev/null | line "new  $label" )
  done
done
