#!/bin/bash
# same-box A/B of tuning knobs on the EXP build: exp_ab.sh label:VAR=val,VAR=val ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_exp.so
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in $(seq ${ROUNDS:-2}); do
  for spec in "$@"; do
    label=${spec%%:*}; envs=${spec#*:}
    ( IFS=,; for kv in $envs; do export "$kv"; done; unset IFS; python bench.py ${CFG:+--config $CFG} ${STEPS} --no-cpu-baseline --no-roofline 2>/dev/null | line "$label" )
  done
done
