#!/bin/bash
# same-box A/B: product library vs the tuning build (libs3d_hip_exp.so) with optional env knobs per line of $KNOBS (';'-separated)
cd /root/repo; export TMPDIR=/tmp
B="python bench.py ${CFG:+--config $CFG} ${STEPS:---steps 400 --warmup 40} --no-roofline --no-cpu-baseline"
o=gpurun_out/r4_exp_ab.txt; : > $o
one() { echo "## $*" >> $o; env "$@" $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print({k:d.get(k) for k in ('ms_per_step','value')})" >> $o 2>&1; }
IFS=';' read -ra KS <<< "${KNOBS:-S3D_DUMMY=1}"
for r in 1 2 3; do
  one S3D_DUMMY=1
  for k in "${KS[@]}"; do one S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_exp.so $k; done
done
cat $o
