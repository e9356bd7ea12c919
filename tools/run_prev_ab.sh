#!/bin/bash
# same-box A/B of two library builds: libs3d_hip_prev.so vs libs3d_hip.so (cfg-2 headline; CFG / STEPS override)
cd /root/repo; export TMPDIR=/tmp
B="python bench.py ${CFG:+--config $CFG} ${STEPS:---steps 400 --warmup 40} --no-roofline --no-cpu-baseline"
o=gpurun_out/r4_prev_ab.txt; : > $o
one() { echo "## $1" >> $o; shift; env "$@" $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print({k:d.get(k) for k in ('ms_per_step','value')})" >> $o 2>&1; }
for r in 1 2 3; do one prev S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_prev.so; one new S3D_DUMMY=1; done
cat $o
