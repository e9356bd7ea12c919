#!/bin/bash
# same-box A/B of libs3d_hip_prev.so vs libs3d_hip.so on several configurations: CFGS="cfg2 cfg3 cfg4 cfg5"
cd /root/repo; export TMPDIR=/tmp
o=gpurun_out/r4_ab_multi.txt; : > $o
for c in ${CFGS:-cfg2 cfg3 cfg4 cfg5}; do
  case $c in cfg2) st="--steps 400 --warmup 40";; cfg3) st="--steps 4 --warmup 1";; *) st="--steps 60 --warmup 10";; esac
  B="python bench.py --config $c $st --no-roofline --no-cpu-baseline"
  for r in 1 2; do
    for w in prev new; do
      if [ $w = prev ]; then export S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_prev.so; else unset S3D_LIB_PATH; fi
      echo "## $c $w" >> $o; $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])" >> $o 2>&1
    done
  done
done
cat $o
