#!/bin/bash
# variants of the fused MLP launch (EXP build): kernel time under rocprofv3 inside the cfg-4 / cfg-5 step, and step times
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_exp.so
for cfg in cfg4 cfg5; do
for v in ${VARS:-0 1 2}; do
  export S3D_FUSED_MLP_VARIANT=$v
  rm -rf gpurun_out/r5/prof_fm
  rocprofv3 --kernel-trace --stats -d gpurun_out/r5/prof_fm -o run -- python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  DB=$(find gpurun_out/r5/prof_fm -name "*.db" | head -1)
  echo "$cfg variant=$v $(python tools/prof_summary.py $DB | grep blk_mlp_full | cut -c1-64)"
  rm -rf gpurun_out/r5/prof_fm
  python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('    step', d['value'], d['ms_per_step'])"
done
S3D_FUSED_MLP_FULL=0 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg three launches: step', d['value'], d['ms_per_step'])"
done
