#!/bin/bash
# A/B of two builds of the library in ONE gpurun call (box-to-box variance is +-2 %): libs3d_hip_prev.so vs libs3d_hip.so, interleaved
cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_prev.so python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | line "prev cfg2"
  python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | line "new  cfg2"
done
for c in ${CONFIGS:-cfg3 cfg4 cfg5}; do
  S="--steps 30 --warmup 5"; [ $c = cfg3 ] && S="--steps 5 --warmup 2"
  S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_prev.so python bench.py --config $c $S --no-cpu-baseline --no-roofline 2>/dev/null | line "prev $c"
  python bench.py --config $c $S --no-cpu-baseline --no-roofline 2>/dev/null | line "new  $c"
done
