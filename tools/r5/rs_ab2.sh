#!/bin/bash
# same-box A/B of the row-stream kernel (EXP build: S3D_ROWSTREAM=0 sends the shapes back to the 128 x 128 tiles)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_exp.so
run() { python bench.py --config $1 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do for cfg in cfg4 cfg5; do
  S3D_ROWSTREAM=0 run $cfg "tiles    "
  S3D_ROWSTREAM=1 run $cfg "rowstream"
done; done
