#!/bin/bash
# fused D = 192 MLP launch: waves per workgroup (S3D_FUSED_MLP_NW, EXP build) at cfg-4 and -- with the row gate lowered -- cfg-5
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
export S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_exp.so
run() { python bench.py --config $1 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['value'], d['ms_per_step'])"; }
for nw in 5 6 7 8 9; do S3D_FUSED_MLP_NW=$nw run cfg4 "nw=$nw"; done
S3D_FUSED_MLP_FULL=0 run cfg5 "three launches"
for nw in 5 6 7 8 9; do S3D_FUSED_MLP_MIN_ROWS=8192 S3D_FUSED_MLP_NW=$nw run cfg5 "fused nw=$nw"; done
