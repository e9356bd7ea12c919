#!/bin/bash
# grouped wgrad: 192-wide last tiles and two k-slices per tile; parity, then same-box A/B (EXP build knobs)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_gpu_kernels.py -k "wgrad_group" -x -q > gpurun_out/r5/wide_tests.log 2>&1; tail -3 gpurun_out/r5/wide_tests.log
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 product', d['value'], d['ms_per_step'])"
export S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_exp.so
run() { python bench.py --config $1 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do for cfg in cfg4 cfg5; do
  S3D_WGRAD_WIDE=0 S3D_WGRAD_KSPLIT=1 run $cfg "128x128, 1 slice           "
  S3D_WGRAD_WIDE=1 S3D_WGRAD_KSPLIT=1 run $cfg "wide,    1 slice,  ring 4  "
  S3D_WGRAD_WIDE=1 S3D_WGRAD_KSPLIT=2 S3D_WGRAD_VARIANT=6 run $cfg "wide,    2 slices, ring 3  "
  S3D_WGRAD_WIDE=1 S3D_WGRAD_KSPLIT=1 S3D_WGRAD_VARIANT=6 run $cfg "wide,    1 slice,  ring 3  "
  S3D_WGRAD_WIDE=1 S3D_WGRAD_KSPLIT=1 S3D_WGRAD_VARIANT=5 run $cfg "wide,    1 slice,  4 x 64  "
done; done
