#!/bin/bash
# nine waves per workgroup for the point path's attention kernels (EXP build: S3D_ATTN_NINE bit 0 forward, 1 dQ, 2 dK/dV): parity, kernel times, steps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
export S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_exp.so
S3D_ATTN_NINE=7 timeout 900 python -m pytest tests/test_gpu_kernels.py -k "attention_fwd_bwd" -x -q 2>&1 | tail -2
for m in 0 1 2 4 7; do
 for cfg in cfg4 cfg5; do
  echo "mask=$m $cfg $(S3D_ATTN_NINE=$m python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
 done
done
for cfg in cfg4 cfg5; do
  rm -rf gpurun_out/r5/prof_at
  S3D_ATTN_NINE=7 rocprofv3 --kernel-trace --stats -d gpurun_out/r5/prof_at -o run -- python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  DB=$(find gpurun_out/r5/prof_at -name "*.db" | head -1)
  python tools/prof_summary.py $DB | grep "attn_" | cut -c1-110
  rm -rf gpurun_out/r5/prof_at
done
