#!/bin/bash
# ablation of the fused MLP launch (EXP build, S3D_FM_DBG bits): kernel time under rocprofv3 inside the cfg-4 / cfg-5 step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_exp.so
for cfg in ${CFGS:-cfg4}; do
for d in ${DBGS:-0 1 2 4 8 6 14 15}; do
  export S3D_FM_DBG=$d
  rm -rf gpurun_out/r5/prof_fm
  rocprofv3 --kernel-trace --stats -d gpurun_out/r5/prof_fm -o run -- python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  DB=$(find gpurun_out/r5/prof_fm -name "*.db" | head -1)
  echo "$cfg dbg=$d $(python tools/prof_summary.py $DB | grep blk_mlp_full | cut -c1-60)"
  rm -rf gpurun_out/r5/prof_fm
done; done
