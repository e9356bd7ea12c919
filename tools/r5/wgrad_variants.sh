#!/bin/bash
# Round 5: ring depth / k-tile variants of the grouped wgrad (EXP build), per-kernel times under rocprofv3 + step times
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_exp.so
mkdir -p gpurun_out/r5
for v in ${VARIANTS:-0 1 2 3 4}; do
  for g in ${GROUPS_:-3 6}; do
    export S3D_WGRAD_VARIANT=$v S3D_WGRAD_GROUP=$g
    tag=wg_v${v}_g${g}
    rm -rf gpurun_out/r5/prof_$tag
    rocprofv3 --kernel-trace --stats -d gpurun_out/r5/prof_$tag -o run -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r5/prof_${tag}_bench.json 2> gpurun_out/r5/prof_$tag.err
    DB=$(find gpurun_out/r5/prof_$tag -name "*.db" | head -1)
    python tools/prof_summary.py $DB > gpurun_out/r5/${tag}_kernel_stats.txt
    rm -rf gpurun_out/r5/prof_$tag
    echo "== variant $v group $g: $(grep wgrad_group_kernel gpurun_out/r5/${tag}_kernel_stats.txt | cut -c1-90)"
    python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   step', d['value'], d['ms_per_step'])"
  done
done
unset S3D_WGRAD_VARIANT
S3D_WGRAD_GROUP=0 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair step', d['value'], d['ms_per_step'])"
