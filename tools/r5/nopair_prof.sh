#!/bin/bash
# Round 5 experiment 1: what do the cfg-2 dgrads and wgrads cost as launches of their own?  (EXP build, S3D_GEMM_NOPAIR=1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r5
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | line "product cfg2"; done
export S3D_LIB_PATH=$PWD/simple3d-former_amd/libs3d_hip_exp.so
python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | line "exp cfg2"
S3D_GEMM_NOPAIR=1 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | line "exp nopair cfg2"
for tag in pair nopair; do
  [ $tag = nopair ] && export S3D_GEMM_NOPAIR=1
  rm -rf gpurun_out/r5/prof_$tag
  rocprofv3 --kernel-trace --stats -d gpurun_out/r5/prof_$tag -o run -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline > gpurun_out/r5/prof_${tag}_bench.json 2> gpurun_out/r5/prof_$tag.err
  DB=$(find gpurun_out/r5/prof_$tag -name "*.db" | head -1)
  python tools/prof_summary.py $DB > gpurun_out/r5/${tag}_kernel_stats.txt
  rm -rf gpurun_out/r5/prof_$tag
  echo "== $tag"; head -22 gpurun_out/r5/${tag}_kernel_stats.txt | cut -c1-170
done
