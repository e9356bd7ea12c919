#!/bin/bash
# point path: the blocks' wgrads as grouped full-K launches (S3D_POINT_WGRAD_GROUP) against the paired launches; parity first
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests/test_gpu_points.py -x -q > gpurun_out/r5/pt_tests.log 2>&1; tail -4 gpurun_out/r5/pt_tests.log
run() { python bench.py --config $1 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['value'], d['ms_per_step'])"; }
for cfg in cfg4 cfg5; do
  for g in ${GROUPS_:-0 6 12}; do S3D_POINT_WGRAD_GROUP=$g run $cfg "group$g"; done
done
rocprofv3 --kernel-trace --stats -d gpurun_out/r5/prof_pt -o run -- python bench.py --config cfg4 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
DB=$(find gpurun_out/r5/prof_pt -name "*.db" | head -1)
python tools/prof_summary.py $DB | head -24 > gpurun_out/r5/pt_chain_cfg4_stats.txt
rm -rf gpurun_out/r5/prof_pt
grep wgrad_group gpurun_out/r5/pt_chain_cfg4_stats.txt
