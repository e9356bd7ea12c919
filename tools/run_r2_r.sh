cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
python bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2/bench_cfg3_b.json 2> gpurun_out/r2/bench_cfg3_b.err
python -c "import sys,json; d=json.loads(open('gpurun_out/r2/bench_cfg3_b.json').read().strip().splitlines()[-1]); print('cfg3', d['ms_per_step'], d['value'])"
S3D_CLS_ONLY=0 python bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cfg3 dense last block', d['ms_per_step'], d['value'])"
rocprofv3 --kernel-trace --stats -d gpurun_out/r2/prof_cfg3 -o run -- python bench.py --config cfg3 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-graphs > /dev/null 2>&1
ls gpurun_out/r2/prof_cfg3
