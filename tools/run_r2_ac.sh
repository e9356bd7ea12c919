cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -2
python bench.py > gpurun_out/final/bench_cfg2.json 2> /dev/null
python bench.py --config cfg3 --steps 5 --warmup 2 > gpurun_out/final/bench_cfg3.json 2> /dev/null
rocprofv3 --kernel-trace --stats -d gpurun_out/final/prof_cfg3 -o run -- python bench.py --config cfg3 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-graphs > /dev/null 2>&1
for c in cfg2 cfg3; do python -c "import json; d=json.loads(open('gpurun_out/final/bench_$c.json').read().strip().splitlines()[-1]); print('$c', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"; done
