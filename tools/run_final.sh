#!/bin/bash
# Round-end check on a gpurun box: GPU test suite, smoke, the four bench configurations, rocprofv3 summary of the headline command.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/final/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1
python bench.py > gpurun_out/final/bench_cfg2.json 2> gpurun_out/final/bench_cfg2.err
python bench.py --config cfg3 --steps 5 --warmup 2 > gpurun_out/final/bench_cfg3.json 2> gpurun_out/final/bench_cfg3.err
python bench.py --config cfg4 --steps 30 --warmup 5 > gpurun_out/final/bench_cfg4.json 2> gpurun_out/final/bench_cfg4.err
python bench.py --config cfg5 --steps 30 --warmup 5 > gpurun_out/final/bench_cfg5.json 2> gpurun_out/final/bench_cfg5.err
rocprofv3 --kernel-trace --stats -d gpurun_out/final/prof_cfg2 -o run -- python bench.py --no-cpu-baseline > gpurun_out/final/bench_cfg2_under_rocprof.json 2> /dev/null
cat gpurun_out/final/gpu_tests.log; tail -1 gpurun_out/final/smoke.log
for c in cfg2 cfg3 cfg4 cfg5; do python -c "import json; d=json.loads(open('gpurun_out/final/bench_$c.json').read().strip().splitlines()[-1]); print('$c', d['value'], d['unit'], d['ms_per_step'], 'ms  roofline', d['roofline']['bound'], d['roofline']['frac'], ' cpu', d['cpu_baseline']['value'])"; done
