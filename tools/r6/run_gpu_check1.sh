#!/bin/bash
# round-6 check 1: new trajectory tests, a quick slice of the suite (arena alignment touches everything), bench lines
mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_trajectory.py -x -q -m gpu -s --durations=10 > gpurun_out/r6/check1_trajectory.txt 2>&1
tail -5 gpurun_out/r6/check1_trajectory.txt
python -m pytest tests/test_gpu_model.py tests/test_gpu_dp_two_ranks.py -x -q -m gpu --durations=10 > gpurun_out/r6/check1_model.txt 2>&1
tail -5 gpurun_out/r6/check1_model.txt
python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r6/check1_bench_cfg2.json 2> gpurun_out/r6/check1_bench_cfg2.err
tail -c 600 gpurun_out/r6/check1_bench_cfg2.json; tail -3 gpurun_out/r6/check1_bench_cfg2.err
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --backward split > gpurun_out/r6/check1_bench_cfg2_split.json 2> gpurun_out/r6/check1_bench_cfg2_split.err
head -c 400 gpurun_out/r6/check1_bench_cfg2_split.json; tail -3 gpurun_out/r6/check1_bench_cfg2_split.err
for mode in "--dp sharded" "--dp sharded --emulate-world 8 --standin-gbps 286 --standin-latency-us 30" "--dp sharded --emulate-world 8 --standin-gbps 286 --standin-latency-us 30 --graph-collectives off"; do
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline $mode > gpurun_out/r6/check1_tmp.json 2> gpurun_out/r6/check1_tmp.err
  echo "$mode: $(python -c "import json;d=json.loads(open('gpurun_out/r6/check1_tmp.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['config']['collectives'])" 2>&1 | tail -1)"; tail -2 gpurun_out/r6/check1_tmp.err
done
