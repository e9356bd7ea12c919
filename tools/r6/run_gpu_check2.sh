#!/bin/bash
mkdir -p gpurun_out/r6
timeout 400 python -m pytest tests/test_gpu_dp_two_ranks.py -x -q -m gpu --durations=5 > gpurun_out/r6/check2_dp.txt 2>&1; echo "dp tests rc=$?"; tail -4 gpurun_out/r6/check2_dp.txt
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r6/check2_bench_cfg2.json 2> gpurun_out/r6/check2_bench_cfg2.err; echo "bench rc=$?"
tail -c 300 gpurun_out/r6/check2_bench_cfg2.json; tail -3 gpurun_out/r6/check2_bench_cfg2.err
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --backward split > gpurun_out/r6/check2_bench_cfg2_split.json 2> gpurun_out/r6/check2_bench_cfg2_split.err; echo "bench split rc=$?"
head -c 300 gpurun_out/r6/check2_bench_cfg2_split.json; tail -3 gpurun_out/r6/check2_bench_cfg2_split.err
i=0
for mode in "--dp sharded" "--dp sharded --emulate-world 8 --standin-gbps 286 --standin-latency-us 30" "--dp sharded --emulate-world 8 --standin-gbps 286 --standin-latency-us 30 --graph-collectives off"; do
  i=$((i+1))
  timeout 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline $mode > gpurun_out/r6/check2_sharded_$i.json 2> gpurun_out/r6/check2_sharded_$i.err; echo "rc=$?"
  echo "$mode: $(python -c "import json;d=json.loads(open('gpurun_out/r6/check2_sharded_$i.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['config']['collectives'], d['loss_first_step'], d['loss_last_step'])" 2>&1 | tail -1)"; tail -2 gpurun_out/r6/check2_sharded_$i.err
done
