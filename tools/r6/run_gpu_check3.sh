#!/bin/bash
mkdir -p gpurun_out/r6
i=0
for mode in "" "--dp sharded" "--dp sharded --graph-collectives off" "--dp sharded --emulate-world 8 --standin-gbps 286 --standin-latency-us 30" "--dp sharded --emulate-world 8 --standin-gbps 286 --standin-latency-us 30 --graph-collectives off" "--dp sharded --emulate-world 8 --standin-gbps 286 --standin-latency-us 30 --bucket-list 3,3,3,3" "--dp sharded --emulate-world 8 --standin-gbps 286 --standin-latency-us 30 --bucket-list 4,4,3,1" "--dp replicated --force-collectives --standin-gbps 143 --standin-latency-us 30"; do
  i=$((i+1))
  timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline $mode > gpurun_out/r6/check3_$i.json 2> gpurun_out/r6/check3_$i.err; rc=$?
  echo "[$mode] rc=$rc $(python -c "import json;d=json.loads(open('gpurun_out/r6/check3_$i.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], '|', d['config']['collectives'], '|', d['config'].get('bucket_blocks'), d['loss_first_step'], d['loss_last_step'])" 2>&1 | tail -1)"; grep -v "amdgpu.ids\|Warning\|warn" gpurun_out/r6/check3_$i.err | tail -2
done
