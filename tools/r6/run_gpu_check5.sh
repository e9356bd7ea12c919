#!/bin/bash
mkdir -p gpurun_out/r6
i=0
run() {
  i=$((i+1))
  timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline $1 > gpurun_out/r6/check5_$i.json 2> gpurun_out/r6/check5_$i.err; rc=$?
  echo "[$1] rc=$rc $(python -c "import json;d=json.loads(open('gpurun_out/r6/check5_$i.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], '|', d['config']['collectives'])" 2>&1 | tail -1)"
}
for bl in 4,4,3,1 4,4,4 12; do
  run "--dp sharded --bucket-list $bl --emulate-world 8 --standin-gbps 100000 --standin-latency-us 0"
  run "--dp sharded --bucket-list $bl --emulate-world 8 --standin-gbps 286 --standin-latency-us 0"
  run "--dp sharded --bucket-list $bl --emulate-world 8 --standin-gbps 600 --standin-latency-us 15"
done
