#!/bin/bash
mkdir -p gpurun_out/r6
i=0
run() {
  i=$((i+1))
  timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline $1 > gpurun_out/r6/check4_$i.json 2> gpurun_out/r6/check4_$i.err; rc=$?
  echo "[$1] rc=$rc $(python -c "import json;d=json.loads(open('gpurun_out/r6/check4_$i.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], '|', d['config']['collectives'])" 2>&1 | tail -1)"
}
E="--emulate-world 8 --standin-gbps 286 --standin-latency-us 30"
for bl in 12 11,1 6,5,1 4,4,4 4,4,3,1 3,3,3,2,1; do
  run "--dp sharded --bucket-list $bl"
  run "--dp sharded --bucket-list $bl --graph-collectives off"
  run "--dp sharded --bucket-list $bl $E"
  run "--dp sharded --bucket-list $bl $E --graph-collectives off"
done
