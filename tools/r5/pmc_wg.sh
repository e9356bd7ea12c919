#!/bin/bash
# HBM-side read traffic of the grouped wgrad launch inside the cfg-4 step (FETCH_SIZE in KB; x2 on gfx950 for wide streaming reads)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r5/pmc_wg; rm -rf $O; mkdir -p $O
for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  n=$(echo $c | cut -c1-5)
  rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "wgrad_group" -d $O/$n -o p --output-format csv -- python $R/bench.py --no-graphs --config ${CFG:-cfg4} --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/$n.log 2>&1
  python - <<PY
import csv,glob,collections
for f in glob.glob('$O/$n/**/*counter_collection.csv', recursive=True):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(k, len(v), sum(v)/len(v))
PY
done
