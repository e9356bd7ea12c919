#!/bin/bash
# one PMC pass over the eager cfg-2 step: tools/pmc_one.sh outdir "COUNTER1 COUNTER2 ..."
OUT=$1; R=$PWD
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $2 -d $OUT/p -o p --output-format csv -- python $R/bench.py --no-graphs --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/p.log 2>&1
cd $R
