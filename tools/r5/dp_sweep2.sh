#!/bin/bash
# Round 5, second pass of the one-rank data-parallel sweep (see dp_sweep.sh): the remaining bucket schemes, and a stand-in so fast that no
# wire time is exposed -- what is left over the "nothing live" line is the cost of the live branches themselves.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="python bench.py --steps ${STEPS:-300} --warmup 30 --no-roofline --no-cpu-baseline --no-diagnostics"
o=gpurun_out/r5/dp_branch_tax2.txt; mkdir -p gpurun_out/r5; : > $o
run() { echo "## $1" >> $o; shift; "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print({k:d.get(k) for k in ('ms_per_step','value')}, d['config'].get('collectives'), d['config'].get('grad_buckets'), d['config'].get('grad_wire'))" >> $o 2>&1; }
S="--force-collectives --standin-gbps 143 --standin-latency-us 30"
F="--force-collectives --standin-gbps 3000"
for rep in 1 2; do
run "single graph, no collectives" $B
run "captured, uniform 3 blocks, 1-rank RCCL (nothing live)" $B --force-collectives --graph-collectives on --bucket-blocks 3
run "captured, uniform 3 blocks, stand-in, single update" $B $S --graph-collectives on --bucket-blocks 3 --single-update
run "captured, uniform 3 blocks, stand-in, sliced" $B $S --graph-collectives on --bucket-blocks 3
run "captured, uniform 4 blocks (3 buckets), stand-in, single update" $B $S --graph-collectives on --bucket-blocks 4 --single-update
run "captured, uniform 6 blocks (2 buckets), stand-in, single update" $B $S --graph-collectives on --bucket-blocks 6 --single-update
run "segment graphs, uniform 3 blocks, stand-in, single update" $B $S --graph-collectives off --bucket-blocks 3 --single-update
run "segment graphs, uniform 4 blocks, stand-in, single update" $B $S --graph-collectives off --bucket-blocks 4 --single-update
run "captured, uniform 3 blocks, FAST stand-in (3 TB/s: ~15 us per bucket), single update" $B $F --graph-collectives on --bucket-blocks 3 --single-update
run "captured, uniform 1 block, FAST stand-in, single update" $B $F --graph-collectives on --bucket-blocks 1 --single-update
run "segment graphs, uniform 3 blocks, FAST stand-in, single update" $B $F --graph-collectives off --bucket-blocks 3 --single-update
run "captured, uniform 3 blocks, stand-in 286 GB/s (bf16 wire time), single update" $B --force-collectives --standin-gbps 286 --standin-latency-us 30 --graph-collectives on --bucket-blocks 3 --single-update
done
cat $o
