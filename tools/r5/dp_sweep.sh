#!/bin/bash
# Round 5: what the N > 1 code path costs at ONE rank when the bucket all-reduces are LIVE kernels on a side branch (a paced copy of the
# bucket's bytes in place of the 1-rank RCCL call, which launches nothing): bucket schemes, captured step graph against segment graphs,
# Adam bucket by bucket against one update behind all buckets.  Stand-in: 143 GB/s copy + 30 us = a ring all-reduce over 8 ranks at
# 250 GB/s bus bandwidth (2 (N-1)/N S / BW + start-up); 286 GB/s = the same with bf16 on the wire.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="python bench.py --steps ${STEPS:-300} --warmup 30 --no-roofline --no-cpu-baseline --no-diagnostics"
o=gpurun_out/r5/dp_branch_tax.txt; mkdir -p gpurun_out/r5; : > $o
run() { echo "## $1" >> $o; shift; "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print({k:d.get(k) for k in ('ms_per_step','value')}, d['config'].get('collectives'), d['config'].get('grad_buckets'), d['config'].get('grad_wire'))" >> $o 2>&1; }
S="--force-collectives --standin-gbps 143 --standin-latency-us 30"
for rep in 1 2; do
run "single graph, no collectives" $B
run "captured, geometric 4 buckets, 1-rank RCCL (nothing live)" $B --force-collectives --graph-collectives on
run "captured, geometric 4 buckets, stand-in" $B $S --graph-collectives on
run "captured, geometric 4 buckets, stand-in, single update" $B $S --graph-collectives on --single-update
run "captured, uniform 3 blocks (4 buckets), stand-in" $B $S --graph-collectives on --bucket-blocks 3
run "captured, uniform 2 blocks (6 buckets), stand-in" $B $S --graph-collectives on --bucket-blocks 2
run "captured, uniform 1 block (12 buckets), stand-in" $B $S --graph-collectives on --bucket-blocks 1
run "captured, uniform 2 blocks, stand-in, single update" $B $S --graph-collectives on --bucket-blocks 2 --single-update
run "segment graphs, uniform 2 blocks, stand-in" $B $S --graph-collectives off --bucket-blocks 2
run "segment graphs, geometric 2 buckets, stand-in" $B $S --graph-collectives off
run "captured, geometric 4 buckets, stand-in 286 GB/s (bf16 wire time)" $B --force-collectives --standin-gbps 286 --standin-latency-us 30 --graph-collectives on
run "captured, uniform 2 blocks, stand-in 286 GB/s (bf16 wire time)" $B --force-collectives --standin-gbps 286 --standin-latency-us 30 --graph-collectives on --bucket-blocks 2
done
cat $o
