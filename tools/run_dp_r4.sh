#!/bin/bash
# round 4: one-rank cost of the data-parallel launch structures (forced collectives on a 1-rank RCCL group) + 2-rank gloo dry run
cd /root/repo; export TMPDIR=/tmp
B="python bench.py --steps 300 --warmup 30 --no-roofline --no-cpu-baseline --no-diagnostics"
o=gpurun_out/r4_dp_sweep.txt; : > $o
run() { echo "## $1" >> $o; shift; "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print({k:d.get(k) for k in ('ms_per_step','value')}, d['config'].get('collectives'), d['config'].get('grad_buckets'), d['config'].get('grad_wire'), d.get('graph_collectives_preflight'))" >> $o 2>&1; }
run "single graph, no collectives" $B
run "default at N>1 (auto): forced collectives" $B --force-collectives
run "segment graphs, 2 buckets fp32" $B --force-collectives --graph-collectives off
run "segment graphs, 4 buckets bf16 (round-3 default)" $B --force-collectives --graph-collectives off --buckets 4 --wire bf16
run "captured, 4 buckets fp32" $B --force-collectives --graph-collectives on --buckets 4
run "single graph again" $B
echo "## 2 ranks on one GPU, gloo (dry run of the driver's launch line; numbers mean nothing)" >> $o
S3D_BENCH_BACKEND=gloo S3D_BENCH_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 20 --warmup 5 --no-roofline --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1500 >> $o
cat $o
