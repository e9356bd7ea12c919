#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
B="python bench.py --steps 400 --warmup 40 --no-roofline --no-cpu-baseline"
o=gpurun_out/r4_ln_ab.txt; : > $o
one() { echo "## $1" >> $o; shift; env "$@" $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print({k:d.get(k) for k in ('ms_per_step','value')})" >> $o 2>&1; }
for r in 1 2; do for n in 208 416 104 312; do one "LayerNorm backward on $n workgroups" S3D_LN_PARTIAL_BLOCKS=$n; done; done
cat $o
