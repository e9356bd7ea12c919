#!/bin/bash
# same-box A/B: fused attention backward (S3D_FUSED_BWD=1, default) vs the seven-launch backward
cd /root/repo; export TMPDIR=/tmp
B="python bench.py --steps 400 --warmup 40 --no-roofline --no-cpu-baseline"
o=gpurun_out/r4_bwd_ab.txt; : > $o
one() { echo "## $1" >> $o; shift; env "$@" $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print({k:d.get(k) for k in ('ms_per_step','value')})" >> $o 2>&1; }
for r in 1 2; do one "seven-launch backward" S3D_FUSED_BWD=0; one "fused attention backward" S3D_FUSED_BWD=1; done
cat $o
