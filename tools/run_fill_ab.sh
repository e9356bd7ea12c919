#!/bin/bash
# same-box A/B: optimizer update as filler workgroups inside the backward launches (S3D_ADAM_FILL=1, default) vs one launch behind it
cd /root/repo; export TMPDIR=/tmp
B="python bench.py --steps 400 --warmup 40 --no-roofline --no-cpu-baseline"
o=gpurun_out/r4_fill_ab.txt; : > $o
one() { echo "## $1" >> $o; shift; env "$@" $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print({k:d.get(k) for k in ('ms_per_step','value')})" >> $o 2>&1; }
one "fill off" S3D_ADAM_FILL=0
for c in 192 512 2048; do for g in 1 2; do one "fill on, <= $c filler workgroups per launch, $g float4 per thread" S3D_ADAM_FILL=1 S3D_FILL_BLOCKS=$c S3D_FILL_GPT=$g; done; done
one "fill off" S3D_ADAM_FILL=0
cat $o
