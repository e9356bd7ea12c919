#!/bin/bash
# whole-row dgrad + LayerNorm backward (192-wide layers): parity, then the cfg-4 / cfg-5 step and kernel times
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_gpu_kernels.py -k "whole_row" -x -q > gpurun_out/r5/lnrows_tests.log 2>&1; tail -3 gpurun_out/r5/lnrows_tests.log

for cfg in cfg4 cfg5; do
  python bench.py --config $cfg --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'])"
  rm -rf gpurun_out/r5/prof_ln
  rocprofv3 --kernel-trace --stats -d gpurun_out/r5/prof_ln -o run -- python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  DB=$(find gpurun_out/r5/prof_ln -name "*.db" | head -1)
  python tools/prof_summary.py $DB | grep "lnrows\|ln_bwd_kernel\|gemm_dmat_kernel<false, true, 4" | cut -c1-120
  rm -rf gpurun_out/r5/prof_ln
done
