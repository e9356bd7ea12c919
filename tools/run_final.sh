#!/bin/bash
# Round-end check on a gpurun box: GPU test suite, smoke, the four bench configurations, rocprofv3 summaries, PMC passes of the headline step.
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; rm -rf $O; mkdir -p $O gpurun_out/r3
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -4 > $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python bench.py --config cfg3 --steps 5 --warmup 2 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --config cfg4 --steps 30 --warmup 5 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python bench.py --config cfg5 --steps 30 --warmup 5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
# headline command under rocprofv3 (kernel trace + stats)
rocprofv3 --kernel-trace --stats -d $O/prof_cfg2 -o run -- python bench.py --no-cpu-baseline > $O/bench_cfg2_under_rocprof.json 2> /dev/null
python tools/prof_summary.py $(find $O/prof_cfg2 -name "*.db" | head -1) > $O/bench_kernel_stats.txt
find $O/prof_cfg2 -type f -delete
for c in cfg3 cfg4 cfg5; do
  S="--steps 20 --warmup 5"; [ $c = cfg3 ] && S="--steps 3 --warmup 1"
  bash tools/run_prof_cfg.sh $c "$S" > /dev/null 2>&1
  cp gpurun_out/r3/prof_${c}_summary.txt $O/${c}_kernel_stats.txt
done
# PMC passes (separate counter groups, eager launches) -> HBM-side traffic per kernel
bash tools/pmc_step.sh $GRAFT_REPO_ROOT/$O/pmcstep > /dev/null 2>&1
S3D_HEAD=$S3D_HEAD python tools/pmc_step_summary.py $O/pmcstep $O/pmc_step_traffic $O/bench_kernel_stats.txt > /dev/null 2>&1
find $O/pmcstep -name "*.csv" -size +20M -delete
cat $O/gpu_tests.log; tail -1 $O/smoke.log
for c in cfg2 cfg3 cfg4 cfg5; do python -c "import json; d=json.loads(open('$O/bench_$c.json').read().strip().splitlines()[-1]); print('$c', d['value'], d['unit'], d['ms_per_step'], 'ms  roofline', d['roofline']['bound'], d['roofline']['frac'], d['roofline'].get('traffic'), ' cpu', d['cpu_baseline']['value'])"; done
