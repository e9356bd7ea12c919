#!/bin/bash
# Round-6 final measurements on a gpurun box: smoke, the four bench configurations, rocprofv3 summaries, PMC passes of the headline step
# (the GPU test suite runs in its own call: gpurun_out/r6 full_suite.log)
cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r6; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python bench.py --config cfg3 --steps 5 --warmup 2 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --config cfg4 --steps 30 --warmup 5 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python bench.py --config cfg5 --steps 30 --warmup 5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
prof() {   # name, bench args
  rm -rf $O/prof_$1
  rocprofv3 --kernel-trace --stats -d $O/prof_$1 -o run -- python bench.py $2 --no-cpu-baseline > $O/bench_$1_under_rocprof.json 2> /dev/null
  python tools/prof_summary.py $(find $O/prof_$1 -name "*.db" | head -1) > $O/$1_kernel_stats.txt
  rm -rf $O/prof_$1
}
prof cfg2 ""
prof cfg3 "--config cfg3 --steps 3 --warmup 1 --no-roofline"
prof cfg4 "--config cfg4 --steps 20 --warmup 5 --no-roofline"
prof cfg5 "--config cfg5 --steps 20 --warmup 5 --no-roofline"
# PMC passes (separate counter groups, eager launches) -> HBM-side traffic per kernel of the cfg-2 step
bash tools/pmc_step.sh $GRAFT_REPO_ROOT/$O/pmcstep > /dev/null 2>&1
S3D_HEAD=$S3D_HEAD python tools/pmc_step_summary.py $O/pmcstep $O/pmc_step_traffic $O/cfg2_kernel_stats.txt > /dev/null 2>&1
rm -rf $O/pmcstep
tail -1 $O/smoke.log
for c in cfg2 cfg3 cfg4 cfg5; do python -c "import json; d=json.loads(open('$O/bench_$c.json').read().strip().splitlines()[-1]); print('$c', d['value'], d['unit'], d['ms_per_step'], 'ms  roofline', d['roofline']['bound'], d['roofline']['frac'], d['roofline'].get('traffic'), ' cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('cores'))"; done
