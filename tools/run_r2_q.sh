cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
for c in cfg4 cfg5; do
  python bench.py --config $c --steps 30 --warmup 5 --pipeline --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$c pipelined', d['ms_per_step'], d['value'])"
  python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$c plain', d['ms_per_step'], d['value'])"
  rocprofv3 --kernel-trace --stats -d gpurun_out/r2/prof_$c -o run -- python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
done
ls gpurun_out/r2/prof_cfg4
