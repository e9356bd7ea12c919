cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_points.py tests/test_gpu_fullsize.py -x -q -k "not cfg3" 2>&1 | grep -E "passed|failed|Error|error" | tail -3
for c in cfg4 cfg5; do python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$c', d['ms_per_step'], d['value'], r['kernel'][:50], r['achieved'], r['frac'])"; done
