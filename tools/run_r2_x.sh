cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_points.py tests/test_gpu_model.py -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -4
for c in cfg5 cfg4; do python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$c', d['ms_per_step'], d['value'])"; done
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cfg2', d['ms_per_step'], d['value'])"
