cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_points.py tests/test_gpu_model.py tests/test_gpu_kernels.py -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -3
python bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cfg3', d['ms_per_step'], d['value'])"
python bench.py --config cfg5 --steps 30 --warmup 5 --force-collectives --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cfg5 forced-collectives pipelined', d['ms_per_step'], d['value'], d['config']['launch'])"
