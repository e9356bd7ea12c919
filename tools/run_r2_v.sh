cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_model.py tests/test_gpu_trajectory.py -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -3
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('bench', d['ms_per_step'], d['value'])"
