cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q 2>&1 | grep -E "passed|failed|Error|error|rel err|rms err" | tail -4
python bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cfg3', d['ms_per_step'], d['value'])"
