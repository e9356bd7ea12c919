cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -6
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('bench', d['ms_per_step'], d['value'])"
python tools/gemm_bench.py 2>&1 | grep -E "fwd.*split=1|dgrad fc2"
