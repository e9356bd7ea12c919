cd $GRAFT_REPO_ROOT
S3D_ATTN_COOP_WAVES=8 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" 2>&1 | grep -E "passed|failed|Error|error" | tail -3
for w in 4 8; do
S3D_ATTN_COOP_WAVES=$w python bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cfg3 waves=$w', d['ms_per_step'], d['value'])"
S3D_ATTN_COOP_WAVES=$w python bench.py --config cfg5 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cfg5 waves=$w', d['ms_per_step'], d['value'])"
S3D_ATTN_COOP_WAVES=$w python bench.py --config cfg4 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cfg4 waves=$w', d['ms_per_step'], d['value'])"
done
