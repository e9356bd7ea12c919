cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_model.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
for f in 1 0; do S3D_FUSE_LOSS_END=$f python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('FUSE_LOSS_END=$f', d['ms_per_step'], d['value'])"; done
