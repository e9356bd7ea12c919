cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_model.py tests/test_gpu_trajectory.py tests/test_gpu_dp_two_ranks.py -x -q 2>&1 | grep -v "amdgpu.ids" | tail -6
for f in 1 0; do S3D_CLS_ONLY=$f python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('CLS_ONLY=$f', d['ms_per_step'], d['value'])"; done
