cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_model.py -x -q -k "cfg2_full or golden" 2>&1 | grep -v "amdgpu.ids" | tail -3
for f in 1 0; do S3D_LN_FUSE=$f python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('LN_FUSE=$f', d['ms_per_step'], d['value'], d['loss_last_step'])"; done
