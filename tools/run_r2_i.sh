cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -x -q 2>&1 | grep -v "amdgpu.ids" | tail -15 > gpurun_out/r2/test_ln.log
cat gpurun_out/r2/test_ln.log
for f in 1 0; do S3D_LN_FUSE=$f python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('LN_FUSE=$f', d['ms_per_step'], d['value'], d['loss_last_step'])"; done
