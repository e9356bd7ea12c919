cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_fullsize.py -x -q -s 2>&1 | grep -v "amdgpu.ids" | tail -40 > gpurun_out/r2/test_new.log
cat gpurun_out/r2/test_new.log
