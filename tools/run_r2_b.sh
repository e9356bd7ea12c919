cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
tools/probes/dma_bw_probe > gpurun_out/r2/dma_bw.txt 2>&1
python -m pytest tests/test_gpu_points.py -x -q > gpurun_out/r2/test_points.log 2>&1
tail -5 gpurun_out/r2/test_points.log
tail -3 gpurun_out/r2/dma_bw.txt
