cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
for ns in 2 3 4; do S3D_DMA_NS32=$ns python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('NS32=$ns', d['ms_per_step'], d['value'])"; done > gpurun_out/r2/ns32.txt
cat gpurun_out/r2/ns32.txt
