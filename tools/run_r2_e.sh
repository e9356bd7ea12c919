cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
python bench.py --config cfg4 --steps 20 --warmup 3 > gpurun_out/r2/bench_cfg4.json 2> gpurun_out/r2/bench_cfg4.err
python bench.py --config cfg5 --steps 20 --warmup 3 > gpurun_out/r2/bench_cfg5.json 2> gpurun_out/r2/bench_cfg5.err
python bench.py --config cfg3 --steps 5 --warmup 2 > gpurun_out/r2/bench_cfg3.json 2> gpurun_out/r2/bench_cfg3.err
S3D_BENCH_BACKEND=gloo S3D_BENCH_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 20 --warmup 5 --no-roofline > gpurun_out/r2/bench_2rank_dry.json 2> gpurun_out/r2/bench_2rank_dry.err
python bench.py --force-collectives --wire bf16 --steps 50 --warmup 5 --no-roofline --no-cpu-baseline > gpurun_out/r2/bench_force_bf16.json 2> gpurun_out/r2/bench_force_bf16.err
python bench.py --force-collectives --graph-collectives --steps 50 --warmup 5 --no-roofline --no-cpu-baseline > gpurun_out/r2/bench_force_graphcoll.json 2> gpurun_out/r2/bench_force_graphcoll.err
for f in cfg4 cfg5 cfg3 2rank_dry force_bf16 force_graphcoll; do echo "== $f"; tail -c 1500 gpurun_out/r2/bench_$f.json; tail -3 gpurun_out/r2/bench_$f.err; done
