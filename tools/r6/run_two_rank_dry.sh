#!/bin/bash
# plain `python bench.py --gpus 2` (no launcher) on a one-GPU box: both ranks on device 0, gloo collectives -- the launch plumbing, not numbers
mkdir -p gpurun_out/r6
export S3D_BENCH_DEVICE=0 S3D_BENCH_BACKEND=gloo
for dp in sharded replicated; do
  timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --dp $dp > gpurun_out/r6/dry2_$dp.json 2> gpurun_out/r6/dry2_$dp.err; echo "rc=$?"
  tail -1 gpurun_out/r6/dry2_$dp.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$dp', d['n_gpus'], d['ms_per_step'], d['config']['dp_design'][:40], '|', d['config']['collectives'], d['loss_first_step'], d['loss_last_step'])"
  grep -v "amdgpu.ids\|Warning\|warn" gpurun_out/r6/dry2_$dp.err | tail -3
done
timeout 300 python bench.py --gpus 2 --config cfg4 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r6/dry2_cfg4.json 2> gpurun_out/r6/dry2_cfg4.err; echo "rc=$?"; tail -1 gpurun_out/r6/dry2_cfg4.json | cut -c1-200
