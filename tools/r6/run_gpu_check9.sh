#!/bin/bash
mkdir -p gpurun_out/r6
timeout 300 python tools/r6/attn_ab.py 64 "100:1" > gpurun_out/r6/check9_ab.txt 2>&1; cat gpurun_out/r6/check9_ab.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_model.py -x -q -m gpu -k "attention or cfg3 or group or encoder" --durations=8 > gpurun_out/r6/check9_tests.txt 2>&1; echo "rc=$?"; tail -14 gpurun_out/r6/check9_tests.txt
timeout 600 python bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r6/check9_bench_cfg3.json 2> gpurun_out/r6/check9_bench_cfg3.err; echo "rc=$?"
python -c "import json;d=json.loads(open('gpurun_out/r6/check9_bench_cfg3.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['loss_first_step'], d['loss_last_step'])"; tail -2 gpurun_out/r6/check9_bench_cfg3.err
