#!/bin/bash
# Quick GPU check: kernel + model parity tests, then short bench lines for the four configurations (no CPU baseline / roofline).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_points.py -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5
for i in 1 2; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', d['value'], d['ms_per_step'])"; done
python bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3', d['value'], d['ms_per_step'])"
python bench.py --config cfg4 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4', d['value'], d['ms_per_step'])"
python bench.py --config cfg5 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5', d['value'], d['ms_per_step'])"
