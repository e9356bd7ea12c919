#!/bin/bash
mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_dp_two_ranks.py -x -q -m gpu -k "sharded" --durations=8 > gpurun_out/r6/check7.txt 2>&1; echo "rc=$?"; tail -25 gpurun_out/r6/check7.txt
