#!/bin/bash
# how many CPU threads should the oracle side of the GPU suite use on this host?
for t in 8 16 32 64 999; do
  export S3D_TEST_THREADS=$t
  s=$SECONDS
  timeout 400 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "test_cfg5_full_size_parity or (test_cfg3_reduced_batch_training_step_matches_oracle and 3-0.0)" 2>&1 | grep -E "passed|failed"
  echo "threads=$t: $((SECONDS - s)) s"
done
