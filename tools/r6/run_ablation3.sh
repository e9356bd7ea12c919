#!/bin/bash
# which ingredient of the shipped (non-deterministic) backward makes the stable-regime fixture train worse than the deterministic mode?
mkdir -p gpurun_out/r6
for v in S3D_LN_BWD_FUSE=0 S3D_WGRAD_GROUP=0 S3D_FUSE_LOSS_END=0 S3D_WGRAD_OVERWRITE=0 S3D_FUSED_BWD=0 S3D_FUSED_BLOCKS=0 S3D_DETERMINISTIC=1; do
  env $v python tools/r6/backward_ablation.py bf16 3 > gpurun_out/r6/ablation3_$v.txt 2>&1
  echo "$v $(grep -h '^bf16' gpurun_out/r6/ablation3_$v.txt)"
done
