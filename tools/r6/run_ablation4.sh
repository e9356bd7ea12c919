#!/bin/bash
# 40 seeds per mode: is the shipped default's deficit on the stable-regime fixture real?
mkdir -p gpurun_out/r6
export SEEDS=9-48
DEVICE=cuda python tools/r6/bwd_precision_emulation.py xxx > gpurun_out/r6/ablation4_emul_fp32.txt 2>&1
python tools/r6/backward_ablation.py bf16 1 > gpurun_out/r6/ablation4_default.txt 2>&1
S3D_DETERMINISTIC=1 python tools/r6/backward_ablation.py bf16 1 > gpurun_out/r6/ablation4_deterministic.txt 2>&1
python tools/r6/backward_ablation.py precise 1 > gpurun_out/r6/ablation4_precise.txt 2>&1
S3D_FUSED_BLOCKS=0 python tools/r6/backward_ablation.py bf16 1 > gpurun_out/r6/ablation4_unfused_blocks.txt 2>&1
for f in gpurun_out/r6/ablation4_*.txt; do echo "$f $(grep -h '^bf16\|^precise\|^xxx' $f)"; done
