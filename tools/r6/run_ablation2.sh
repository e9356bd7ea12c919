#!/bin/bash
# second ablation call: statistics of the shipped default, the deterministic mode, and the unstable-regime fixture under emulation
mkdir -p gpurun_out/r6
python tools/r6/backward_ablation.py bf16 4 > gpurun_out/r6/ablation2_default_x4.txt 2>&1
S3D_DETERMINISTIC=1 python tools/r6/backward_ablation.py bf16 1 > gpurun_out/r6/ablation2_deterministic.txt 2>&1
FIXTURE=adam60 python tools/r6/backward_ablation.py bf16,precise 2 > gpurun_out/r6/ablation2_adam60_hip.txt 2>&1
FIXTURE=adam60 DEVICE=cuda python tools/r6/bwd_precision_emulation.py xxx,bbb,sss > gpurun_out/r6/ablation2_adam60_emul.txt 2>&1
grep -h "^bf16\|^precise\|^xxx\|^bbb\|^sss\|Error\|error" gpurun_out/r6/ablation2_*.txt
