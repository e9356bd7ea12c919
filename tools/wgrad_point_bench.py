#!/usr/bin/env python
"""Point-path transformer-block backward GEMMs (deit_tiny: D = 192, rows = B x 257 tokens) through the C ABI (tuning aid).
Env: ROWS (32896 = cfg-4, 16416 = cfg-5), S3D_GEMM_TILE / S3D_GEMM_SPLITK to force tile and split."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple3d_former_amd import ops  # noqa: E402
from tools.gemm_bench import timeit, DEV  # noqa: E402

ROWS = int(os.environ.get('ROWS', '32896'))
D = int(os.environ.get('D', '192'))
for name, O, I in [('qkv', 3 * D, D), ('proj', D, D), ('fc1', 4 * D, D), ('fc2', D, 4 * D)]:
    dy = torch.randn(ROWS, O, device=DEV).bfloat16()
    x = torch.randn(ROWS, I, device=DEV).bfloat16()
    w = torch.randn(O, I, device=DEV).bfloat16()
    dW = torch.zeros(O, I, device=DEV); db = torch.zeros(O, device=DEV); dx = torch.empty(ROWS, I, device=DEV)

    def wg():
        ops.gemm(1, 1, 0, 'ATOMIC', splitk=0, A_hi=dy, lda=O, B_hi=x, ldb=I, M=O, N=I, K=ROWS, C=dW, ldc=I, bias_grad=db)

    def dg():
        ops.gemm(0, 1, 0, 'F32', A_hi=dy, lda=O, B_hi=w, ldb=I, M=ROWS, N=I, K=O, C=dx, ldc=I)
    for tag, f in (('wgrad', wg), ('dgrad', dg)):
        us = timeit(f)
        print(f'{name:5s} {tag} rows={ROWS} out={O:5d} in={I:5d}  {us:9.1f} us  {2.0 * ROWS * O * I / us / 1e6:8.1f} TFLOP/s', flush=True)
