#!/usr/bin/env python
"""cfg-3 backward GEMM shapes (rows = tokens of a slice of the batch) through the C ABI: TN wgrads with split-K atomics and NN dgrads
(tuning aid).  Env: ROWS (default 65536), S3D_GEMM_SPLITK to force the split."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple3d_former_amd import ops  # noqa: E402
from tools.gemm_bench import timeit, DEV  # noqa: E402

ROWS = int(os.environ.get('ROWS', '65536'))
for name, O, I in [('qkv', 2304, 768), ('proj', 768, 768), ('fc1', 3072, 768), ('fc2', 768, 3072)]:
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
