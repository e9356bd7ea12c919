#!/usr/bin/env python
"""Yardstick only (not part of the product): hipBLASLt bf16 GEMM times via torch.matmul on the cfg-3 backward / forward shapes."""
import os
import torch

DEV = 'cuda'
ROWS = int(os.environ.get('ROWS', '94080'))


def timeit(fn, n=5, reps=4):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n * reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


for name, O, I in [('qkv', 2304, 768), ('proj', 768, 768), ('fc1', 3072, 768), ('fc2', 768, 3072)]:
    dy = torch.randn(ROWS, O, device=DEV).bfloat16()
    x = torch.randn(ROWS, I, device=DEV).bfloat16()
    w = torch.randn(O, I, device=DEV).bfloat16()
    for tag, f in (('fwd   x@W^T ', lambda: torch.matmul(x, w.t())), ('dgrad dy@W  ', lambda: torch.matmul(dy, w)),
                   ('wgrad dy^T@x', lambda: torch.matmul(dy.t(), x))):
        us = timeit(f)
        print(f'{name:5s} {tag} rows={ROWS} out={O:5d} in={I:5d}  {us:9.1f} us  {2.0 * ROWS * O * I / us / 1e6:8.1f} TFLOP/s', flush=True)
