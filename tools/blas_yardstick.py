#!/usr/bin/env python
"""Yardstick only (not part of the product): hipBLASLt/rocBLAS bf16 GEMM times via torch.matmul on the cfg-2 shapes,
measured with the same HIP-graph harness as tools/gemm_bench.py."""
import torch

DEV = 'cuda'


def timeit(fn, n=20, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


M, D = 1664, 384
for name, (m, n, k), tb in [('fwd qkv x@W^T', (M, 3 * D, D), True), ('fwd proj', (M, D, D), True), ('fwd fc1', (M, 4 * D, D), True),
                            ('fwd fc2', (M, D, 4 * D), True), ('dgrad fc2 dy@W', (M, 4 * D, D), False), ('dgrad fc1', (M, D, 4 * D), False),
                            ('wgrad fc1 dy^T@x', (4 * D, D, M), None)]:
    if tb is None:
        a = torch.randn(k, m, device=DEV, dtype=torch.bfloat16); b = torch.randn(k, n, device=DEV, dtype=torch.bfloat16)
        f = lambda: torch.matmul(a.t(), b)
    elif tb:
        a = torch.randn(m, k, device=DEV, dtype=torch.bfloat16); b = torch.randn(n, k, device=DEV, dtype=torch.bfloat16)
        f = lambda: torch.matmul(a, b.t())
    else:
        a = torch.randn(m, k, device=DEV, dtype=torch.bfloat16); b = torch.randn(k, n, device=DEV, dtype=torch.bfloat16)
        f = lambda: torch.matmul(a, b)
    us = timeit(f)
    print(f'{name:20s} M={m:5d} N={n:5d} K={k:5d}  {us:7.2f} us  {2.0 * m * n * k / us / 1e6:7.1f} TFLOP/s')
# elementwise floor: an empty-ish kernel
x = torch.zeros(64, device=DEV)
print('tiny kernel (x += 1) floor', round(timeit(lambda: x.add_(1)), 2), 'us')
