#!/usr/bin/env python
"""K / M sweep of one forward GEMM shape: separates the fixed cost (launch ramp, first-load latency, epilogue) from the
per-k-tile cost (tuning aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple3d_former_amd import ops  # noqa: E402
from tools.gemm_bench import timeit, planes, DEV  # noqa: E402


def run(M, N, K, split, epi='GELU'):
    ah, al = planes(M, K); bh, bl = planes(N, K)
    bias = torch.randn(N, device=DEV)
    R = torch.randn(M, N, device=DEV); C = torch.empty(M, N, device=DEV)
    oh = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); ol = torch.empty_like(oh); aux = torch.empty_like(oh)

    def f():
        ops.gemm(0, 0, split, epi, A_hi=ah, A_lo=al, lda=K, B_hi=bh, B_lo=bl, ldb=K, M=M, N=N, K=K, bias=bias, R=R, ldr=N,
                 C=C, ldc=N, O_hi=oh, O_lo=ol, ldo=N, aux=aux, ldaux=N)
    return timeit(f)


if __name__ == '__main__':
    for split in (1, 0):
        for N in (384, 1536):
            for K in (64, 128, 256, 384, 768, 1536):
                print(f'split={split} M=1664 N={N:5d} K={K:5d}  {run(1664, N, K, split):7.2f} us', flush=True)
    for M in (416, 832, 1664, 3328, 6656):
        print(f'split=1 M={M:5d} N=1536 K=384  {run(M, 1536, 384, 1):7.2f} us', flush=True)
    print('--- epilogue cost at K=64 (fixed part)')
    for epi in ('GELU', 'BF16_BIAS', 'F32', 'RESID'):
        for M in (32, 416, 1664):
            print(f'epi={epi:10s} split=1 M={M:5d} N=1536 K=64  {run(M, 1536, 64, 1, epi):7.2f} us   split=0 {run(M, 1536, 64, 0, epi):7.2f} us', flush=True)
