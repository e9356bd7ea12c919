#!/usr/bin/env python
"""Micro-benchmark of the cfg-2 GEMM launches through the C ABI (tuning aid).  Env: S3D_GEMM_TILE, S3D_GEMM_SPLITK."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple3d_former_amd import ops  # noqa: E402

M, D = 1664, 384
DEV = 'cuda'


def timeit(fn, n=20, reps=10):
    """Device-side time per launch: n launches captured into one HIP graph (no host launch cost), replayed reps times."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


def planes(r, c):
    t = torch.randn(r, c, device=DEV)
    return ops.split_bf16(t)


def main():
    only = os.environ.get('ONLY', '')
    res = []
    # forward NT split: (name, N, K, epi)
    for name, N, K, epi in [('qkv', 3 * D, D, 'BF16_BIAS'), ('proj', D, D, 'RESID'), ('fc1', 4 * D, D, 'GELU'), ('fc2', D, 4 * D, 'RESID')]:
        if only and only != 'fwd_' + name:
            continue
        ah, al = planes(M, K); bh, bl = planes(N, K)
        bias = torch.randn(N, device=DEV)
        R = torch.randn(M, N, device=DEV); C = torch.empty(M, N, device=DEV)
        oh = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); ol = torch.empty_like(oh); aux = torch.empty_like(oh)
        for split in (1, 0):
            def f():
                ops.gemm(0, 0, split, epi, A_hi=ah, A_lo=al, lda=K, B_hi=bh, B_lo=bl, ldb=K, M=M, N=N, K=K, bias=bias, R=R, ldr=N,
                         C=C, ldc=N, O_hi=oh, O_lo=ol, ldo=N, aux=aux, ldaux=N)
            us = timeit(f)
            res.append((f'fwd {name} split={split}', M, N, K, us))
    # dgrad NN
    for name, N, K, epi in [('fc2', 4 * D, D, 'DGELU'), ('fc1', D, 4 * D, 'F32'), ('proj', D, D, 'BF16_BIAS'), ('qkv', D, 3 * D, 'F32')]:
        if only and only != 'dgrad_' + name:
            continue
        ah, _ = planes(M, K); bh, _ = planes(K, N)
        C = torch.empty(M, N, device=DEV); oh = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); aux = torch.randn(M, N, device=DEV).to(torch.bfloat16)
        def f():
            ops.gemm(0, 1, 0, epi, A_hi=ah, lda=K, B_hi=bh, ldb=N, M=M, N=N, K=K, C=C, ldc=N, O_hi=oh, ldo=N, aux=aux, ldaux=N)
        res.append((f'dgrad {name}', M, N, K, timeit(f)))
    # wgrad TN: dW[O][I] = dy[M][O]^T x[M][I]
    for name, O, I in [('fc2', D, 4 * D), ('fc1', 4 * D, D), ('proj', D, D), ('qkv', 3 * D, D)]:
        if only and only != 'wgrad_' + name:
            continue
        dy, _ = planes(M, O); x, _ = planes(M, I)
        dW = torch.zeros(O, I, device=DEV); db = torch.zeros(O, device=DEV)
        def f():
            ops.gemm(1, 1, 0, 'ATOMIC', splitk=0, A_hi=dy, lda=O, B_hi=x, ldb=I, M=O, N=I, K=M, C=dW, ldc=I, bias_grad=db)
        res.append((f'wgrad {name}', O, I, M, timeit(f)))
    tot = 0
    for name, m, n, k, us in res:
        fl = 2.0 * m * n * k
        print(f'{name:22s} M={m:5d} N={n:5d} K={k:5d}  {us:8.2f} us  {fl / us / 1e6:8.1f} TFLOP/s(alg)')
        if 'split=0' not in name:
            tot += us
    print(f'sum (one block fwd split + bwd) = {tot:.1f} us  -> x12 = {tot * 12 / 1e3:.2f} ms; tag={os.environ.get("TAG", "")}'
          f' tile={os.environ.get("S3D_GEMM_TILE", "auto")} splitk={os.environ.get("S3D_GEMM_SPLITK", "auto")}')


if __name__ == '__main__':
    main()
