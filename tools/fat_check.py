#!/usr/bin/env python
"""256x256 split forward tile (S3D_GEMM_NT_FAT=3, tuning build): outputs against fp64 on the same operand planes + timing at the
cfg-3 shapes (tuning aid).  Env: M (rows, default 94080), CHECK_M (rows of the parity pass, default 20037)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple3d_former_amd import ops  # noqa: E402
from tools.gemm_bench import planes, DEV  # noqa: E402


def timeit(fn, n=4, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n * reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


def setup(M, N, K):
    ah, al = planes(M, K); bh, bl = planes(N, K)
    bh, bl = (bh.float() * K ** -0.5).bfloat16(), (bl.float() * K ** -0.5).bfloat16()
    bias = torch.randn(N, device=DEV)
    R = torch.randn(M, N, device=DEV); C = torch.empty(M, N, device=DEV)
    oh = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); ol = torch.empty_like(oh); aux = torch.empty_like(oh)
    return dict(A_hi=ah, A_lo=al, lda=K, B_hi=bh, B_lo=bl, ldb=K, M=M, N=N, K=K, bias=bias, R=R, ldr=N, C=C, ldc=N, O_hi=oh, O_lo=ol,
                ldo=N, aux=aux, ldaux=N)


def check(M):
    worst = 0.0
    for name, N, K, epi in [('qkv', 2304, 768, 'BF16_BIAS'), ('proj', 768, 768, 'RESID'), ('fc1', 3072, 768, 'GELU'), ('fc2', 768, 3072, 'RESID')]:
        f = setup(M, N, K)
        for t in (f['C'], f['O_hi'], f['O_lo'], f['aux']):
            t.fill_(float('nan'))
        ops.gemm(0, 0, 1, epi, **f)
        torch.cuda.synchronize()
        rows = torch.cat([torch.arange(0, 300), torch.randint(0, M, (600,)), torch.arange(M - 300, M)]).to(DEV)
        a = (f['A_hi'][rows].double() + f['A_lo'][rows].double())
        b = (f['B_hi'].double() + f['B_lo'].double())
        # the product drops lo x lo: remove it from the reference too
        ref = a @ b.t() - f['A_lo'][rows].double() @ f['B_lo'].double().t() + f['bias'].double()
        if epi == 'RESID':
            got = f['C'][rows].double(); ref = ref + f['R'][rows].double()
        elif epi == 'BF16_BIAS':
            got = f['O_hi'][rows].double() + f['O_lo'][rows].double()
        else:
            pre = f['aux'][rows].double()
            e_pre = (pre - ref).abs().max().item()
            got = f['O_hi'][rows].double() + f['O_lo'][rows].double(); ref = torch.nn.functional.gelu(ref)
            print(f'   fc1 pre-activation (bf16) max err {e_pre:.3e}')
        err = (got - ref).abs().max().item()
        nan = sum(int(torch.isnan(t.float()).any()) for t in ((f['C'],) if epi == 'RESID' else (f['O_hi'], f['O_lo'])))
        print(f'check {name:5s} {epi:10s} M={M} max abs err {err:.3e}  nan-planes {nan}', flush=True)
        worst = max(worst, err)
    return worst


if __name__ == '__main__':
    print('S3D_GEMM_NT_FAT =', os.environ.get('S3D_GEMM_NT_FAT'), ' lib', os.environ.get('S3D_LIB_PATH'))
    w = check(int(os.environ.get('CHECK_M', '20037')))
    M = int(os.environ.get('M', '94080'))
    for name, N, K, epi in [('qkv', 2304, 768, 'BF16_BIAS'), ('proj', 768, 768, 'RESID'), ('fc1', 3072, 768, 'GELU'), ('fc2', 768, 3072, 'RESID')]:
        f = setup(M, N, K)
        for _ in range(int(os.environ.get('LOOPS', '1'))):
            us = timeit(lambda: ops.gemm(0, 0, 1, epi, **f))
        print(f'{name:5s} M={M} N={N:5d} K={K:5d}  {us:9.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s(alg)', flush=True)
    print('worst err', w)
