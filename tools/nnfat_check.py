#!/usr/bin/env python
"""256x256 dgrad tile (default; S3D_DGRAD_FAT=2 in the tuning build = the 128x128 / 256x128 tiles): outputs against fp64 on the same bf16 operands + timing at the cfg-3 shapes."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple3d_former_amd import ops  # noqa: E402
from tools.fat_check import timeit  # noqa: E402

DEV = 'cuda'
ROWS = int(os.environ.get('ROWS', '188160'))
print('S3D_DGRAD_FAT =', os.environ.get('S3D_DGRAD_FAT'))
for name, O, I, epi in [('qkv', 2304, 768, 'F32'), ('fc1', 3072, 768, 'F32'), ('fc2', 768, 3072, 'DGELU'), ('proj', 768, 768, 'BF16_BIAS')]:
    g = torch.Generator().manual_seed(O + I)
    dy = torch.randn(ROWS, O, generator=g).to(DEV).bfloat16()
    w = (torch.randn(O, I, generator=g) * O ** -0.5).to(DEV).bfloat16()
    aux = torch.randn(ROWS, I, generator=g).to(DEV).bfloat16()
    dx32 = torch.full((ROWS, I), float('nan'), device=DEV)
    dx16 = torch.full((ROWS, I), float('nan'), device=DEV, dtype=torch.bfloat16)
    kw = dict(A_hi=dy, lda=O, B_hi=w, ldb=I, M=ROWS, N=I, K=O)
    if epi == 'F32':
        kw.update(C=dx32, ldc=I)
    else:
        kw.update(O_hi=dx16, ldo=I)
        if epi == 'DGELU':
            kw.update(aux=aux, ldaux=I)
    f = lambda: ops.gemm(0, 1, 0, epi, **kw)
    f(); torch.cuda.synchronize()
    rows = torch.cat([torch.arange(0, 300), torch.randint(0, ROWS, (500,)), torch.arange(ROWS - 300, ROWS)]).to(DEV)
    ref = dy[rows].double() @ w.double()
    if epi == 'DGELU':
        a = aux[rows].double().requires_grad_(True)
        ref = ref * torch.autograd.grad(F.gelu(a).sum(), a)[0]
    got = (dx32 if epi == 'F32' else dx16)[rows].double()
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    us = timeit(f)
    print(f'{name:5s} {epi:10s} rows={ROWS} K={O:5d} N={I:5d}  {us:9.1f} us  {2.0 * ROWS * O * I / us / 1e6:8.1f} TFLOP/s   rel err {err:.2e}  nan {int(torch.isnan(got).any())}', flush=True)
