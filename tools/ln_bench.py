#!/usr/bin/env python
"""LayerNorm fwd / bwd micro-benchmark at the cfg-2 shape (tuning aid): with and without the dgamma/dbeta reduction."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple3d_former_amd import _lib as L  # noqa: E402
from tools.gemm_bench import timeit, DEV  # noqa: E402

rows, D = 1664, 384
x = torch.randn(rows, D, device=DEV); dy = torch.randn(rows, D, device=DEV); dres = torch.randn(rows, D, device=DEV)
mean = x.mean(1); rstd = (x.var(1, unbiased=False) + 1e-6).rsqrt(); gamma = torch.randn(D, device=DEV)
dx = torch.empty_like(x); dx_bf = torch.empty(rows, D, dtype=torch.bfloat16, device=DEV)
dg = torch.zeros(D, device=DEV); db = torch.zeros(D, device=DEV)
lib = L.lib()
for name, kw in [('full', dict(dgamma=dg, dbeta=db)), ('no-dgamma', dict()), ('no-dgamma no-dres', dict(nodres=1)),
                 ('full, no dx_bf', dict(dgamma=dg, dbeta=db, nobf=1))]:
    a = L.fill(L.S3dLnBwdArgs(), dy=dy, lddy=D, x=x, ldx=D, mean=mean, rstd=rstd, gamma=gamma, dres=None if kw.get('nodres') else dres,
               lddres=D, dx=dx, lddx=D, dx_bf=None if kw.get('nobf') else dx_bf, lddxbf=D, rows=rows, D=D,
               **{k: v for k, v in kw.items() if k in ('dgamma', 'dbeta')})
    print(f'ln_bwd {name:20s} {timeit(lambda: L.check(lib.s3d_layernorm_bwd(ctypes.byref(a), L.current_stream()))):6.2f} us')
