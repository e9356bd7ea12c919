#!/usr/bin/env python
"""Large-M forward GEMM shapes of cfg-3 (deit_base, M = tokens of a slice of the batch) through the C ABI (tuning aid).
Env: S3D_GEMM_DMA=1 selects the LDS-DMA 128x128 kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_sweep import run  # noqa: E402

M = int(os.environ.get('M', '65536'))
for split in (1, 0):
    for name, N, K, epi in [('qkv', 2304, 768, 'BF16_BIAS'), ('proj', 768, 768, 'RESID'), ('fc1', 3072, 768, 'GELU'), ('fc2', 768, 3072, 'RESID')]:
        us = run(M, N, K, split, epi)
        print(f'{name:5s} split={split} M={M} N={N:5d} K={K:5d}  {us:9.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s(alg)', flush=True)
