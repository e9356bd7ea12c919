#!/usr/bin/env python
"""In-kernel timeline of the cfg-2 forward GEMM launches (debug build: make -C simple3d-former_amd/csrc TL=1).

    S3D_LIB_PATH=simple3d-former_amd/libs3d_hip_tl.so python tools/timeline_probe.py

Wave 0 of every workgroup stamps s_memtime at the phase boundaries of gemm_nt_dma_kernel; this prints, per launch, where the
workgroups spend their cycles (prologue issue, first tile landed, per-k-tile cadence, epilogue, store acknowledgement), how the
workgroup start / end times spread over the kernel's span and how many workgroups each CU ran."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple3d_former_amd import _lib as L, ops  # noqa: E402

SLOTS = 40
M, D = int(os.environ.get('TL_M', '1664')), int(os.environ.get('TL_D', '384'))      # TL_M=32768 TL_D=768 TL_BK=32 S3D_GEMM_NT_TILE=2: cfg-3
DEV = 'cuda'
TILES = {0: (32, 64), 1: (64, 64), 3: (32, 32), 2: (128, 128), 4: (128, 96), 5: (64, 128), 6: (64, 64), 7: (128, 128), 8: (64, 96), 9: (32, 64), 10: (32, 32), 11: (128, 128), 12: (64, 64), 13: (32, 64), 14: (128, 128), 15: (64, 192), 16: (128, 128), 17: (128, 96), 18: (64, 64), 19: (32, 64)}


def planes(r, c):
    return ops.split_bf16(torch.randn(r, c, device=DEV))


def run(name, N, K, epi, tile, cold):
    bm, bn = {'1': (128, 256), '2': (128, 256), '3': (256, 256)}.get(os.environ.get('S3D_GEMM_NT_FAT', ''), TILES[tile]) if tile == 2 and M >= 8192 else TILES[tile]
    nwg = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
    ah, al = planes(M, K); bh, bl = planes(N, K)
    bias = torch.randn(N, device=DEV)
    R = torch.randn(M, N, device=DEV); C = torch.empty(M, N, device=DEV)
    oh = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); ol = torch.empty_like(oh); aux = torch.empty_like(oh)
    buf = torch.zeros(nwg, SLOTS, dtype=torch.int64, device=DEV)
    junk = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=DEV)
    lib = L.lib()

    def f():
        ops.gemm(0, 0, 1, epi, A_hi=ah, A_lo=al, lda=K, B_hi=bh, B_lo=bl, ldb=K, M=M, N=N, K=K, bias=bias, R=R, ldr=N,
                 C=C, ldc=N, O_hi=oh, O_lo=ol, ldo=N, aux=aux, ldaux=N)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    assert lib.s3d_debug_timeline_set(ctypes.c_void_p(buf.data_ptr())) == 0
    if cold:
        junk.fill_(1.0)                      # 1 GB stream: evicts L2 and the Infinity Cache (as Adam does before the forward)
        (ah.float().sum() + al.float().sum()).item()        # the A planes were just written by the LayerNorm in the real step
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record()
    torch.cuda.synchronize()
    lib.s3d_debug_timeline_set(ctypes.c_void_p(0))
    t = buf.cpu().numpy().astype(np.int64)
    nk = K // int(os.environ.get('TL_BK', '64'))
    real0, real1 = t[:, 0], t[:, 7]
    span = (real1.max() - real0.min()) * 10e-3                      # us (100 MHz)
    start = (real0 - real0.min()) * 10e-3
    dur = (real1 - real0) * 10e-3
    hw = t[:, 1]
    xcc = hw >> 32
    cu = ((hw & 0xffffffff) >> 8) & 0xf
    se = ((hw & 0xffffffff) >> 13) & 0x7
    sh = ((hw & 0xffffffff) >> 12) & 0x1
    cuid = xcc * 1000 + se * 100 + sh * 20 + cu
    uniq, cnt = np.unique(cuid, return_counts=True)
    c = t[:, 2:].astype(np.float64)
    c2 = t[:, 2]
    issue = t[:, 3] - c2
    first = t[:, 8] - c2
    ksteps = np.diff(t[:, 8:8 + nk], axis=1) if nk > 1 else np.zeros((nwg, 1))
    tail = t[:, 4] - t[:, 8 + nk - 1]
    epi_issue = t[:, 5] - t[:, 4]
    ack = t[:, 6] - t[:, 5]
    total = t[:, 6] - c2
    print(f'--- {name} N={N} K={K} tile {bm}x{bn} ({nwg} WGs) {"COLD" if cold else "warm"}: events {e0.elapsed_time(e1) * 1e3:.1f} us, '
          f'first-start..last-end {span:.2f} us')
    print(f'    WG start offset us: p50 {np.percentile(start, 50):.2f} p90 {np.percentile(start, 90):.2f} max {start.max():.2f};  '
          f'WG duration us: p10 {np.percentile(dur, 10):.2f} p50 {np.percentile(dur, 50):.2f} p90 {np.percentile(dur, 90):.2f} max {dur.max():.2f}')
    print(f'    CUs used {len(uniq)}, WGs/CU min {cnt.min()} max {cnt.max()};  clock ~ {np.median(total / np.maximum(dur, 1e-3)) / 1e3:.2f} GHz')
    f3 = lambda a: f'{np.percentile(a, 10):.0f}/{np.percentile(a, 50):.0f}/{np.percentile(a, 90):.0f}'
    print(f'    cycles p10/p50/p90: prologue issue {f3(issue)}  first tile landed {f3(first)}  k-step {f3(ksteps.reshape(-1))} '
          f'(x{nk - 1})  last compute {f3(tail)}  epilogue issue {f3(epi_issue)}  store ack {f3(ack)}  total {f3(total)}')
    if nk > 1:
        print('    k-step medians:', ' '.join(f'{np.median(ksteps[:, i]):.0f}' for i in range(nk - 1)))
    if t[:, 32].any():         # fat tile: inside k-step 4 -- landed barrier, fragments read, release barrier, MFMA block / DMA piece 0, 1, 7
        names = ['frags read', 'release barrier', 'mfma blk0 issued', 'dma piece0 issued', 'mfma blk1 issued', 'dma piece1 issued', 'mfma blk7 issued', 'dma piece7 issued']
        base = t[:, 8 + 4]
        print('    inside k-step 4 (cycles after the landed barrier, p50):', '  '.join(f'{n} {np.median(t[:, 32 + i] - base):.0f}' for i, n in enumerate(names)),
              f'  next landed barrier {np.median(t[:, 8 + 5] - base):.0f}')
    early = start < np.percentile(start, 30)
    late = start > np.percentile(start, 70)
    print(f'    early WGs (first 30%): total p50 {np.median(total[early]):.0f} cyc; late WGs (last 30%): total p50 {np.median(total[late]):.0f} cyc')


if __name__ == '__main__':
    shapes = [('qkv', 3 * D, D, 'BF16_BIAS'), ('proj', D, D, 'RESID'), ('fc1', 4 * D, D, 'GELU'), ('fc2', D, 4 * D, 'RESID')]
    only = os.environ.get('ONLY', '')
    for name, N, K, epi in shapes:
        if only and name not in only.split(','):
            continue
        # the library reads S3D_GEMM_NT_TILE once per process: one tile choice per run (0 = 32x64, 1 = 64x64, 3 = 32x32)
        tile = int(os.environ.get('S3D_GEMM_NT_TILE', {'qkv': 1, 'proj': 3, 'fc1': 0, 'fc2': 3}[name]))
        for cold in (True, False):
            run(name, N, K, epi, tile, cold)
