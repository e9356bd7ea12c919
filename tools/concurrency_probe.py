#!/usr/bin/env python
"""Does running two half-batch kernel chains on two graph branches beat one full-batch chain?  (tuning probe)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple3d_former_amd import ops  # noqa: E402
from tools.gemm_bench import planes, DEV  # noqa: E402


def make(M, N, K, epi):
    ah, al = planes(M, K); bh, bl = planes(N, K)
    bias = torch.randn(N, device=DEV)
    R = torch.randn(M, N, device=DEV); C = torch.empty(M, N, device=DEV)
    oh = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); ol = torch.empty_like(oh); aux = torch.empty_like(oh)

    def f():
        ops.gemm(0, 0, 1, epi, A_hi=ah, A_lo=al, lda=K, B_hi=bh, B_lo=bl, ldb=K, M=M, N=N, K=K, bias=bias, R=R, ldr=N,
                 C=C, ldc=N, O_hi=oh, O_lo=ol, ldo=N, aux=aux, ldaux=N)
    return f


def chain(M):
    fs = [make(M, 1152, 384, 'BF16_BIAS'), make(M, 384, 384, 'RESID'), make(M, 1536, 384, 'GELU'), make(M, 384, 1536, 'RESID')]

    def run():
        for _ in range(6):
            for f in fs:
                f()
    return run


def time_graph(build, reps=20):
    build()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        build()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


full = chain(1664)
ha, hb = chain(832), chain(832)
qs = [chain(416) for _ in range(4)]
side = [torch.cuda.Stream() for _ in range(3)]


def two():
    cur = torch.cuda.current_stream()
    side[0].wait_stream(cur)
    ha()
    with torch.cuda.stream(side[0]):
        hb()
    cur.wait_stream(side[0])


def four():
    cur = torch.cuda.current_stream()
    for s in side:
        s.wait_stream(cur)
    qs[0]()
    for s, q in zip(side, qs[1:]):
        with torch.cuda.stream(s):
            q()
    for s in side:
        cur.wait_stream(s)


def serial_halves():
    ha(); hb()


print(f'one chain  M=1664 (24 GEMMs)          {time_graph(full):8.1f} us')
print(f'two chains M=832 serial (48 GEMMs)    {time_graph(serial_halves):8.1f} us')
print(f'two chains M=832 on 2 graph branches  {time_graph(two):8.1f} us')
print(f'four chains M=416 on 4 graph branches {time_graph(four):8.1f} us')
