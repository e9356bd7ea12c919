#!/usr/bin/env python
"""In-kernel timeline of the fused block launches of the cfg-2 step (debug build: make -C simple3d-former_amd/csrc TL=1).

    S3D_LIB_PATH=simple3d-former_amd/libs3d_hip_tl.so python tools/fused_timeline_probe.py [--batch 64]

Thread 0 of every workgroup of blk_attn / blk_mlp1 / blk_attn_bwd stamps s_memtime at its phase boundaries (fused_block.hip: FB_TL)
while the captured step graph replays; the stamps of the LAST block of the step survive.  Prints, per kernel, the median (p10 .. p90)
over workgroups of every phase's length in shader cycles, and the launch's span from the chip-wide 100 MHz clock."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import simple3d_former_amd as s3d  # noqa: E402
from simple3d_former_amd import _lib as L  # noqa: E402
from oracle import voxel_oracle as vo  # noqa: E402  (synthetic inputs only)

SLOTS = 32
PHASES = {
    'blk_attn': (0, 192, [(1, 2, 'entry -> addresses'), (2, 3, 'ring prologue issued'), (3, 4, 'rows arrived + LN statistics'), (4, 5, 'barrier (gamma / beta)'),
                          (5, 6, 'slab 0 + stage 0 landed'), (6, 17, 'k-steps 0 .. 10'), (10, 26, '  k-step 4: ring refill issued'), (26, 27, '  k-step 4: fragments read'), (27, 28, '  k-step 4: MFMAs issued'), (28, 11, '  k-step 4: wait + barrier of k-step 5'),
                          (17, 20, 'last k-step'), (20, 21, 'drain barrier'),
                          (21, 22, 'q|k|v staged + stored'), (22, 23, 'attention (wave 0)'), (23, 24, 'stores acknowledged')]),
    'blk_mlp1': (256, 256, [(1, 2, 'entry -> addresses'), (2, 3, 'ring prologue issued'), (3, 4, 'rows arrived + LN statistics'), (4, 5, 'barrier (gamma / beta)'),
                            (5, 6, 'slab 0 + stage 0 landed'), (6, 17, 'k-steps 0 .. 10'), (10, 26, '  k-step 4: ring refill issued'), (26, 27, '  k-step 4: fragments read'), (27, 28, '  k-step 4: MFMAs issued'), (28, 11, '  k-step 4: wait + barrier of k-step 5'),
                            (17, 20, 'last k-step'), (20, 21, 'drain barrier'),
                            (21, 22, 'bias + GELU + split -> LDS'), (22, 23, 'row stores issued'), (23, 24, 'stores acknowledged')]),
    'blk_attn_bwd': (512, 192, [(1, 2, 'staging loads -> LDS (this thread)'), (2, 3, 'barrier'), (3, 4, 'proj dgrad (12 k-tiles) + dO tile'), (4, 5, 'barrier'),
                                (5, 6, 'S, dP, delta + barrier'), (6, 7, 'dQ (wave 0) + stores issued'), (7, 8, 'stores acknowledged')]),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    args = ap.parse_args()
    conf = bench.CONFIGS['cfg2']
    dev = torch.device('cuda', 0)
    eng = s3d.VoxelEngine(device=dev, split=True, pos_embedding=conf['pos_embedding'], **conf['cfg'])
    eng.load_state_dict(vo.init_state_dict(seed=9, pos_embedding=conf['pos_embedding'], **conf['cfg']))
    eng.set_optimizer(lr=1e-3)
    x, y = vo.synthetic_batch(args.batch, conf['cfg']['voxel_size'], conf['cfg']['n_classes'], seed=9)
    g, sx, sy, _ = eng.capture_train_step(args.batch)
    sx.copy_(x.to(dev)); sy.copy_(y.to(dev))
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    lib = L.lib()
    buf = torch.zeros(768, SLOTS, dtype=torch.int64, device=dev)
    assert lib.s3d_debug_fused_timeline_set(ctypes.c_void_p(buf.data_ptr())) == 0
    g.replay()
    torch.cuda.synchronize()
    lib.s3d_debug_fused_timeline_set(ctypes.c_void_p(0))
    t = buf.cpu().numpy().astype(np.int64)
    for name, (base, n, phases) in PHASES.items():
        rows = t[base:base + n]
        rows = rows[rows[:, 1] != 0]
        if len(rows) == 0:
            print(f'{name}: no stamps'); continue
        end_slot = 25 if name != 'blk_attn_bwd' else 9
        done = rows[rows[:, end_slot] != 0]
        span = (done[:, end_slot].max() - done[:, 0].min()) * 10e-3
        dur = (done[:, end_slot] - done[:, 0]) * 10e-3
        start = (rows[:, 0] - rows[:, 0].min()) * 10e-3
        print(f'== {name}: {len(rows)} workgroups stamped, launch span {span:.2f} us; per-workgroup lifetime median {np.median(dur):.2f} us '
              f'(min {dur.min():.2f}, max {dur.max():.2f}); start spread median {np.median(start):.2f} us, max {start.max():.2f} us')
        total = 0
        for a, b, label in phases:
            ok = (rows[:, a] != 0) & (rows[:, b] != 0)
            d = (rows[ok, b] - rows[ok, a]).astype(np.float64)
            if len(d) == 0:
                continue
            if not label.startswith('  '):
                total += np.median(d)
            print(f'   {label:38s} median {np.median(d):8.0f} cycles   p10 {np.percentile(d, 10):8.0f}   p90 {np.percentile(d, 90):8.0f}')
        print(f'   {"sum of medians":38s}        {total:8.0f} cycles')


if __name__ == '__main__':
    main()
