#!/usr/bin/env python
"""In-kernel timeline of the fused block launches (csrc/fused_block.hip; debug build: make -C simple3d-former_amd/csrc TL=1).

    S3D_LIB_PATH=simple3d-former_amd/libs3d_hip_tl.so python tools/fused_timeline_probe.py

Lane 0 of every workgroup stamps s_memtime at: 0 entry, 1 own residual row loaded + statistics, 2 A operand in LDS (barrier),
3 main loop done, 4 outputs staged (barrier), 5 / 6 stores issued, 7 stores acknowledged.  Prints the median cycles per phase."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple3d_former_amd import _lib as L  # noqa: E402
from simple3d_former_amd.engine import ParamArena, _BlockWorkspace  # noqa: E402

DEV = 'cuda'
D, H, N, Bb = 384, 6, 26, int(os.environ.get('TL_B', '64'))
Hd, M = 4 * D, Bb * N
shapes = {}
p = 'blocks.0.'
shapes.update({p + 'norm1.weight': (D,), p + 'norm1.bias': (D,), p + 'attn.qkv.weight': (3 * D, D), p + 'attn.qkv.bias': (3 * D,),
               p + 'attn.proj.weight': (D, D), p + 'attn.proj.bias': (D,), p + 'norm2.weight': (D,), p + 'norm2.bias': (D,),
               p + 'mlp.fc1.weight': (Hd, D), p + 'mlp.fc1.bias': (Hd,), p + 'mlp.fc2.weight': (D, Hd), p + 'mlp.fc2.bias': (D,)})
g = torch.Generator().manual_seed(1)
sd = {k: torch.randn(s, generator=g) * 0.05 + (1.0 if 'norm' in k and k.endswith('weight') else 0.0) for k, s in shapes.items()}
arena = ParamArena(shapes, torch.device(DEV)); arena.load(sd); arena.refresh_planes()
bp = (L.S3dBlockParams * 1)()
L.fill(bp[0], ln1_w=arena.param(p + 'norm1.weight'), ln1_b=arena.param(p + 'norm1.bias'), ln2_w=arena.param(p + 'norm2.weight'),
       ln2_b=arena.param(p + 'norm2.bias'), qkv_b=arena.param(p + 'attn.qkv.bias'), proj_b=arena.param(p + 'attn.proj.bias'),
       fc1_b=arena.param(p + 'mlp.fc1.bias'), fc2_b=arena.param(p + 'mlp.fc2.bias'),
       qkv_w_hi=arena.hi_of(p + 'attn.qkv.weight'), qkv_w_lo=arena.lo_of(p + 'attn.qkv.weight'),
       proj_w_hi=arena.hi_of(p + 'attn.proj.weight'), proj_w_lo=arena.lo_of(p + 'attn.proj.weight'),
       fc1_w_hi=arena.hi_of(p + 'mlp.fc1.weight'), fc1_w_lo=arena.lo_of(p + 'mlp.fc1.weight'),
       fc2_w_hi=arena.hi_of(p + 'mlp.fc2.weight'), fc2_w_lo=arena.lo_of(p + 'mlp.fc2.weight'),
       qkv_wp_hi=arena.hi_pk_of(p + 'attn.qkv.weight'), qkv_wp_lo=arena.lo_pk_of(p + 'attn.qkv.weight'),
       fc1_wp_hi=arena.hi_pk_of(p + 'mlp.fc1.weight'), fc1_wp_lo=arena.lo_pk_of(p + 'mlp.fc1.weight'))
ws = _BlockWorkspace(1, Bb, N, D, H, Hd, DEV, True, fuse=True)
ws.x[0].copy_(torch.randn(M, D, generator=g))
lib = L.lib()
NWG = 512
buf = torch.zeros(NWG, 16, dtype=torch.int64, device=DEV)
junk = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device=DEV)


def fwd():
    L.check(lib.s3d_blocks_fwd(ctypes.byref(ws.shape), bp, ws.acts, 1, L.current_stream()), 'blocks_fwd')


for _ in range(3):
    fwd()
torch.cuda.synchronize()
NAMES = ['rows + LN -> LDS', 'barrier', 'main loop', 'outputs staged', 'copy-out issued', 'tail (attention)', 'stores acked']
for cold in (False, True):
    # the two fused kernels run in one blocks_fwd call: the second overwrites the first's stamps of the same workgroup index, so stamp
    # them in separate calls by moving the buffer (attn uses 192 workgroups, mlp1 208: offset the mlp1 run? -> run twice with masks)
    for which in ('attn', 'mlp1'):
        buf.zero_()
        if cold:
            junk.fill_(1.0)
        torch.cuda.synchronize()
        os.environ['S3D_FB_TL_ONLY'] = which
        assert lib.s3d_debug_fused_timeline_set(ctypes.c_void_p(buf.data_ptr())) == 0
        assert lib.s3d_debug_fused_timeline_only(0 if which == 'attn' else 1) == 0
        fwd()
        torch.cuda.synchronize()
        lib.s3d_debug_fused_timeline_set(ctypes.c_void_p(0))
        t = buf.cpu().numpy().astype(np.int64)
        t = t[t[:, 0] != 0]
        d = np.diff(t[:, :8], axis=1)
        total = t[:, 7] - t[:, 0]
        print(f'{which} ({"cold" if cold else "warm"} caches): {len(t)} workgroups, median total {np.median(total):.0f} cycles '
              f'(p10 {np.percentile(total, 10):.0f}, p90 {np.percentile(total, 90):.0f}); start spread {np.ptp(t[:, 0]):.0f} cycles')
        for i, nm in enumerate(NAMES):
            print(f'    {nm:22s} {np.median(d[:, i]):8.0f}  (p10 {np.percentile(d[:, i], 10):.0f}, p90 {np.percentile(d[:, i], 90):.0f})')
