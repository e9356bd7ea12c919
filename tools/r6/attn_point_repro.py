"""Bitwise repeatability of the point path's long-sequence attention kernels (hd = 64; cfg-4: 128 x 3 heads x 257 tokens, cfg-5: 32 x 3 x 513) at the
kernel level, forward and backward, R launches each -- the model-level audit (tools/r6/repro_audit.py) cannot see a race in the point path's
backward behind its atomics (tools only).   python tools/r6/attn_point_repro.py [R=200]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simple3d_former_amd import ops
R = int(sys.argv[1]) if len(sys.argv) > 1 else 200
side = torch.cuda.Stream()
for name, Bb, H, N, hd in (('cfg-4 geometry', 128, 3, 257, 64), ('cfg-5 geometry', 32, 3, 513, 64), ('250 tokens (8 tiles, four waves)', 16, 3, 250, 64)):
    D = H * hd
    g = torch.Generator(device='cuda').manual_seed(5)
    qkv = torch.randn(Bb * N, 3 * D, generator=g, device='cuda')
    hi, lo = ops.split_bf16(qkv)
    dout = torch.randn(Bb * N, D, generator=g, device='cuda').to(torch.bfloat16)
    first, bad_f, bad_b = None, 0, 0
    for r in range(R):
        if r % 3 == 1:
            with torch.cuda.stream(side): torch.randn(1 << 22, device='cuda').sum()
        o_hi, o_lo, lse = ops.attention_fwd(hi, lo, Bb, H, N, D, N, 1, split=True)
        dqkv = ops.attention_bwd(hi, o_hi, o_lo, lse, dout, Bb, H, N, D, N, 1)
        cur = (o_hi, o_lo, lse, dqkv)
        if first is None: first = [t.clone() for t in cur]
        else:
            bad_f += int(not all(torch.equal(a, b) for a, b in zip(cur[:3], first[:3])))
            bad_b += int(not torch.equal(cur[3], first[3]))
    torch.cuda.synchronize()
    print(f'{name} (Bb {Bb}, H {H}, N {N}, hd {hd}): forward {bad_f} of {R - 1} repeats differ, backward {bad_b} of {R - 1}', flush=True)
