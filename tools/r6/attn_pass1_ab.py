"""Same-process A/B of the FIRST-pass attention of group_embed at the cfg-3 geometry (64 x 196 sequences of 15 tokens, 3 heads, hd = 256, split
forward): the one-tile kernel (attn_fwd_tile256_kernel) against the per-wave kernel it replaces (s3d_debug_knob 8 = 0), bit-compared (tools only).
    python tools/r6/attn_pass1_ab.py [B=64]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simple3d_former_amd import _lib as L, ops
lib = L.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Bb, H, N, hd = B * 196, 3, 15, 256
D = H * hd
g = torch.Generator(device='cuda').manual_seed(3)
qkv = torch.randn(Bb * N, 3 * D, generator=g, device='cuda') * 0.5
hi, lo = ops.split_bf16(qkv); del qkv
dout = torch.randn(Bb * N, D, generator=g, device='cuda').to(torch.bfloat16)

def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

base = None
for knob in (0, -1, 0, -1):
    lib.s3d_debug_knob(8, knob)
    res = {}
    def fwd(): res['o'] = ops.attention_fwd(hi, lo, Bb, H, N, D, N, 1, split=True)
    t = timed(fwd)
    o_hi, o_lo, lse = res['o']
    tb = timed(lambda: ops.attention_bwd(hi, o_hi, o_lo, lse, dout, Bb, H, N, D, N, 1))
    cur = (o_hi.clone(), o_lo.clone(), lse.clone())
    note = ''
    if base is None: base = cur
    else: note = '  vs first: ' + ' '.join(f'{n} {bool(torch.equal(a, b))}' for n, a, b in zip(('out_hi', 'out_lo', 'lse'), cur, base)) + f' max |d out| {float((cur[0].float() + cur[1].float() - base[0].float() - base[1].float()).abs().max()):.2e}'
    gb = (3 * Bb * N * D * 4 + Bb * N * D * 4) / 1e9
    print(f'knob 8 = {knob:2d}: forward {t:7.1f} us ({gb / t * 1e6 / 1e3:.2f} TB/s)  backward {tb:7.1f} us{note}', flush=True)
