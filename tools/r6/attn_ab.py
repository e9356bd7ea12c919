"""Same-process A/B of the seq-first encoder attention (cfg-3: 15 x 4 heads, N = 196 B keys, hd = 192, dropout 0.1 with the stored mask):
times forward / backward per s3d_debug_knob setting and checks the variants against each other bit for bit (tools only).
    python tools/r6/attn_ab.py [B=64] [knob_id:value,...;...]     (id 100 = S3dAttnArgs::p_single_plane)"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simple3d_former_amd import _lib as L, ops
lib = L.lib()
DEV = 'cuda'

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    settings = [dict()] + [dict((int(kv.split(':')[0]), int(kv.split(':')[1])) for kv in grp.split(',')) for grp in (sys.argv[2].split(';') if len(sys.argv) > 2 else [])]
    Bb, H, hd = 15, 4, 192
    N, D = B * 196, H * hd
    rows = Bb * N
    g = torch.Generator().manual_seed(6)
    qkv = (torch.randn(rows, 3 * D, generator=g) * 0.5).to(DEV)
    hi, lo = ops.split_bf16(qkv)
    del qkv
    dout = torch.randn(rows, D, generator=g).to(DEV).to(torch.bfloat16)
    seed = torch.tensor([4321], dtype=torch.int64, device=DEV)
    T = (N + 31) // 32
    mbuf = torch.zeros(Bb * H * T * T * 32 + 256, dtype=torch.int32, device=DEV)
    drop = (0.1, seed, 0)
    base = None
    for st in settings:
        for k in range(16): lib.s3d_debug_knob(k, -1)
        for k, v in st.items():
            if k < 16: lib.s3d_debug_knob(k, v)
        res = {}
        def fwd():
            res['o'] = ops.attention_fwd(hi, lo, Bb, H, N, D, 1, Bb, split=True, drop=drop, drop_mask=mbuf, p_single_plane=st.get(100, 0))
        t_f = timed(fwd)
        out_hi, out_lo, lse = res['o']
        def bwd():
            res['d'] = ops.attention_bwd(hi, out_hi, out_lo, lse, dout, Bb, H, N, D, 1, Bb, drop=drop, drop_mask=mbuf)
        t_b = timed(bwd)
        cur = (out_hi.clone(), out_lo.clone(), lse.clone(), res['d'].clone())
        note = ''
        if base is None:
            base = cur
        else:
            same = [bool(torch.equal(a, b)) for a, b in zip(cur, base)]
            dmax = float((cur[3].float() - base[3].float()).abs().max())
            dl = (cur[2] - base[2]).abs()
            note = f'  [lse: {int((dl > 0).sum())} of {dl.numel()} differ, max {float(dl.max()):.2e}]  vs shipped: out_hi {same[0]} out_lo {same[1]} lse {same[2]} dqkv {same[3]} (max |d dqkv| {dmax:.2e})'
        print(f'knobs {st or "shipped"}: forward {t_f:8.3f} ms  backward {t_b:8.3f} ms{note}', flush=True)

if __name__ == '__main__':
    main()
