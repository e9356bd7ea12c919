"""Run-to-run reproducibility of the seq-first encoder attention forward at the cfg-3 geometry: R launches per variant, every result compared
bit for bit with the first (a race on the K / V rings would show as a few differing rows).  python tools/r6/attn_repro_stress.py [B=64] [R=40]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simple3d_former_amd import _lib as L, ops
lib = L.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = int(sys.argv[2]) if len(sys.argv) > 2 else 40
Bb, H, hd = 15, 4, 192
N, D = B * 196, H * hd
g = torch.Generator().manual_seed(6)
qkv = (torch.randn(Bb * N, 3 * D, generator=g) * 0.5).cuda()
hi, lo = ops.split_bf16(qkv); del qkv
seed = torch.tensor([4321], dtype=torch.int64, device='cuda')
T = (N + 31) // 32
mbuf = torch.zeros(Bb * H * T * T * 32 + 256, dtype=torch.int32, device='cuda')
side = torch.cuda.Stream()
VARIANTS = {'base': ('pipelined, no dropout', -1, 0, False), 'qwait': ('pipelined, wait after the Q loads', -1, 4, False), 'bar': ('pipelined, barrier after the first scores', -1, 8, False),
            'drop': ('pipelined, full split, dropout', -1, 0, True), 'r5': ('round-5 kernel', 2, 0, True), 'p1': ('pipelined, one plane of P, dropout', -1, 1, True)}
for name, knob0, flag, drop in [VARIANTS[v] for v in (sys.argv[3].split(',') if len(sys.argv) > 3 else ['drop', 'base', 'r5'])]:
    lib.s3d_debug_knob(0, knob0)
    kw = dict(drop=(0.1, seed, 0), drop_mask=mbuf) if drop else {}
    first, bad = None, []
    for r in range(R):
        if r % 2:                                  # every other launch with unrelated traffic on a second stream (clock / cache state differs)
            with torch.cuda.stream(side): junk = torch.randn(1 << 26, device='cuda').sum()
        o_hi, o_lo, lse = ops.attention_fwd(hi, lo, Bb, H, N, D, 1, Bb, split=True, p_single_plane=flag, **kw)
        cur = (o_hi, o_lo, lse, mbuf.clone())
        if first is None: first = [t.clone() for t in cur]
        else:
            d = [int((a != b).sum()) for a, b in zip(cur, first)]
            if any(d):
                dl = (lse - first[2]).abs().view(-1)
                rows = torch.nonzero(dl > 0).view(-1)
                do = ((o_hi.float() + o_lo.float()) - (first[0].float() + first[1].float())).abs()
                bad.append((r, d, f'max |d lse| {float(dl.max()):.2e} (lse ~ {float(first[2].abs().mean()):.2f}), max |d out| {float(do.max()):.2e}, '
                               f'lse rows (bh, q): {[(int(i) // N, int(i) % N) for i in rows[:3]]} .. {(int(rows[-1]) // N, int(rows[-1]) % N) if len(rows) else None}'))
    torch.cuda.synchronize()
    print(f'{name}: {len(bad)} of {R - 1} repeats differ from the first' + (''.join(f'\n     repeat {b[0]}: {b[1]} (out_hi, out_lo, lse, mask words) {b[2]}' for b in bad[:4])), flush=True)
