"""Run-to-run reproducibility audit of the four benched configurations at FULL size (tools only; found nothing after the attention forward's
prologue race was fixed -- profiles/r06_repro_audit.txt): (a) the forward R times, logits compared bit for bit with the first (the forward has
no atomics), (b) in deterministic mode (s3d_set_deterministic: no split-K atomics) forward + backward R times, the whole gradient arena bit for bit.
A race shows as a sporadic mismatch; unrelated traffic on a second stream varies the timing between repeats.
    python tools/r6/repro_audit.py [cfg2,cfg3,cfg4,cfg5] [R=30]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import simple3d_former_amd as s3d
from simple3d_former_amd import _lib as L
from simple3d_former_amd.point_engine import PointEngine
from oracle import voxel_oracle as vo, point_oracle as po
which = (sys.argv[1] if len(sys.argv) > 1 else 'cfg2,cfg3,cfg4,cfg5').split(',')
R = int(sys.argv[2]) if len(sys.argv) > 2 else 30
DEV = 'cuda'
side = torch.cuda.Stream()
lib = L.lib()

def noise(r):
    if r % 3 == 1:
        with torch.cuda.stream(side): torch.randn(1 << 24, device=DEV).sum()

def audit(name, fwd, fwd_bwd, grads, reps):
    t0 = time.time()
    first, bad = None, 0
    for r in range(reps):
        noise(r)
        out = fwd().clone()
        if first is None: first = out
        elif not torch.equal(out, first): bad += 1
    print(f'{name}: forward, {bad} of {reps - 1} repeats differ from the first', flush=True)
    lib.s3d_set_deterministic(1)
    try:
        first, bad, worst = None, 0, 0.0
        for r in range(reps):
            noise(r)
            fwd_bwd()
            g = grads().clone()
            if first is None: first = g
            elif not torch.equal(g, first):
                bad += 1; worst = max(worst, float((g - first).abs().max() / first.abs().max()))
        print(f'{name}: deterministic forward + backward, {bad} of {reps - 1} repeats differ from the first' + (f' (largest difference {worst:.2e} of the largest gradient)' if bad else '') + f'   [{time.time() - t0:.0f} s]', flush=True)
    finally:
        lib.s3d_set_deterministic(0)

for c in which:
    if c in ('cfg2', 'cfg3'):
        kw = dict(backbone='deit_small_patch16_224', embed_layer='VoxelEmbed', voxel_size=32, cell=6, patch=5, n_classes=40) if c == 'cfg2' else \
             dict(backbone='deit_base_patch16_224', embed_layer='VoxelEmbed_no_average', voxel_size=128, cell=9, patch=14, n_classes=55)
        pe = 'default' if c == 'cfg2' else 'group_embed'
        sd = vo.init_state_dict(seed=9, pos_embedding=pe, **kw)
        x, y = vo.synthetic_batch(64, kw['voxel_size'], kw['n_classes'], seed=9)
        eng = s3d.VoxelEngine(device=DEV, pos_embedding=pe, **kw); eng.load_state_dict(sd)
        xd, yd = x.to(DEV), y.to(DEV)
        if c == 'cfg3': eng.set_dropout(0.1, seed=3)          # same masks every repeat: the seed is not advanced by forward()
        def fb():
            eng.forward(xd); eng.cross_entropy(64, yd); eng.zero_grad(); eng.backward(64)
        audit(c, lambda: eng.forward(xd), fb, lambda: eng.arena.g, R * (6 if c == 'cfg2' else 1))
        del eng
    else:
        task, npts, dpts, ncls, B = ('cls', 1024, 6, 40, 128) if c == 'cfg4' else ('seg', 2048, 22, 50, 32)
        sd = po.init_state_dict(backbone='deit_tiny_patch16_224', n_classes=ncls, d_points=dpts, seed=9)
        x, y, starts = po.synthetic_points(B, npts, dpts, ncls, task, seed=9)
        eng = PointEngine(backbone='deit_tiny_patch16_224', n_points=npts, d_points=dpts, n_classes=ncls, task=task, device=DEV); eng.load_state_dict(sd)
        xd, yd, sts = x.to(DEV), y.to(DEV), tuple(s.to(DEV) for s in starts)
        def fb():
            eng.forward(xd, sts); eng.cross_entropy(B, yd); eng.zero_grad(); eng.backward(B)
        audit(c + ' (eval-mode forward; train-mode backward)', lambda: eng.forward(xd, sts, training=False), fb, lambda: eng.arena.g, R)
        del eng
    torch.cuda.empty_cache()
