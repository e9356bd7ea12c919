import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import simple3d_former_amd as s3d
from simple3d_former_amd import _lib as L
from oracle import voxel_oracle as vo
KW = dict(backbone='deit_small_patch16_224', embed_layer='VoxelEmbed', voxel_size=32, cell=6, patch=5, n_classes=40)
L.lib().s3d_set_deterministic(1)
sd = vo.init_state_dict(seed=9, exercise_all=True, **KW)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x, y = vo.synthetic_batch(B, 32, 40, seed=11)
x, y = x.cuda(), y.cuda()
res = []
for fill in (False, True):
    eng = s3d.VoxelEngine(device='cuda', lr=1e-3, **KW); eng.load_state_dict(sd); eng.adam_fill = fill
    for step in range(int(sys.argv[2]) if len(sys.argv) > 2 else 1):
        eng.train_step(x, y)
    torch.cuda.synchronize()
    res.append(eng)
    if fill: print(eng.adam_fill_stats, sorted(eng._last_done)[:3] if hasattr(eng, '_last_done') else '')
a, b = res
print('adam_state', a.adam_state.tolist(), b.adam_state.tolist())
for name in ('p', 'm', 'v', 'g', 'hi', 'lo'):
    ta, tb = getattr(a.arena, name), getattr(b.arena, name)
    if not torch.equal(ta, tb):
        bad = (ta != tb).nonzero().flatten()
        print(name, 'differs at', bad.numel(), 'entries; first', int(bad[0]), 'last', int(bad[-1]))
        offs = sorted((o, k) for k, o in a.arena.offsets.items())
        import bisect
        keys = set()
        for idx in bad[:: max(1, bad.numel() // 50)].tolist():
            i = bisect.bisect_right([o for o, _ in offs], idx) - 1
            keys.add(offs[i][1])
        print('   tensors:', sorted(keys)[:20])
    else:
        print(name, 'equal')
