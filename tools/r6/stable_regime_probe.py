"""Probe (build container only): which synthetic task / Adam regime gives the reference model an informative held-out accuracy."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle', 'timm_shim')); sys.path.insert(1, '/root/reference')
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from oracle import voxel_oracle as vo
from make_golden import build_reference_model

def task(batch, V, labels, seed, step, base):
    labels = list(labels)
    pick = (vo.portable_uniform((batch,), seed, 2002) * len(labels)).long().clamp_(max=len(labels) - 1)
    y = torch.tensor(labels, dtype=torch.long)[pick]
    u = vo.portable_uniform((batch, 1, V, V, V), seed, 2001)
    dens = (base + step * pick.double()).view(batch, 1, 1, 1, 1)
    return (u < dens).to(torch.int32).float(), y

def run(lr, steps, nb, B, dstep, base, labels, init_seed=9, warm=0):
    cfg = dict(backbone='deit_small_patch16_224', embed_layer='VoxelEmbed', voxel_size=30, cell=6, patch=5, n_classes=40, pos_embedding='default', head='default')
    sd = vo.init_state_dict(seed=init_seed, exercise_all=False, portable=True, **cfg)
    model = build_reference_model(cfg); model.load_state_dict(sd, strict=True); model.train()
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    data = [task(B, 30, labels, 500 + i, dstep, base) for i in range(nb)]
    xh, yh = task(256, 30, labels, 999, dstep, base)
    t0 = time.time(); losses = []
    for s in range(steps):
        if warm: 
            for g in opt.param_groups: g['lr'] = lr * min(1.0, (s + 1) / warm)
        x, y = data[s % nb]
        opt.zero_grad(); loss = torch.nn.functional.cross_entropy(model(x), y); loss.backward(); opt.step(); losses.append(float(loss))
        if (s + 1) % 20 == 0:
            model.eval()
            with torch.no_grad(): lg = model(xh)
            model.train()
            print(f'  step {s+1}: loss(last10) {np.mean(losses[-10:]):.4f} held acc {float((lg.argmax(1)==yh).float().mean()):.3f} distinct {len(set(lg.argmax(1).tolist()))} t {time.time()-t0:.0f}s', flush=True)
    return losses

if __name__ == '__main__':
    torch.set_num_threads(8)
    lr, steps, nb, B, dstep, base = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), float(sys.argv[6])
    nl = int(sys.argv[7]); warm = int(sys.argv[8]) if len(sys.argv) > 8 else 0
    labels = [0, 3, 7, 12, 18, 21, 26, 33, 38, 5, 15, 29][:nl]
    run(lr, steps, nb, B, dstep, base, labels, warm=warm)
