"""Where does the shipped backward's gradient differ from the split-precision one in a TRAINED state?  (tools only)
Trains the stable-regime fixture's seed-9 model for N steps with precise_backward, then computes the gradient of one batch with every
backward variant and prints, per parameter group: relative rms error, regression coefficient, and the share of elements whose error exceeds
half their own magnitude (what Adam's per-element normalisation turns into a random step)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import simple3d_former_amd as s3d
from simple3d_former_amd import engine as E, _lib as L
from oracle import voxel_oracle as vo
DEV = 'cuda'

def groups(eng):
    out = {}
    for name in eng.arena.offsets:
        if name.startswith('blocks.'):
            key = name.split('.', 2)[2]
        else:
            key = name
        out.setdefault(key, []).append(name)
    return out

def grad_of(eng, x, y):
    eng.zero_grad()
    eng.forward_loss(x, y)
    eng.backward(x.shape[0])
    torch.cuda.synchronize()
    return {n: eng.arena.grad(n).detach().clone().double() for n in eng.arena.offsets}

def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'trained_stable_cfg1_small_v30_adam400.npz'))
    cfg = json.loads(str(z['cfg']))
    kw = {k: cfg[k] for k in ('backbone', 'embed_layer', 'voxel_size', 'cell', 'patch', 'n_classes', 'pos_embedding', 'head')}
    dk = dict(base=cfg['density_base'], step=cfg['density_step'])
    data = [vo.synthetic_class_batch(cfg['batch'], cfg['voxel_size'], cfg['cell'], cfg['labels'], seed=500 + i, **dk) for i in range(cfg['n_batches'])]
    data = [(x.to(DEV), y.to(DEV)) for x, y in data]
    sd = vo.init_state_dict(seed=9, exercise_all=False, portable=True, **kw)
    ref = s3d.VoxelEngine(device=DEV, lr=cfg['lr'], precise_backward=True, **kw)
    ref.load_state_dict(sd)
    for s in range(steps):
        ref.train_step(*data[s % len(data)])
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    x, y = data[3]
    gref = grad_of(ref, x, y)
    lib = L.lib()
    variants = [('default', {}, 0), ('deterministic', {}, 1), ('LN_BWD_FUSE=0', dict(LN_BWD_FUSE=False), 0), ('WGRAD_GROUP=0', dict(WGRAD_GROUP=0), 0),
                ('FUSED_BWD=0', dict(FUSED_BWD=False), 0), ('FUSED_BLOCKS=0', dict(FUSED_BLOCKS=False), 0), ('FUSE_LOSS_END=0', dict(FUSE_LOSS_END=False), 0)]
    saved = {k: getattr(E, k) for k in ('LN_BWD_FUSE', 'WGRAD_GROUP', 'FUSED_BWD', 'FUSED_BLOCKS', 'FUSE_LOSS_END')}
    for name, patch, det in variants:
        for k, v in saved.items():
            setattr(E, k, v)
        for k, v in patch.items():
            setattr(E, k, v)
        lib.s3d_set_deterministic(det)
        eng = s3d.VoxelEngine(device=DEV, lr=cfg['lr'], **kw)
        eng.load_state_dict(sd)
        g = grad_of(eng, x, y)
        lib.s3d_set_deterministic(0)
        print(f'--- {name}')
        for key, names in groups(eng).items():
            a = torch.cat([g[n].flatten() for n in names]); b = torch.cat([gref[n].flatten() for n in names])
            if float(b.norm()) == 0:
                continue
            rel = float((a - b).norm() / b.norm())
            coef = float((a * b).sum() / (b * b).sum())
            bad = float(((a - b).abs() > 0.5 * b.abs()).double().mean())
            worst = max(float((g[n] - gref[n]).norm() / gref[n].norm().clamp_min(1e-30)) for n in names)
            print(f'   {key:34s} rel rms {rel:.2e}  coef {coef:.5f}  |err| > |g|/2: {bad:.3f}   worst block {worst:.2e}')

if __name__ == '__main__':
    main()
