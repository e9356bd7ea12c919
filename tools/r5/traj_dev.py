"""Trained-state fixture (reference model + torch.optim.Adam, 60 steps): how far apart do trajectories end that differ by a perturbation?
Split-precision backward with lr * (1 + eps) against the default bf16 backward: last-ten-step loss and held-out accuracy at step 60."""
import json, sys
import numpy as np, torch
sys.path.insert(0, '.')
import simple3d_former_amd as s3d
from oracle import voxel_oracle as vo
z = np.load('tests/golden/trained_cfg1_small_v30_adam60.npz')
cfg = json.loads(str(z['cfg']))
kw = {k: cfg[k] for k in ('backbone', 'embed_layer', 'voxel_size', 'cell', 'patch', 'n_classes', 'pos_embedding', 'head')}
xh, yh = vo.synthetic_class_batch(cfg['held_batch'], cfg['voxel_size'], cfg['cell'], cfg['labels'], seed=999)
print('reference: last-ten loss %.3f, held-out accuracy %.3f' % (float(z['losses'][-10:].mean()), float((z['held_argmax'] == yh.numpy()).mean())))
for precise, eps in ((True, 0.0), (True, 1e-4), (True, -1e-4), (True, 1e-3), (True, -1e-3), (True, 1e-2), (True, -1e-2), (False, 0.0), (False, 0.0)):
    sd = vo.init_state_dict(seed=9, exercise_all=False, portable=True, **kw)
    eng = s3d.VoxelEngine(device='cuda', lr=cfg['lr'] * (1 + eps), precise_backward=precise, **kw)
    eng.load_state_dict(sd)
    data = [vo.synthetic_class_batch(cfg['batch'], cfg['voxel_size'], cfg['cell'], cfg['labels'], seed=500 + i) for i in range(cfg['n_batches'])]
    data = [(x.cuda(), y.cuda()) for x, y in data]
    losses = []
    for step in range(cfg['steps']):
        x, y = data[step % len(data)]
        losses.append(float(eng.train_step(x, y)))
    acc = float((eng.forward(xh.cuda()).cpu().argmax(1) == yh).float().mean())
    dev = np.abs(np.array(losses) - z['losses']) / z['losses']
    print('precise %d lr*(1%+.0e): worst dev %.3f, last-ten loss %.3f, held-out accuracy %.3f' % (precise, eps, dev.max(), np.mean(losses[-10:]), acc))
