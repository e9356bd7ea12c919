"""per-step deviation of the HIP fused step from the trained-state fixture (reference model + torch.optim.Adam), default and deterministic mode"""
import json, sys
import numpy as np, torch
sys.path.insert(0, '.')
import simple3d_former_amd as s3d
from simple3d_former_amd import _lib as L
from oracle import voxel_oracle as vo
z = np.load('tests/golden/trained_cfg1_small_v30_adam60.npz')
cfg = json.loads(str(z['cfg']))
kw = {k: cfg[k] for k in ('backbone', 'embed_layer', 'voxel_size', 'cell', 'patch', 'n_classes', 'pos_embedding', 'head')}
for det, precise in ((0, True), (1, True), (0, False)):
    L.lib().s3d_set_deterministic(det)
    sd = vo.init_state_dict(seed=9, exercise_all=False, portable=True, **kw)
    eng = s3d.VoxelEngine(device='cuda', lr=cfg['lr'], precise_backward=precise, **kw)
    eng.load_state_dict(sd)
    data = [vo.synthetic_class_batch(cfg['batch'], cfg['voxel_size'], cfg['cell'], cfg['labels'], seed=500 + i) for i in range(cfg['n_batches'])]
    data = [(x.cuda(), y.cuda()) for x, y in data]
    dev = []
    for step in range(cfg['steps']):
        x, y = data[step % len(data)]
        loss = float(eng.train_step(x, y))
        dev.append((loss - float(z['losses'][step])) / float(z['losses'][step]))
    print('det', det, 'precise', precise, ' '.join(f'{d * 100:+.2f}' for d in dev))
L.lib().s3d_set_deterministic(0)
