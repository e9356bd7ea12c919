"""Trained-state golden fixture (run ONLY in the build container, where /root/reference exists):

    python tests/golden/make_golden_trained.py

VERDICT r04 item 6: random-init logits barely depend on the input, so "class indices bit-exact" on the other fixtures says little.  Here the
REFERENCE model (models/vit_3d_2d_pretrain.py Feature3D_ViT2D_V2 + embed_layer_3d_modality.VoxelEmbed, unmodified, on oracle/timm_shim)
is TRAINED with the reference's own optimizer -- torch.optim.Adam(lr=1e-3), train_cls_voxel.py:195 -- for 60 steps on a fixed, learnable
synthetic batch set in cfg-1 geometry (deit_small, 30^3 grid, cell 6, patch 5, 40 classes), the loop of train_cls_voxel.py:277-288
(zero_grad, forward, F.cross_entropy, backward, step).  Stored: the per-step training loss, the logits / argmax / top-2 gap on a held-out
batch after training, and a fingerprint of the initial parameters.  Only numbers travel.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle', 'timm_shim'))
sys.path.insert(1, '/root/reference')

from oracle import voxel_oracle as vo  # noqa: E402
from make_golden import build_reference_model, fingerprint  # noqa: E402

CFG = dict(backbone='deit_small_patch16_224', embed_layer='VoxelEmbed', voxel_size=30, cell=6, patch=5, n_classes=40,
           pos_embedding='default', head='default', batch=16, steps=60, n_batches=4, lr=1e-3,
           labels=[0, 3, 7, 12, 18, 21, 26, 33, 38], held_batch=32)


def batches(cfg):
    return [vo.synthetic_class_batch(cfg['batch'], cfg['voxel_size'], cfg['cell'], cfg['labels'], seed=500 + i) for i in range(cfg['n_batches'])]


def main():
    cfg = CFG
    kw = {k: cfg[k] for k in ('backbone', 'embed_layer', 'voxel_size', 'cell', 'patch', 'n_classes', 'pos_embedding', 'head')}
    sd = vo.init_state_dict(seed=9, exercise_all=False, portable=True, **kw)          # the reference's own initialisation scheme
    model = build_reference_model(cfg)
    model.load_state_dict(sd, strict=True)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=cfg['lr'])                          # train_cls_voxel.py:195
    data = batches(cfg)
    losses = []
    for step in range(cfg['steps']):
        x, y = data[step % len(data)]
        opt.zero_grad()                                                               # train_cls_voxel.py:277
        pred = model(x)
        loss = torch.nn.functional.cross_entropy(pred, y)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    model.eval()
    xh, yh = vo.synthetic_class_batch(cfg['held_batch'], cfg['voxel_size'], cfg['cell'], cfg['labels'], seed=999)
    with torch.no_grad():
        logits = model(xh)
    top2 = logits.topk(2, dim=1).values
    gap = (top2[:, 0] - top2[:, 1]).numpy()
    am = logits.argmax(1)
    distinct = len(set(am.tolist()))
    acc = float((am == yh).float().mean())
    assert distinct >= 6, f'only {distinct} distinct predicted classes: the fixture would not test the argmax criterion'
    out = dict(cfg=np.array(json.dumps(cfg)), fingerprint=fingerprint(sd), losses=np.array(losses), held_logits=logits.numpy(),
               held_argmax=am.numpy(), held_target=yh.numpy(), held_top2_gap=gap)
    np.savez_compressed(os.path.join(HERE, 'trained_cfg1_small_v30_adam60.npz'), **out)
    print(f'loss {losses[0]:.4f} -> {losses[-1]:.4f}; held-out: {distinct} distinct classes, accuracy {acc:.3f}, min gap {gap.min():.4f}, '
          f'{int((gap > 2e-3).sum())}/{len(gap)} decisions above the 2e-3 gap')


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 6: the same loop in a STABLE regime, several reference seeds -- the fixture carries the reference's own
# seed-to-seed spread of held-out accuracy and final loss, which is what the benched (plain-bf16-backward) HIP step is asserted against.
#   * lr 3e-5: the README recipe (Adam, lr 1e-3) runs under pytorch_warmup.UntunedLinearWarmup dampened once per EPOCH
#     (train_cls_voxel.py:197-198,293-294): epoch e trains at 1e-3 (e + 1) / 1999, i.e. at <= 3e-5 for its first 60 epochs;
#   * 12 classes encoded as occupancy 0.05 + 0.015 i (five sigma of a 30^3 grid's density apart: learnable, not trivial), 256 training
#     samples in 16 batches, 400 steps (25 epochs), 256 held-out samples evaluated at six checkpoints (steps 300, 320 .. 400);
#   * twelve seeds of the reference's initialisation scheme (the per-seed outcome varies by +-0.04: the spread needs the samples).
STABLE = dict(backbone='deit_small_patch16_224', embed_layer='VoxelEmbed', voxel_size=30, cell=6, patch=5, n_classes=40,
              pos_embedding='default', head='default', batch=16, steps=400, n_batches=16, lr=3e-5,
              labels=[0, 3, 7, 12, 18, 21, 26, 33, 38, 5, 15, 29], density_base=0.05, density_step=0.015,
              held_batch=256, checkpoints=[300, 320, 340, 360, 380, 400], seeds=list(range(9, 21)), tail=40)


def stable_batches(cfg):
    kw = dict(base=cfg['density_base'], step=cfg['density_step'])
    train = [vo.synthetic_class_batch(cfg['batch'], cfg['voxel_size'], cfg['cell'], cfg['labels'], seed=500 + i, **kw) for i in range(cfg['n_batches'])]
    held = vo.synthetic_class_batch(cfg['held_batch'], cfg['voxel_size'], cfg['cell'], cfg['labels'], seed=999, **kw)
    return train, held


def main_stable():
    cfg = STABLE
    kw = {k: cfg[k] for k in ('backbone', 'embed_layer', 'voxel_size', 'cell', 'patch', 'n_classes', 'pos_embedding', 'head')}
    data, (xh, yh) = stable_batches(cfg)
    out = dict(cfg=np.array(json.dumps(cfg)), held_target=yh.numpy())
    for seed in cfg['seeds']:
        sd = vo.init_state_dict(seed=seed, exercise_all=False, portable=True, **kw)
        model = build_reference_model(cfg)
        model.load_state_dict(sd, strict=True)
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=cfg['lr'])                      # train_cls_voxel.py:195
        losses, accs, ams = [], [], []
        for step in range(cfg['steps']):
            x, y = data[step % len(data)]
            opt.zero_grad()                                                           # train_cls_voxel.py:277-288
            loss = torch.nn.functional.cross_entropy(model(x), y)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
            if step + 1 in cfg['checkpoints']:
                model.eval()                                                          # train_cls_voxel.py:306-329
                with torch.no_grad():
                    am = model(xh).argmax(1)
                model.train()
                ams.append(am.numpy())
                accs.append(float((am == yh).float().mean()))
        out[f'fingerprint_{seed}'] = fingerprint(sd)
        out[f'losses_{seed}'] = np.array(losses)
        out[f'held_acc_{seed}'] = np.array(accs)
        out[f'held_argmax_{seed}'] = np.stack(ams).astype(np.int16)
        print(f'seed {seed}: loss {losses[0]:.4f} -> last-{cfg["tail"]} mean {np.mean(losses[-cfg["tail"]:]):.4f}; held-out accuracy at the checkpoints '
              f'{" ".join(f"{a:.3f}" for a in accs)} (mean {np.mean(accs):.3f})', flush=True)
    np.savez_compressed(os.path.join(HERE, 'trained_stable_cfg1_small_v30_adam400.npz'), **out)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'stable':
        main_stable()
    else:
        main()
