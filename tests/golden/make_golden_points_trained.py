"""Trained-state golden fixture for the point path (run ONLY in the build container, where /root/reference exists):

    python tests/golden/make_golden_points_trained.py

The reference's PointTransformerCls (models/3DViT/model.py, unmodified, on oracle/timm_shim; train mode: BatchNorm batch statistics) trained
with the reference's own optimizer -- torch.optim.SGD(lr = 0.01, momentum = 0.9), train_cls.py:91, the loop of train_cls.py:117-123 -- on a fixed,
learnable synthetic batch set (oracle.point_oracle.synthetic_class_points), torch.randint patched to hand out the recorded FPS start indices.
Stored: the per-step training loss, the eval-mode logits / argmax / top-2 gap on a held-out batch after training.  Only numbers travel."""
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

from oracle import point_oracle as po  # noqa: E402
from tests.golden.make_golden_points import build_reference  # noqa: E402

CFG = dict(task='cls', backbone='deit_tiny_patch16_224', n_points=64, d_points=6, n_classes=40, batch=8, steps=80, n_batches=4,
           lr=0.01, momentum=0.9, labels=[1, 5, 9, 14, 22, 27, 31, 36], held_batch=32)


def main():
    cfg = CFG
    sd = po.init_state_dict(backbone=cfg['backbone'], n_classes=cfg['n_classes'], d_points=cfg['d_points'], seed=9)
    model = build_reference(cfg)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    data = [po.synthetic_class_points(cfg['batch'], cfg['n_points'], cfg['labels'], seed=600 + i) for i in range(cfg['n_batches'])]
    held = po.synthetic_class_points(cfg['held_batch'], cfg['n_points'], cfg['labels'], seed=999)
    opt = torch.optim.SGD(model.parameters(), lr=cfg['lr'], momentum=cfg['momentum'])      # train_cls.py:91
    queue, orig = [], torch.randint
    torch.randint = lambda *a, **k: queue.pop(0).clone()
    losses = []
    try:
        model.train()
        for step in range(cfg['steps']):
            x, y, starts = data[step % len(data)]
            queue[:] = list(starts)
            opt.zero_grad()
            loss = torch.nn.functional.cross_entropy(model(x), y)                             # train_cls.py:119-121
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        model.eval()
        queue[:] = list(held[2])
        with torch.no_grad():
            logits = model(held[0])
    finally:
        torch.randint = orig
    top2 = logits.topk(2, dim=1).values
    gap = (top2[:, 0] - top2[:, 1]).numpy()
    am = logits.argmax(1)
    distinct, acc = len(set(am.tolist())), float((am == held[1]).float().mean())
    print(f'loss {losses[0]:.4f} -> {losses[-1]:.4f}; held-out: {distinct} distinct classes, accuracy {acc:.3f}, min gap {gap.min():.4f}, '
          f'{int((gap > 2e-3).sum())}/{len(gap)} decisions above the 2e-3 gap')
    assert distinct >= 5, f'only {distinct} distinct predicted classes'
    out = dict(cfg=np.array(json.dumps(cfg)), losses=np.array(losses), held_logits=logits.numpy(), held_argmax=am.numpy(),
               held_target=held[1].numpy(), held_top2_gap=gap)
    np.savez_compressed(os.path.join(HERE, 'trained_pts_cls_tiny_n64_sgd80.npz'), **out)


if __name__ == '__main__':
    main()
