"""Golden-vector generator for the point-cloud path (run ONLY in the build container, where /root/reference exists):

    python tests/golden/make_golden_points.py

Runs the reference's unmodified models/3DViT/model.py (PointTransformerCls / PointTransformerSeg), the PointTransformerSeg of
models/3DViT_1_layer, models/3DViT_0_layer and models/3DViT_LWF (incl. forward_images and, for the LWF case, the
train_partseg_lwf.py:207-228 loss CE(points) + lambda * CE(images) in ONE backward) and data/pointnet_util.py on CPU
in TRAIN mode (BatchNorm batch statistics), on top of oracle/timm_shim, with the `data` package stubbed (data/__init__.py imports modules that do not exist) and torch.randint patched to hand out recorded FPS
start indices.  Stores logits, loss, gradient summaries, updated BatchNorm running statistics, eval-mode logits."""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle', 'timm_shim'))
sys.path.insert(1, '/root/reference')

from oracle import point_oracle as po  # noqa: E402
from tests.golden.make_golden import sample_idx  # noqa: E402

CASES = {
    'pts_cls_tiny_n64_b3': dict(task='cls', backbone='deit_tiny_patch16_224', n_points=64, d_points=6, n_classes=40, batch=3),
    'pts_seg_tiny_n64_b2': dict(task='seg', backbone='deit_tiny_patch16_224', n_points=64, d_points=22, n_classes=50, batch=2),
    'pts_cls_tiny_n1024_b2': dict(task='cls', backbone='deit_tiny_patch16_224', n_points=1024, d_points=6, n_classes=40, batch=2),
    'pts_seg_tiny_n2048_b1': dict(task='seg', backbone='deit_tiny_patch16_224', n_points=2048, d_points=22, n_classes=50, batch=1),
    # the other model directories (config/model/3DViT_{1_layer,0_layer,lwf}.yaml); seg only
    'pts_seg1_tiny_n64_b2': dict(task='seg', variant='3DViT_1_layer', backbone='deit_tiny_patch16_224', n_points=64, d_points=22,
                                 n_classes=50, batch=2),
    'pts_seg1_small_n256_b2': dict(task='seg', variant='3DViT_1_layer', backbone='deit_small_patch16_224', n_points=256, d_points=22,
                                   n_classes=50, batch=2),
    'pts_seg0_tiny_n64_b2': dict(task='seg', variant='3DViT_0_layer', backbone='deit_tiny_patch16_224', n_points=64, d_points=22,
                                 n_classes=50, batch=2),
    # cfg.model.head == 'AMSoftmax' (models/3DViT/model.py:427-428): AMSoftmaxLayer as the per-point head
    'pts_seg_tiny_n64_am_b2': dict(task='seg', backbone='deit_tiny_patch16_224', n_points=64, d_points=22, n_classes=50, batch=2,
                                   head='AMSoftmax'),
    'pts_seglwf_tiny_n64_b2': dict(task='seg', variant='3DViT_LWF', backbone='deit_tiny_patch16_224', n_points=64, d_points=22,
                                   n_classes=50, batch=2, lwf=True, lambda_weight=0.1),
}


def build_reference(cfg):
    stub = types.ModuleType('data')
    stub.__path__ = ['/root/reference/data']
    sys.modules['data'] = stub
    variant = cfg.get('variant', '3DViT')
    mod = importlib.import_module(f'models.{variant}.model')
    c = types.SimpleNamespace(num_point=cfg['n_points'], num_class=cfg['n_classes'], input_dim=cfg['d_points'],
                              model=types.SimpleNamespace(nblocks=4, nneighbor=16, transformer_dim=512, head=cfg.get('head', 'default'),
                                                          transformer_backbone=cfg['backbone'], pretrained=False, name=variant))
    return getattr(mod, 'PointTransformerCls' if cfg['task'] == 'cls' else 'PointTransformerSeg')(c)


def run_case(name, cfg):
    variant = cfg.get('variant', '3DViT')
    vv = po.VARIANTS[variant]
    sd = po.init_state_dict(backbone=cfg['backbone'], n_classes=cfg['n_classes'], d_points=cfg['d_points'], seed=9, variant=variant,
                            head=cfg.get('head', 'default'))
    model = build_reference(cfg)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected                      # every generated key exists in the reference (name + shape contract)
    if vv['image']:
        assert all('last_pos_embed' in k for k in missing), missing
    else:
        assert all(k.startswith(('pos_embed', 'patch_embed.')) or 'last_pos_embed' in k for k in missing), missing
    x, y, starts = po.synthetic_points(cfg['batch'], cfg['n_points'], cfg['d_points'], cfg['n_classes'], cfg['task'], seed=9,
                                       variant=variant)
    img = yi = None
    if vv['image']:
        img = (po.vo.portable_uniform((cfg['batch'], 3, 224, 224), 9, 7001) * 2 - 1).float()
        yi = (po.vo.portable_uniform((cfg['batch'],), 9, 7002) * 1000).long()
    queue = []
    orig = torch.randint

    def fake_randint(*a, **k):
        return queue.pop(0).clone()

    out = dict(cfg=np.array(json.dumps(cfg)))
    for i, st in enumerate(starts):
        out[f'start{i}'] = st.numpy()
    torch.randint = fake_randint
    try:
        model.train()
        queue[:] = list(starts)
        logits = model(x)
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, cfg['n_classes']), y.reshape(-1))
        if img is not None:
            img_pred = model.forward_images(img)
            out.update(img_logits=img_pred.detach().numpy(), img_target=yi.numpy(), img_argmax=img_pred.argmax(1).numpy())
            t2 = img_pred.detach().topk(2, dim=1).values
            out['img_top2_gap'] = (t2[:, 0] - t2[:, 1]).numpy()
            if cfg.get('lwf'):
                out['loss_points'] = np.array(loss.item())
                loss_i = torch.nn.functional.cross_entropy(img_pred, yi)
                out['loss_image'] = np.array(loss_i.item())
                loss = loss + cfg['lambda_weight'] * loss_i
        loss.backward()
        model.eval()
        queue[:] = list(starts)
        with torch.no_grad():
            out['logits_eval'] = model(x).numpy()
    finally:
        torch.randint = orig
    out.update(logits=logits.detach().numpy(), loss=np.array(loss.item()), target=y.numpy(),
               argmax=logits.detach().argmax(-1).numpy())
    top2 = logits.detach().topk(2, dim=-1).values
    out['top2_gap'] = (top2[..., 0] - top2[..., 1]).numpy()
    for k, v in model.state_dict().items():
        if 'running_' in k and 'last_pos' not in k:
            out['stat/' + k] = v.numpy()
    names = []
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().flatten()
        names.append(k)
        idx = sample_idx(g.numel())
        out['gnorm/' + k] = np.array(float(g.double().norm())); out['gsum/' + k] = np.array(float(g.double().sum()))
        out['gidx/' + k] = idx; out['gval/' + k] = g[idx].numpy()
        if g.numel() <= 4096:
            out['gfull/' + k] = p.grad.detach().numpy()
    out['grad_names'] = np.array(json.dumps(names))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(f'{name}: loss {loss.item():.6f} logits {tuple(logits.shape)} gap_min {out["top2_gap"].min():.4f} {len(names)} grads')


def write_state_dict_keys():
    """{model: {state_dict key: shape}} of the reference modules (deit_tiny, cfg-4 / cfg-5 shapes) -> point_state_dict_keys.json:
    the checkpoint-compatibility contract tests/test_host_cpu.py holds the drop-in modules to."""
    out = {}
    base = dict(backbone='deit_tiny_patch16_224')
    table = {'cls': dict(task='cls', n_points=1024, d_points=6, n_classes=40),
             'seg': dict(task='seg', n_points=2048, d_points=22, n_classes=50)}
    for v in ('3DViT_1_layer', '3DViT_0_layer', '3DViT_LWF'):
        table[v] = dict(task='seg', variant=v, n_points=2048, d_points=22, n_classes=50)
    for name, c in table.items():
        model = build_reference({**base, **c})
        out[name] = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(HERE, 'point_state_dict_keys.json'), 'w') as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print({k: len(v) for k, v in out.items()})


if __name__ == '__main__':
    torch.set_num_threads(8)
    if sys.argv[1:] == ['keys']:
        write_state_dict_keys()
        sys.exit(0)
    for n, c in CASES.items():
        if len(sys.argv) == 1 or n in sys.argv[1:]:
            run_case(n, c)
