"""Golden-vector generator (run ONLY in the build container, where /root/reference exists):

    python tests/golden/make_golden.py

Imports the reference's own, unmodified files (models/embed_layer_3d_modality.py,
models/vit_3d_2d_pretrain.py) on top of oracle/timm_shim, loads a deterministic parameter set
(oracle.voxel_oracle.init_state_dict, strict=True so the state_dict key/shape contract is
checked too), runs the REFERENCE forward/backward on CPU in fp32 and stores inputs' recipe +
reference outputs as small .npz fixtures.  No reference source travels: only numbers.

Fixture content per case: the case config (json), a fingerprint of the regenerated parameters
(so a replay on another box can prove it rebuilt the same weights), logits, loss, argmax,
top-2 logit gap, and per-parameter gradient summaries (norm, sum, 96 sampled entries; full
tensor when <= 4096 elements).
"""
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

from oracle import voxel_oracle as vo  # noqa: E402

CASES = {
    # name: cfg  (cfg-1 / cfg-2 real geometry first)
    'cfg1_small_v30_b8': dict(backbone='deit_small_patch16_224', embed_layer='VoxelEmbed', voxel_size=30,
                              cell=6, patch=5, n_classes=40, pos_embedding='default', head='default', batch=8),
    'cfg2_small_v32_b4': dict(backbone='deit_small_patch16_224', embed_layer='VoxelEmbed', voxel_size=32,
                              cell=6, patch=5, n_classes=40, pos_embedding='default', head='default', batch=4),
    'tiny_v12_default_b3': dict(backbone='deit_tiny_patch16_224', embed_layer='VoxelEmbed', voxel_size=12,
                                cell=4, patch=3, n_classes=10, pos_embedding='default', head='default', batch=3),
    # NOTE: pos_embedding='no_embed' is broken in the reference as shipped (voxel_pos_embed is never
    # created because patch_embed.num_patches == 196 always; forward raises AttributeError) -> no golden.
    'tiny_v12_noavg_default_b2': dict(backbone='deit_tiny_patch16_224', embed_layer='VoxelEmbed_no_average',
                                      voxel_size=12, cell=4, patch=3, n_classes=10, pos_embedding='default',
                                      head='default', batch=2),
    'tiny_v12_naive_b2': dict(backbone='deit_tiny_patch16_224', embed_layer='VoxelNaiveProjection',
                              voxel_size=12, cell=4, patch=3, n_classes=10, pos_embedding='default',
                              head='default', batch=2),
    'small_v30_amsoftmax_b4': dict(backbone='deit_small_patch16_224', embed_layer='VoxelEmbed', voxel_size=30,
                                   cell=6, patch=5, n_classes=40, pos_embedding='default', head='AMSoftmax', batch=4),
    'tiny_v12_group_b3': dict(backbone='deit_tiny_patch16_224', embed_layer='VoxelEmbed_no_average',
                              voxel_size=12, cell=4, patch=3, n_classes=10, pos_embedding='group_embed',
                              head='default', batch=3),
    'cfg3_base_v128_group_b1': dict(backbone='deit_base_patch16_224', embed_layer='VoxelEmbed_no_average',
                                    voxel_size=128, cell=9, patch=14, n_classes=55, pos_embedding='group_embed',
                                    head='default', batch=1),
}


def fingerprint(sd):
    keys = sorted(sd)
    return np.array([[float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in keys])


def sample_idx(n, count=96):
    if n <= count:
        return np.arange(n)
    return np.unique(np.linspace(0, n - 1, count).astype(np.int64))


def build_reference_model(cfg):
    from models import embed_layer_3d_modality as ref_embed
    from models.vit_3d_2d_pretrain import Feature3D_ViT2D_V2
    D = vo.BACKBONES[cfg['backbone']]['embed_dim']
    layer = getattr(ref_embed, cfg['embed_layer'])(voxel_size=cfg['voxel_size'], cell_size=cfg['cell'],
                                                   patch_size=cfg['patch'], embed_dim=D)
    return Feature3D_ViT2D_V2(embed_layer=layer, n_classes=cfg['n_classes'],
                              transformer_backbone=cfg['backbone'], pretrained=False,
                              pos_embedding=cfg['pos_embedding'], head=cfg['head'])


def run_case(name, cfg):
    kw = {k: cfg[k] for k in ('backbone', 'embed_layer', 'voxel_size', 'cell', 'patch', 'n_classes',
                              'pos_embedding', 'head')}
    sd = vo.init_state_dict(seed=9, exercise_all=True, portable=True, **kw)
    model = build_reference_model(cfg)
    model.load_state_dict(sd, strict=True)           # key names + shapes are part of the contract
    model.eval()                                     # only matters for group_embed's dropout(0.1)
    x, y = vo.synthetic_batch(cfg['batch'], cfg['voxel_size'], cfg['n_classes'], seed=9, portable=True)
    logits = model(x)
    loss = torch.nn.functional.cross_entropy(logits, y)
    loss.backward()
    out = dict(cfg=np.array(json.dumps(cfg)), fingerprint=fingerprint(sd),
               logits=logits.detach().numpy(), loss=np.array(loss.item()),
               argmax=logits.argmax(1).numpy(), target=y.numpy())
    top2 = logits.detach().topk(2, dim=1).values
    out['top2_gap'] = (top2[:, 0] - top2[:, 1]).numpy()
    names = []
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().flatten()
        names.append(k)
        idx = sample_idx(g.numel())
        out['gnorm/' + k] = np.array(float(g.double().norm()))
        out['gsum/' + k] = np.array(float(g.double().sum()))
        out['gidx/' + k] = idx
        out['gval/' + k] = g[idx].numpy()
        if g.numel() <= 4096:
            out['gfull/' + k] = p.grad.detach().numpy()
    out['grad_names'] = np.array(json.dumps(names))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(f'{name}: loss {loss.item():.6f} argmax {logits.argmax(1).tolist()} gap_min {out["top2_gap"].min():.4f} '
          f'{len(names)} grads')


def lwf_case(name='tiny_v12_lwf_b2'):
    """forward_images + the LwF loss of train_cls_voxel.py:250-267 (teacher labels are just seeded integers here)."""
    cfg = dict(backbone='deit_tiny_patch16_224', embed_layer='VoxelEmbed', voxel_size=12, cell=4, patch=3, n_classes=10,
               pos_embedding='default', head='default', batch=2, lambda_weight=0.1)
    kw = {k: cfg[k] for k in ('backbone', 'embed_layer', 'voxel_size', 'cell', 'patch', 'n_classes', 'pos_embedding', 'head')}
    sd = vo.init_state_dict(seed=9, exercise_all=True, portable=True, **kw)
    model = build_reference_model(cfg)
    model.load_state_dict(sd, strict=True)
    model.eval()
    x, y = vo.synthetic_batch(cfg['batch'], cfg['voxel_size'], cfg['n_classes'], seed=9, portable=True)
    img = (vo.portable_uniform((cfg['batch'], 3, 224, 224), 9, 7001) * 2 - 1).float()
    yi = (vo.portable_uniform((cfg['batch'],), 9, 7002) * 1000).long()
    pred = model(x)
    img_pred = model.forward_images(img)
    loss_v = torch.nn.functional.cross_entropy(pred, y)
    loss_i = torch.nn.functional.cross_entropy(img_pred, yi)
    loss = loss_v + cfg['lambda_weight'] * loss_i
    loss.backward()
    out = dict(cfg=np.array(json.dumps(cfg)), fingerprint=fingerprint(sd), logits=pred.detach().numpy(),
               img_logits=img_pred.detach().numpy(), loss=np.array(loss.item()), loss_voxel=np.array(loss_v.item()),
               loss_image=np.array(loss_i.item()), target=y.numpy(), img_target=yi.numpy(),
               img_argmax=img_pred.argmax(1).numpy())
    top2 = img_pred.detach().topk(2, dim=1).values
    out['img_top2_gap'] = (top2[:, 0] - top2[:, 1]).numpy()
    names = []
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().flatten()
        names.append(k)
        idx = sample_idx(g.numel())
        out['gnorm/' + k] = np.array(float(g.double().norm()))
        out['gsum/' + k] = np.array(float(g.double().sum()))
        out['gidx/' + k] = idx
        out['gval/' + k] = g[idx].numpy()
        if g.numel() <= 4096:
            out['gfull/' + k] = p.grad.detach().numpy()
    out['grad_names'] = np.array(json.dumps(names))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(f'{name}: loss {loss.item():.6f} (voxel {loss_v.item():.6f}, image {loss_i.item():.6f}) img argmax '
          f'{img_pred.argmax(1).tolist()} gap_min {out["img_top2_gap"].min():.4f} {len(names)} grads')


def tokenizer_cases():
    from models import embed_layer_3d_modality as ref_embed
    out = {}
    sid = [0]

    def pu(shape):
        sid[0] += 1
        return vo.portable_uniform(tuple(shape), 11, sid[0]).float()

    for tag, cls, V, c, P, D, B in [('ve30', 'VoxelEmbed', 30, 6, 5, 384, 2), ('ve32', 'VoxelEmbed', 32, 6, 5, 384, 2),
                                    ('na128', 'VoxelEmbed_no_average', 128, 9, 14, 768, 1),
                                    ('np30', 'VoxelNaiveProjection', 30, 6, 5, 384, 2),
                                    ('ve128', 'VoxelEmbed', 128, 16, 8, 768, 1)]:
        m = getattr(ref_embed, cls)(voxel_size=V, cell_size=c, patch_size=P, embed_dim=D)
        conv = m.proj[0]
        w = (pu(conv.weight.shape) - 0.5) * 0.2
        b = (pu(conv.bias.shape) - 0.5) * 0.2
        with torch.no_grad():
            conv.weight.copy_(w); conv.bias.copy_(b)
        x = (pu((B, 1, V, V, V)) < 0.1).float()
        y = m(x).detach()
        flat = y.flatten()
        idx = sample_idx(flat.numel(), 4096)
        out[tag + '/shape'] = np.array(y.shape)
        out[tag + '/num_patches'] = np.array(m.num_patches)
        out[tag + '/idx'] = idx
        out[tag + '/val'] = flat[idx].numpy()
        out[tag + '/sum'] = np.array(float(flat.double().sum()))
        out[tag + '/abssum'] = np.array(float(flat.double().abs().sum()))
        out[tag + '/cfg'] = np.array(json.dumps(dict(cls=cls, V=V, c=c, P=P, D=D, B=B)))
        print(tag, tuple(y.shape), float(flat.abs().max()))
    np.savez_compressed(os.path.join(HERE, 'tokenizers.npz'), **out)


if __name__ == '__main__':
    torch.manual_seed(9)
    torch.set_num_threads(8)
    only = sys.argv[1:]
    if not only or 'tokenizers' in only:
        tokenizer_cases()
    for n, c in CASES.items():
        if not only or n in only:
            run_case(n, c)
    if not only or 'lwf' in only:
        lwf_case()
