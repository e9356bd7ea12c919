"""CPU: the oracle (oracle/voxel_oracle.py) replayed against fixtures captured from the
reference itself (tests/golden/make_golden.py).  This is what pins the oracle."""
import json

import numpy as np
import pytest
import torch

from oracle import voxel_oracle as vo
from tests._util import VOXEL_CASES, load_case, rebuild_inputs, fwd_kwargs, check_grads_against_golden, GOLDEN


@pytest.mark.parametrize('name', [c for c in VOXEL_CASES if not c.startswith('cfg3')])
def test_voxel_model_matches_reference(name):
    z, cfg = load_case(name)
    sd, x, y = rebuild_inputs(cfg, z)
    np.testing.assert_array_equal(y.numpy(), z['target'])
    logits, loss, grads = vo.loss_and_grads(sd, x, y, **fwd_kwargs(cfg))
    np.testing.assert_allclose(logits.numpy(), z['logits'], rtol=0, atol=1e-5)      # <=1e-5 fp32 bar
    assert abs(float(loss) - float(z['loss'])) <= 1e-5
    np.testing.assert_array_equal(logits.argmax(1).numpy(), z['argmax'])
    assert set(grads) == set(json.loads(str(z['grad_names'])))                       # same used-parameter set
    check_grads_against_golden(z, grads, rtol=1e-4, atol=1e-7)


def test_cfg3_real_geometry_forward_matches_reference():
    """deit_base as built (H=3) + VoxelEmbed_no_average(128, 9, 14) + group_embed, B=1 (eval)."""
    z, cfg = load_case('cfg3_base_v128_group_b1')
    sd, x, y = rebuild_inputs(cfg, z)
    with torch.no_grad():
        logits = vo.forward(sd, x, **fwd_kwargs(cfg))
    np.testing.assert_allclose(logits.numpy(), z['logits'], rtol=0, atol=2e-5)
    np.testing.assert_array_equal(logits.argmax(1).numpy(), z['argmax'])


def test_tokenizers_match_reference():
    z = np.load(f'{GOLDEN}/tokenizers.npz')
    sid = [0]

    def pu(shape):
        sid[0] += 1
        return vo.portable_uniform(tuple(shape), 11, sid[0]).float()

    for tag in ['ve30', 've32', 'na128', 'np30', 've128']:
        c = json.loads(str(z[tag + '/cfg']))
        wshape = (c['D'], 1, c['c'], c['c'], c['c']) if c['cls'] != 'VoxelNaiveProjection' else (c['D'], 1, c['c'], c['c'])
        w = (pu(wshape) - 0.5) * 0.2
        b = (pu((c['D'],)) - 0.5) * 0.2
        x = (pu((c['B'], 1, c['V'], c['V'], c['V'])) < 0.1).float()
        fn = {'VoxelEmbed': vo.voxel_embed, 'VoxelEmbed_no_average': vo.voxel_embed_no_average,
              'VoxelNaiveProjection': vo.voxel_naive_projection}[c['cls']]
        y = fn(x, w, b, c['c'])
        assert tuple(y.shape) == tuple(z[tag + '/shape'])
        flat = y.flatten()
        np.testing.assert_allclose(flat[torch.from_numpy(z[tag + '/idx'])].numpy(), z[tag + '/val'], atol=2e-6, rtol=0)
        assert abs(float(flat.double().sum()) - float(z[tag + '/sum'])) < 1e-2
        if c['cls'] == 'VoxelEmbed':   # the z-folded form the HIP tokenizer uses is the same function
            y2 = vo.voxel_embed_folded(x, w, b, c['c'])
            np.testing.assert_allclose(y2.numpy(), y.numpy(), atol=2e-6, rtol=0)


def test_reference_quirks_are_reproduced():
    # deit_base is built with 3 heads (vit_3d_2d_pretrain.py:298-306)
    assert vo.BACKBONES['deit_base_patch16_224']['num_heads'] == 3
    # voxel_pos_embed / group tokens are zero-initialised (vit_3d_2d_pretrain.py:370-383)
    sd = vo.init_state_dict(backbone='deit_tiny_patch16_224', embed_layer='VoxelEmbed_no_average', voxel_size=12,
                            cell=4, patch=3, n_classes=10, pos_embedding='group_embed')
    for k in ('voxel_pos_embed', 'group_pos_embed', 'group_cls_token'):
        assert float(sd[k].abs().sum()) == 0.0
    assert sd['voxel_pos_embed'].shape == (1, 10, 192) and sd['group_pos_embed'].shape == (1, 4, 192)
    # group_embed attention mixes samples of the batch: changing sample 1 changes sample 0's logits
    sd = vo.init_state_dict(backbone='deit_tiny_patch16_224', embed_layer='VoxelEmbed_no_average', voxel_size=12,
                            cell=4, patch=3, n_classes=10, pos_embedding='group_embed', exercise_all=True)
    x, _ = vo.synthetic_batch(2, 12, 10)
    kw = dict(backbone='deit_tiny_patch16_224', embed_layer='VoxelEmbed_no_average', cell=4, patch=3,
              pos_embedding='group_embed')
    with torch.no_grad():
        a = vo.forward(sd, x, **kw)
        x2 = x.clone(); x2[1] = 1 - x2[1]
        b = vo.forward(sd, x2, **kw)
    assert float((a[0] - b[0]).abs().max()) > 1e-5


def test_adam_step_matches_torch_optim():
    torch.manual_seed(0)
    p = torch.randn(1000); g = torch.randn(1000)
    q = torch.nn.Parameter(p.clone()); opt = torch.optim.Adam([q], lr=1e-3)
    m = torch.zeros(1000); v = torch.zeros(1000); pp = p.clone()
    for step in range(1, 4):
        q.grad = g * step
        opt.step()
        vo.adam_step(pp, g * step, m, v, step)
        np.testing.assert_allclose(pp.numpy(), q.detach().numpy(), rtol=0, atol=1e-7)


def _lwf_inputs(cfg):
    img = (vo.portable_uniform((cfg['batch'], 3, 224, 224), 9, 7001) * 2 - 1).float()
    yi = (vo.portable_uniform((cfg['batch'],), 9, 7002) * 1000).long()
    return img, yi


def test_forward_images_and_lwf_loss_match_reference():
    """2-D branch (vit_3d_2d_pretrain.py:435-451) + the LwF loss of train_cls_voxel.py:250-267, reference-generated fixture."""
    z, cfg = load_case('tiny_v12_lwf_b2')
    sd, x, y = rebuild_inputs(cfg, z)
    img, yi = _lwf_inputs(cfg)
    np.testing.assert_array_equal(yi.numpy(), z['img_target'])
    lv, li, loss, grads = vo.lwf_loss_and_grads(sd, x, y, img, yi, lambda_weight=cfg['lambda_weight'], **fwd_kwargs(cfg))
    np.testing.assert_allclose(lv.numpy(), z['logits'], rtol=0, atol=1e-5)
    np.testing.assert_allclose(li.numpy(), z['img_logits'], rtol=0, atol=1e-5)
    np.testing.assert_array_equal(li.argmax(1).numpy(), z['img_argmax'])
    assert abs(float(loss) - float(z['loss'])) <= 1e-5
    assert set(grads) == set(json.loads(str(z['grad_names'])))
    check_grads_against_golden(z, grads, rtol=1e-4, atol=1e-7)


def _trained_fixture():
    z = np.load(f'{GOLDEN}/trained_cfg1_small_v30_adam60.npz')
    cfg = json.loads(str(z['cfg']))
    kw = {k: cfg[k] for k in ('backbone', 'embed_layer', 'voxel_size', 'cell', 'patch', 'n_classes', 'pos_embedding', 'head')}
    sd = vo.init_state_dict(seed=9, exercise_all=False, portable=True, **kw)
    fp = np.array([[float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in sorted(sd)])
    np.testing.assert_allclose(fp, z['fingerprint'], rtol=1e-12, atol=1e-12)         # the same initial parameters as the reference run
    data = [vo.synthetic_class_batch(cfg['batch'], cfg['voxel_size'], cfg['cell'], cfg['labels'], seed=500 + i) for i in range(cfg['n_batches'])]
    held = vo.synthetic_class_batch(cfg['held_batch'], cfg['voxel_size'], cfg['cell'], cfg['labels'], seed=999)
    return z, cfg, sd, data, held


def test_oracle_training_trajectory_matches_the_reference_optimizer():
    """Trained-state fixture (tests/golden/make_golden_trained.py): the REFERENCE model trained with torch.optim.Adam(lr = 1e-3) for 60
    steps.  The oracle's forward / autograd / adam_step walk the same losses and end at the same held-out decisions: 6 distinct classes,
    every top-2 gap above 1e-2 -- the argmax criterion on logits that DO depend on the input."""
    z, cfg, sd, data, held = _trained_fixture()
    fk = dict(backbone=cfg['backbone'], embed_layer=cfg['embed_layer'], cell=cfg['cell'], patch=cfg['patch'])
    names = vo.used_param_names(sd)
    m = {k: torch.zeros_like(sd[k]) for k in names}
    v = {k: torch.zeros_like(sd[k]) for k in names}
    for step in range(cfg['steps']):
        x, y = data[step % len(data)]
        _, loss, grads = vo.loss_and_grads(sd, x, y, **fk)
        assert abs(float(loss) - float(z['losses'][step])) <= 2e-4 * max(1.0, abs(float(z['losses'][step]))), (step, float(loss), float(z['losses'][step]))
        for k, g in grads.items():
            vo.adam_step(sd[k], g, m[k], v[k], step + 1, lr=cfg['lr'])
    with torch.no_grad():
        logits = vo.forward(sd, held[0], **fk)
    np.testing.assert_allclose(logits.numpy(), z['held_logits'], rtol=0, atol=5e-3)      # 60 steps of fp32 round-off apart
    np.testing.assert_array_equal(logits.argmax(1).numpy(), z['held_argmax'])
    assert len(set(z['held_argmax'].tolist())) >= 6 and float(z['held_top2_gap'].min()) > 1e-2
