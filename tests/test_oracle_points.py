"""CPU: the point-cloud oracle (oracle/point_oracle.py) replayed against fixtures captured from the reference's own
models/3DViT/model.py + data/pointnet_util.py (tests/golden/make_golden_points.py), train-mode BatchNorm."""
import json

import numpy as np
import pytest
import torch

from oracle import point_oracle as po
from tests._util import check_grads_against_golden, GOLDEN

POINT_CASES = ['pts_cls_tiny_n64_b3', 'pts_seg_tiny_n64_b2', 'pts_cls_tiny_n1024_b2', 'pts_seg_tiny_n2048_b1',
               'pts_seg_tiny_n64_am_b2']       # cfg.model.head == 'AMSoftmax' (models/3DViT/model.py:427-428)
# models/3DViT_1_layer, 3DViT_0_layer, 3DViT_LWF (PointTransformerSeg + forward_images)
VARIANT_CASES = ['pts_seg1_tiny_n64_b2', 'pts_seg1_small_n256_b2', 'pts_seg0_tiny_n64_b2', 'pts_seglwf_tiny_n64_b2']


def load_point_case(name):
    z = np.load(f'{GOLDEN}/{name}.npz')
    cfg = json.loads(str(z['cfg']))
    variant = cfg.setdefault('variant', '3DViT')
    sd = po.init_state_dict(backbone=cfg['backbone'], n_classes=cfg['n_classes'], d_points=cfg['d_points'], seed=9, variant=variant,
                            head=cfg.get('head', 'default'))
    x, y, starts = po.synthetic_points(cfg['batch'], cfg['n_points'], cfg['d_points'], cfg['n_classes'], cfg['task'], seed=9,
                                       variant=variant)
    for i, st in enumerate(starts):
        np.testing.assert_array_equal(st.numpy(), z[f'start{i}'])
    np.testing.assert_array_equal(y.numpy(), z['target'])
    return z, cfg, sd, x, y, starts


def lwf_images(cfg):
    """The image batch / teacher labels make_golden_points.py fed to forward_images."""
    img = (po.vo.portable_uniform((cfg['batch'], 3, 224, 224), 9, 7001) * 2 - 1).float()
    yi = (po.vo.portable_uniform((cfg['batch'],), 9, 7002) * 1000).long()
    return img, yi


@pytest.mark.parametrize('name', POINT_CASES)
def test_point_model_matches_reference(name):
    z, cfg, sd, x, y, starts = load_point_case(name)
    kw = dict(task=cfg['task'], backbone=cfg['backbone'], starts=starts)
    logits, loss, grads, stats = po.loss_and_grads(sd, x, y, training=True, **kw)
    tol = 2e-5 * max(1.0, float(np.abs(z['logits']).max()) / 5)       # fp32 rounding scales with the logits (AM-softmax: s = 30)
    np.testing.assert_allclose(logits.numpy(), z['logits'], rtol=0, atol=tol)
    assert abs(float(loss) - float(z['loss'])) <= 1e-5
    sure = z['top2_gap'] > 1e-3
    np.testing.assert_array_equal(logits.argmax(-1).numpy()[sure], z['argmax'][sure])
    assert set(grads) == set(json.loads(str(z['grad_names'])))
    # atol 2e-6: the gradients of a conv/linear bias that feeds a train-mode BatchNorm are exactly zero in theory and
    # pure rounding noise (~2e-7) in both implementations
    # rtol 1e-3: fp32 reduction-order noise over up to 32k grouped rows (conv2d in the reference vs a row GEMM here)
    check_grads_against_golden(z, grads, rtol=1e-3, atol=tol / 10)       # 2e-6 for logits of order one
    for k, v in stats.items():                                   # BatchNorm running statistics after one train-mode forward
        np.testing.assert_allclose(v.numpy(), z['stat/' + k], rtol=1e-5, atol=1e-6)
    with torch.no_grad():        # the golden eval pass ran after the train-mode pass had updated the running statistics
        ev = po.forward({**sd, **stats}, x, training=False, **kw)
    np.testing.assert_allclose(ev.numpy(), z['logits_eval'], rtol=0, atol=tol)


@pytest.mark.parametrize('name', VARIANT_CASES)
def test_point_variants_match_reference(name):
    z, cfg, sd, x, y, starts = load_point_case(name)
    kw = dict(task='seg', backbone=cfg['backbone'], starts=starts, variant=cfg['variant'])
    img, yi = lwf_images(cfg)
    np.testing.assert_array_equal(yi.numpy(), z['img_target'])
    if cfg.get('lwf'):
        logits, li, loss, grads, stats = po.lwf_loss_and_grads(sd, x, y, img, yi, cfg['lambda_weight'], training=True, **kw)
        assert abs(float(po.loss_fn(logits, y)) - float(z['loss_points'])) <= 1e-5
        assert abs(float(torch.nn.functional.cross_entropy(li, yi)) - float(z['loss_image'])) <= 1e-5
    else:
        logits, loss, grads, stats = po.loss_and_grads(sd, x, y, training=True, **kw)
        with torch.no_grad():
            li = po.forward_images(sd, img, backbone=cfg['backbone'])
    np.testing.assert_allclose(logits.numpy(), z['logits'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(li.numpy(), z['img_logits'], rtol=0, atol=2e-5)
    sure = z['img_top2_gap'] > 1e-3
    np.testing.assert_array_equal(li.argmax(1).numpy()[sure], z['img_argmax'][sure])
    assert abs(float(loss) - float(z['loss'])) <= 1e-5
    sure = z['top2_gap'] > 1e-3
    np.testing.assert_array_equal(logits.argmax(-1).numpy()[sure], z['argmax'][sure])
    assert set(grads) == set(json.loads(str(z['grad_names'])))
    check_grads_against_golden(z, grads, rtol=1e-3, atol=2e-6)
    for k, v in stats.items():
        np.testing.assert_allclose(v.numpy(), z['stat/' + k], rtol=1e-5, atol=1e-6)
    with torch.no_grad():
        ev = po.forward({**sd, **stats}, x, training=False, **kw)
    np.testing.assert_allclose(ev.numpy(), z['logits_eval'], rtol=0, atol=2e-5)


def test_metrics_and_sgd():
    logits = torch.tensor([[2.0, 1.0, 0.0], [0.0, 3.0, 1.0], [0.0, 0.0, 5.0], [4.0, 0.0, 1.0]])
    tgt = torch.tensor([0, 1, 1, 0])
    inst, cls = po.cls_accuracy(logits, tgt, 3)
    assert abs(inst - 0.75) < 1e-9 and abs(cls - 0.75) < 1e-9
    # part IoU with the argmax restricted to the shape's own parts
    seg_classes = {'a': [0, 1], 'b': [2, 3]}
    lg = torch.zeros(1, 4, 4); lg[0, :, 3] = 9.0; lg[0, :2, 0] = 1.0; lg[0, 2:, 1] = 1.0     # part 3 is NOT a part of category a
    tg = torch.tensor([[0, 0, 1, 0]])
    (cat, iou), = po.part_iou(lg, tg, seg_classes)
    assert cat == 'a' and abs(iou - (2 / 3 + 1 / 2) / 2) < 1e-9
    p = torch.randn(50); q = torch.nn.Parameter(p.clone()); opt = torch.optim.SGD([q], lr=0.01, momentum=0.9)
    buf = torch.zeros(50); pp = p.clone()
    for step in range(3):
        g = torch.randn(50)
        q.grad = g.clone(); opt.step()
        po.sgd_momentum_step(pp, g, buf, first=(step == 0))
        assert float((pp - q.detach()).abs().max()) < 1e-7


def test_point_oracle_sgd_trajectory_matches_the_reference_optimizer():
    """Trained-state fixture of the point path (tests/golden/make_golden_points_trained.py): the REFERENCE PointTransformerCls trained with
    torch.optim.SGD(lr = 0.01, momentum = 0.9) for 80 train-mode steps (BatchNorm batch statistics, running statistics updated).  The oracle's
    forward / autograd / sgd_momentum_step walk the same losses and end at the same eval-mode decisions on a held-out batch (5 classes)."""
    import json
    z = np.load(f'{GOLDEN}/trained_pts_cls_tiny_n64_sgd80.npz')
    cfg = json.loads(str(z['cfg']))
    sd = po.init_state_dict(backbone=cfg['backbone'], n_classes=cfg['n_classes'], d_points=cfg['d_points'], seed=9)
    data = [po.synthetic_class_points(cfg['batch'], cfg['n_points'], cfg['labels'], seed=600 + i) for i in range(cfg['n_batches'])]
    held = po.synthetic_class_points(cfg['held_batch'], cfg['n_points'], cfg['labels'], seed=999)
    names = po.used_param_names(sd)
    buf = {k: torch.zeros_like(sd[k]) for k in names}
    for step in range(cfg['steps']):
        x, y, starts = data[step % len(data)]
        _, loss, grads, stats = po.loss_and_grads(sd, x, y, backbone=cfg['backbone'], starts=starts, task='cls')
        ref = float(z['losses'][step])
        # train-mode BatchNorm on 8 clouds + momentum SGD is a chaotic map: two fp32 implementations that agree to 4e-7 after five steps are
        # 2e-4 apart after ten and ~1e-2 after forty (measured); the first ten steps pin the optimizer, the rest bounds the drift
        tol = 1e-3 * max(1.0, abs(ref)) if step < 10 else 4e-2 * max(abs(ref), 0.05)
        assert abs(float(loss) - ref) <= tol, (step, float(loss), ref)
        for k, g in grads.items():
            po.sgd_momentum_step(sd[k], g, buf[k], lr=cfg['lr'], momentum=cfg['momentum'], first=step == 0)
        sd.update(stats)                                          # the BatchNorm running statistics of the train-mode forward
    with torch.no_grad():
        logits = po.forward(sd, held[0], backbone=cfg['backbone'], starts=held[2], task='cls', training=False)
    clear = z['held_top2_gap'] > 1.5                             # (the drift moves the held-out logits by up to 0.6)
    assert int(clear.sum()) >= 20 and len(set(z['held_argmax'][clear].tolist())) >= 4
    np.testing.assert_array_equal(logits.argmax(1).numpy()[clear], z['held_argmax'][clear])
