"""GPU (MI355X) parity of the point-cloud path: geometry kernels bit-exact on indices vs the oracle, BatchNorm / interpolation
vs plain fp32 PyTorch, and the PointEngine (PointTransformerCls / PointTransformerSeg, train-mode BatchNorm) vs fixtures
captured from the reference's own models/3DViT/model.py.  Bars: neighbour / FPS indices bit-exact, logits and loss within
1e-3, gradients within 3 % rms / 12 % worst sampled entry (plain-bf16 backward)."""
import ctypes
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from simple3d_former_amd import _lib as L
    from simple3d_former_amd.point_engine import PointEngine

from oracle import point_oracle as po
from tests._util import check_grads_against_golden, GOLDEN
from tests.test_oracle_points import POINT_CASES, VARIANT_CASES, load_point_case, lwf_images

DEV = 'cuda'


def rel_err(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('B,N,npoint', [(3, 64, 64), (2, 1024, 256), (2, 2048, 2048), (4, 1000, 250)])
def test_fps_indices_bit_exact(B, N, npoint):
    x, _, _ = po.synthetic_points(B, N if N % 4 == 0 else N + (4 - N % 4), 6, 40, 'cls', seed=N)
    xyz = x[:, :N, :3].contiguous()
    start = torch.randint(0, N, (B,), generator=torch.Generator().manual_seed(1))
    ref = po.farthest_point_sample(xyz, npoint, start)
    idx = torch.empty(B, npoint, dtype=torch.int32, device=DEV); nx = torch.empty(B, npoint, 3, device=DEV)
    xd, sd_ = xyz.to(DEV), start.to(DEV)            # keep device tensors alive: L.ptr() only takes the address
    L.check(L.lib().s3d_fps(L.ptr(xd), ctypes.c_long(3), L.ptr(sd_), B, N, npoint, L.ptr(idx), L.ptr(nx), L.current_stream()), 'fps')
    assert torch.equal(idx.cpu().long(), ref)
    assert torch.equal(nx.cpu(), po.index_points(xyz, ref))


@pytest.mark.parametrize('B,S,N', [(2, 64, 64), (2, 256, 1024), (1, 2048, 2048), (3, 100, 300)])
def test_knn16_and_3nn_indices_bit_exact(B, S, N):
    g = torch.Generator().manual_seed(S + N)
    ref = torch.rand(B, N, 3, generator=g) * 2 - 1
    q = ref[:, torch.randperm(N, generator=g)[:S]].contiguous() if S <= N else torch.rand(B, S, 3, generator=g)
    want = po.knn_indices(q, ref, 16)
    idx = torch.empty(B, S, 16, dtype=torch.int32, device=DEV)
    qd, rd = q.to(DEV), ref.to(DEV)
    L.check(L.lib().s3d_knn(L.ptr(qd), L.ptr(rd), B, S, N, 16, L.ptr(idx), None, L.current_stream()), 'knn')
    d = po.square_distance(q, ref)
    got = idx.cpu().long()
    # indices equal except inside exact distance ties, where any order is a valid argsort
    same = got == want
    tied = torch.gather(d, 2, got) == torch.gather(d, 2, want)
    assert bool((same | tied).all()) and float(same.float().mean()) > 0.999
    widx, w = po.three_nn_weights(q, ref)
    i3 = torch.empty(B, S, 3, dtype=torch.int32, device=DEV); w3 = torch.empty(B, S, 3, device=DEV)
    L.check(L.lib().s3d_knn(L.ptr(qd), L.ptr(rd), B, S, N, 3, L.ptr(i3), L.ptr(w3), L.current_stream()), 'knn3')
    assert float((i3.cpu().long() == widx).float().mean()) > 0.999
    assert rel_err(w3, w) < 1e-5


@pytest.mark.parametrize('rows,C,K', [(640, 96, 0), (2048, 192, 16), (4096, 48, 0), (3 * 64 * 16, 96, 16), (1024, 384, 16), (520, 768, 0)])
def test_batchnorm_relu_max_fwd_bwd(rows, C, K):
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g) * 1.5 + 0.3
    gamma = 1 + 0.1 * torch.randn(C, generator=g); beta = 0.1 * torch.randn(C, generator=g)
    rm = torch.randn(C, generator=g) * 0.1; rv = 1 + 0.2 * torch.rand(C, generator=g)
    xr = x.clone().double().requires_grad_(True); gr = gamma.double().requires_grad_(True); br = beta.double().requires_grad_(True)
    rm_ref, rv_ref = rm.clone().double(), rv.clone().double()
    y = F.relu(F.batch_norm(xr, rm_ref, rv_ref, gr, br, training=True, momentum=0.1, eps=1e-5))
    if K:
        y = y.view(rows // K, K, C).max(dim=1)[0]
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())

    xd = x.to(DEV)
    f = lambda t: t.to(DEV)
    mean = torch.zeros(C, device=DEV); rstd = torch.zeros(C, device=DEV); sums = torch.zeros(2 * C, dtype=torch.float64, device=DEV)
    rmd, rvd, gd, bd = f(rm), f(rv), f(gamma), f(beta)
    out = torch.empty(y.shape, device=DEV); arg = torch.zeros(y.shape, dtype=torch.uint8, device=DEV)
    dx = torch.empty(rows, C, dtype=torch.bfloat16, device=DEV); dg = torch.zeros(C, device=DEV); db = torch.zeros(C, device=DEV)
    dyd = f(dy)
    a = L.fill(L.S3dBnArgs(), x=xd, ldx=C, rows=rows, C=C, K=K, eps=1e-5, momentum=0.1, gamma=gd, beta=bd, mean=mean, rstd=rstd,
               run_mean=rmd, run_var=rvd, sums=sums, y=out, ldo=C, arg=arg, dy=dyd, lddy=C, dx=dx, lddx=C, dgamma=dg, dbeta=db)
    L.check(L.lib().s3d_batchnorm_fwd(ctypes.byref(a), L.current_stream()), 'bn fwd')
    assert rel_err(out, y.detach()) < 1e-5
    assert rel_err(rmd, rm_ref) < 1e-5 and rel_err(rvd, rv_ref) < 1e-5
    L.check(L.lib().s3d_batchnorm_bwd(ctypes.byref(a), L.current_stream()), 'bn bwd')
    assert rel_err(dx.float(), xr.grad) < 1.5e-2               # bf16 output
    assert rel_err(dg, gr.grad) < 1e-4 and rel_err(db, br.grad) < 1e-4


@pytest.mark.parametrize('rows,C,K', [(524288, 64, 0), (32768 * 16, 64, 16)])
def test_batchnorm_backward_bf16_dx_keeps_its_zero_column_sums(rows, C, K):
    """The gradient that leaves a train-mode BatchNorm sums to zero over the rows of every channel (pointnet_util.py:238-241); its bf16
    copy (S3dBnArgs::dx) keeps that property to within half an ulp per WORKGROUP because the apply kernel carries the rounding residue
    from element to element (points.hip: bn_bwd_apply_vec_kernel) -- plain rounding leaves a random walk over all rows, which the
    consumers multiply by their columns' means (the 4 - 7 % gradient errors of cfg-4's input layers until round 4)."""
    g = torch.Generator().manual_seed(C + K)
    x = torch.randn(rows, C, generator=g) * 1.5 + 0.3
    gamma = 1 + 0.1 * torch.randn(C, generator=g); beta = 0.1 * torch.randn(C, generator=g)
    xr = x.clone().double().requires_grad_(True)
    y = F.relu(F.batch_norm(xr, None, None, gamma.double(), beta.double(), training=True, eps=1e-5))
    if K:
        y = y.view(rows // K, K, C).max(dim=1)[0]
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    ref = xr.grad
    f = lambda t: t.to(DEV)
    xd, gd, bd, dyd = f(x), f(gamma), f(beta), f(dy)
    mean = torch.zeros(C, device=DEV); rstd = torch.zeros(C, device=DEV); sums = torch.zeros(2 * C, dtype=torch.float64, device=DEV)
    rmd, rvd = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    out = torch.empty(y.shape, device=DEV); arg = torch.zeros(y.shape, dtype=torch.uint8, device=DEV)
    dx = torch.empty(rows, C, dtype=torch.bfloat16, device=DEV); dg = torch.zeros(C, device=DEV); db = torch.zeros(C, device=DEV)
    a = L.fill(L.S3dBnArgs(), x=xd, ldx=C, rows=rows, C=C, K=K, eps=1e-5, momentum=0.1, gamma=gd, beta=bd, mean=mean, rstd=rstd,
               run_mean=rmd, run_var=rvd, sums=sums, y=out, ldo=C, arg=arg, dy=dyd, lddy=C, dx=dx, lddx=C, dgamma=dg, dbeta=db)
    L.check(L.lib().s3d_batchnorm_fwd(ctypes.byref(a), L.current_stream()), 'bn fwd')
    L.check(L.lib().s3d_batchnorm_bwd(ctypes.byref(a), L.current_stream()), 'bn bwd')
    got = dx.double().cpu()
    assert rel_err(got, ref) < 1.5e-2
    assert float(ref.sum(0).abs().max()) < 1e-9 * rows                     # the exact gradient: zero column sums
    walk = (ref.float().to(torch.bfloat16).double() - ref).sum(0)           # what rounding every element on its own leaves per channel
    kept = got.sum(0)
    ratio = float(kept.pow(2).mean().sqrt() / walk.pow(2).mean().sqrt())
    print(f'bf16 dx column sums: rms {float(kept.pow(2).mean().sqrt()):.3e} against {float(walk.pow(2).mean().sqrt()):.3e} for plain rounding ({ratio:.3f})')
    assert ratio < 0.4


def test_gather_scatter_interp():
    g = torch.Generator().manual_seed(5)
    B, N, S, K, C = 2, 40, 10, 16, 8
    xyz = torch.rand(B, N, 3, generator=g); feats = torch.randn(B, N, C, generator=g)
    new_xyz = xyz[:, :S].contiguous()
    idx = po.knn_indices(new_xyz, xyz, K)
    ref = torch.cat([po.index_points(xyz, idx) - new_xyz[:, :, None], po.index_points(feats, idx)], dim=-1).reshape(B * S * K, 3 + C)
    lda = 16
    A = torch.zeros(2, B * S * K, lda, dtype=torch.bfloat16, device=DEV)
    idxd = idx.to(torch.int32).to(DEV)
    xyzd, nxd, fd = xyz.to(DEV), new_xyz.to(DEV), feats.to(DEV)
    L.check(L.lib().s3d_group_gather(L.ptr(xyzd), L.ptr(nxd), L.ptr(fd), L.ptr(idxd), B, N, S, K, C,
                                     L.ptr(A[0]), L.ptr(A[1]), lda, L.current_stream()), 'gather')
    got = (A[0].float() + A[1].float()).cpu()
    assert rel_err(got[:, :3 + C], ref) < 1e-4 and float(got[:, 3 + C:].abs().max()) == 0.0
    dA = torch.randn(B * S * K, lda, generator=g)
    dfe = torch.zeros(B * N, C, device=DEV)
    dAd = dA.to(DEV)
    L.check(L.lib().s3d_group_scatter(L.ptr(dAd), lda, L.ptr(idxd), B, N, S, K, C, L.ptr(dfe), L.current_stream()), 'scatter')
    want = torch.zeros(B, N, C).scatter_add_(1, idx.reshape(B, -1, 1).expand(-1, -1, C), dA[:, 3:3 + C].reshape(B, S * K, C))
    assert rel_err(dfe.view(B, N, C), want) < 1e-5
    # interpolation
    f1 = torch.randn(B, S, C, generator=g); f2 = torch.randn(B, N, C, generator=g)
    i3, w3 = po.three_nn_weights(xyz, new_xyz)
    ref_out = (po.index_points(f1, i3) * w3[..., None]).sum(2) + f2
    out = torch.empty(B * N, C, device=DEV)
    i3d, w3d = i3.to(torch.int32).to(DEV), w3.to(DEV)
    f1d, f2d = f1.to(DEV), f2.to(DEV)
    L.check(L.lib().s3d_interp3(L.ptr(f1d), S, L.ptr(f2d), L.ptr(i3d), L.ptr(w3d), B, N, C, L.ptr(out), L.current_stream()), 'interp')
    assert rel_err(out.view(B, N, C), ref_out) < 1e-5


@pytest.fixture
def deterministic_reductions():
    """Gradient parity against the reference goldens runs without split-K / atomic-order noise (s3d_set_deterministic): the error
    that is left is the bf16 rounding of the backward operands, so the bars need no allowance for run-to-run variation."""
    lib = L.lib()
    was = lib.s3d_get_deterministic()
    lib.s3d_set_deterministic(1)
    yield
    lib.s3d_set_deterministic(was)


@pytest.mark.parametrize('name', POINT_CASES)
def test_point_engine_matches_reference_golden(name, deterministic_reductions):
    z, cfg, sd, x, y, starts = load_point_case(name)
    eng = PointEngine(backbone=cfg['backbone'], n_points=cfg['n_points'], d_points=cfg['d_points'], n_classes=cfg['n_classes'],
                      task=cfg['task'], device=DEV, head=cfg.get('head', 'default'))
    eng.load_state_dict(sd)
    sts = tuple(s.to(DEV) for s in starts)
    logits = eng.forward(x.to(DEV), sts).cpu()
    err = float(np.abs(logits.numpy().reshape(z['logits'].shape) - z['logits']).max())
    assert err <= 1e-3, f'logits max abs err {err:.3e}'
    sure = z['top2_gap'] > 2e-3
    np.testing.assert_array_equal(logits.argmax(-1).numpy().reshape(z['argmax'].shape)[sure], z['argmax'][sure])
    loss = float(eng.cross_entropy(cfg['batch'], y.to(DEV)))
    assert abs(loss - float(z['loss'])) <= 1e-3
    for k, bn in eng.bns.items():                                # BatchNorm running statistics after one train-mode forward
        np.testing.assert_allclose(bn.run_mean.cpu().numpy(), z['stat/' + k + '.running_mean'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(bn.run_var.cpu().numpy(), z['stat/' + k + '.running_var'], rtol=1e-4, atol=1e-5)
    eng.zero_grad()
    eng.backward(cfg['batch'])
    grads = {k: eng.arena.grad(k) for k in eng.shapes}
    assert set(grads) == set(json.loads(str(z['grad_names'])))
    # biases that feed a train-mode BatchNorm (and the final LayerNorm bias, which only reaches the loss through one) have a
    # theoretically ZERO gradient; both implementations return rounding noise there, so they are excluded from the comparison
    zero_theory = ('mlp_convs.0.bias', 'mlp_convs.1.bias', 'fc1.0.bias', 'fc2.0.bias', 'norm.bias')
    zero_theory = tuple(k for k in grads if k.endswith(zero_theory) and (k.startswith('transition_') or k == 'norm.bias'))
    zero_theory += ('fc1.2.bias', 'fc_pos_embed.2.bias')       # constant shifts of f: cancelled by the BatchNorms downstream
    # AM-softmax head: logits = 30 * cosine -> d(logits) and every rounding error of the backward are ~30x a Linear head's
    worst = check_grads_against_golden(z, grads, rtol=5e-3 if cfg.get('head') == 'AMSoftmax' else 3e-3, atol=3e-6, skip=zero_theory)
    print(f'{name}: logits err {err:.2e}, worst sampled grad err / rms {worst:.3f}')


@pytest.mark.parametrize('name', VARIANT_CASES)
def test_point_engine_variants_match_reference_golden(name):
    """models/3DViT_1_layer, 3DViT_0_layer, 3DViT_LWF: PointTransformerSeg forward / backward, forward_images on the shared
    blocks, and (LWF fixture) the gradient of CE(points) + lambda * CE(images) accumulated by the two backward passes."""
    z, cfg, sd, x, y, starts = load_point_case(name)
    eng = PointEngine(backbone=cfg['backbone'], n_points=cfg['n_points'], d_points=cfg['d_points'], n_classes=cfg['n_classes'],
                      task='seg', device=DEV, variant=cfg['variant'])
    eng.load_state_dict(sd)
    sts = tuple(s.to(DEV) for s in starts)
    logits = eng.forward(x.to(DEV), sts).cpu()
    err = float(np.abs(logits.numpy().reshape(z['logits'].shape) - z['logits']).max())
    assert err <= 1e-3, f'logits max abs err {err:.3e}'
    sure = z['top2_gap'] > 2e-3
    np.testing.assert_array_equal(logits.argmax(-1).numpy().reshape(z['argmax'].shape)[sure], z['argmax'][sure])
    loss = float(eng.cross_entropy(cfg['batch'], y.to(DEV)))
    assert abs(loss - float(z['loss_points'] if cfg.get('lwf') else z['loss'])) <= 1e-3
    for k, bn in eng.bns.items():
        np.testing.assert_allclose(bn.run_mean.cpu().numpy(), z['stat/' + k + '.running_mean'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(bn.run_var.cpu().numpy(), z['stat/' + k + '.running_var'], rtol=1e-4, atol=1e-5)
    eng.zero_grad()
    eng.backward(cfg['batch'])
    img, yi = lwf_images(cfg)
    li = eng.images.forward(img.to(DEV)).cpu()
    ierr = float(np.abs(li.numpy() - z['img_logits']).max())
    assert ierr <= 1e-3, f'image logits max abs err {ierr:.3e}'
    sure = z['img_top2_gap'] > 2e-3
    np.testing.assert_array_equal(li.argmax(1).numpy()[sure], z['img_argmax'][sure])
    image_only = ('patch_embed.', 'pos_embed', 'head.')
    if cfg.get('lwf'):
        loss_i = float(eng.images.cross_entropy(cfg['batch'], yi.to(DEV), grad_scale=cfg['lambda_weight']))
        assert abs(loss_i - float(z['loss_image'])) <= 1e-3
        eng.images.backward(cfg['batch'])
        grads = {k: eng.arena.grad(k) for k in eng.shapes}
    else:
        grads = {k: eng.arena.grad(k) for k in eng.shapes if not k.startswith(image_only)}
        for k in eng.shapes:
            if k.startswith(image_only):
                assert float(eng.arena.grad(k).abs().max()) == 0.0, k       # the point backward never touches the 2-D stem / head
    assert set(grads) == set(json.loads(str(z['grad_names'])))
    zero_theory = ('mlp_convs.0.bias', 'mlp_convs.1.bias', 'fc1.0.bias', 'fc2.0.bias', 'norm.bias')
    zero_theory = tuple(k for k in grads if k.endswith(zero_theory) and (k.startswith('transition_') or k == 'norm.bias'))
    if eng.levels:               # constant shifts of f are cancelled by the BatchNorms downstream ...
        zero_theory += ('fc1.2.bias', 'fc_pos_embed.2.bias')
    if cfg.get('lwf') or not eng.levels:   # ... but 3DViT_0_layer has none, and the image loss reaches norm.bias directly
        zero_theory = tuple(k for k in zero_theory if k != 'norm.bias')
    worst = check_grads_against_golden(z, grads, rtol=3e-3, atol=3e-6, skip=zero_theory)
    print(f'{name}: logits err {err:.2e}, image logits err {ierr:.2e}, worst sampled grad err / rms {worst:.3f}')


def test_point_engine_lwf_train_step_and_errors():
    """PointEngine.lwf_train_step = train_partseg_lwf.py:207-228 (one SGD step on CE(points) + lambda CE(images)) vs the oracle."""
    z, cfg, sd, x, y, starts = load_point_case('pts_seglwf_tiny_n64_b2')
    img, yi = lwf_images(cfg)
    kw = dict(task='seg', backbone=cfg['backbone'], starts=starts, variant=cfg['variant'])
    _, _, _, grads, _ = po.lwf_loss_and_grads(sd, x, y, img, yi, 0.1, training=True, **kw)
    eng = PointEngine(backbone=cfg['backbone'], n_points=64, d_points=22, n_classes=50, task='seg', device=DEV, variant='3DViT_LWF')
    eng.load_state_dict(sd)
    lp, li = eng.lwf_train_step(x.to(DEV), y.to(DEV), tuple(s.to(DEV) for s in starts), img.to(DEV), yi.to(DEV), 0.1)
    assert abs(float(lp) - float(z['loss_points'])) <= 1e-3 and abs(float(li) - float(z['loss_image'])) <= 1e-3
    new = eng.state_dict()
    for k in ('blocks.3.mlp.fc1.weight', 'pos_embed', 'head.weight', 'new_head.weight', 'transition_downs.1.sa.mlp_convs.1.weight'):
        want = sd[k] - 0.01 * grads[k]                            # first SGD step: buf = g, p -= lr * g
        step = float((0.01 * grads[k]).abs().max())
        assert float((new[k].cpu() - want).abs().max()) <= 0.05 * step + 1e-7, k
    base = PointEngine(backbone=cfg['backbone'], n_points=64, d_points=22, n_classes=50, task='seg', device=DEV)
    with pytest.raises(RuntimeError, match='no forward_images'):
        base.lwf_train_step(x.to(DEV), y.to(DEV), (), img.to(DEV), yi.to(DEV))
    with pytest.raises(ValueError, match='PointTransformerSeg only'):
        PointEngine(backbone=cfg['backbone'], n_points=64, d_points=6, n_classes=40, task='cls', device=DEV, variant='3DViT_1_layer')


def test_point_engine_sgd_training_reduces_loss():
    cfg = dict(backbone='deit_tiny_patch16_224', n_points=64, d_points=6, n_classes=40)
    sd = po.init_state_dict(backbone=cfg['backbone'], n_classes=40, d_points=6, seed=3)
    x, y, starts = po.synthetic_points(8, 64, 6, 40, 'cls', seed=4)
    eng = PointEngine(task='cls', device=DEV, **cfg)
    eng.load_state_dict(sd)
    sts = tuple(s.to(DEV) for s in starts)
    l0 = float(eng.train_step(x.to(DEV), y.to(DEV), sts))
    for _ in range(15):
        l1 = float(eng.train_step(x.to(DEV), y.to(DEV), sts))
    assert l1 < l0, f'{l0} -> {l1}'


@pytest.mark.parametrize('name', ['pts_cls_tiny_n64_b3', 'pts_seg_tiny_n64_b2'])
def test_drop_in_point_module_matches_reference(name):
    """classifier = PointTransformerCls(cfg).cuda(); pred = classifier(points); loss.backward(); SGD.step()
    (train_cls.py:69,117-123 / train_partseg.py:74,143-152) on the drop-in module, train- and eval-mode."""
    import types
    import simple3d_former_amd as s3d
    z, cfg, sd, x, y, starts = load_point_case(name)
    c = types.SimpleNamespace(num_point=cfg['n_points'], num_class=cfg['n_classes'], input_dim=cfg['d_points'],
                              model=types.SimpleNamespace(nblocks=4, nneighbor=16, transformer_dim=512, head='default',
                                                          transformer_backbone=cfg['backbone'], pretrained=False, name='3DViT'))
    model = (s3d.PointTransformerCls if cfg['task'] == 'cls' else s3d.PointTransformerSeg)(c)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected
    model = model.to(DEV).train()
    model.s3d_fps_starts = starts
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
    opt.zero_grad()
    pred = model(x.to(DEV))
    assert tuple(pred.shape) == tuple(z['logits'].shape)
    assert float(np.abs(pred.detach().cpu().numpy() - z['logits']).max()) <= 1e-3
    loss = F.cross_entropy(pred.reshape(-1, cfg['n_classes']), y.to(DEV).reshape(-1))
    assert abs(float(loss) - float(z['loss'])) <= 1e-3
    loss.backward()
    g = model.transition_ups[0].fc1[0].weight.grad
    assert g is not None and abs(float(g.double().norm()) - float(z['gnorm/transition_ups.0.fc1.0.weight'])) <= 0.03 * float(z['gnorm/transition_ups.0.fc1.0.weight'])
    assert model.pos_embed.grad is None and model.patch_embed.conv1.weight.grad is None      # unused, as in the reference
    opt.step()
    # BatchNorm buffers were updated in place by the train-mode forward (module buffers alias the engine's)
    np.testing.assert_allclose(model.transition_downs[0].sa.mlp_bns[0].running_mean.cpu().numpy(),
                               z['stat/transition_downs.0.sa.mlp_bns.0.running_mean'], rtol=1e-4, atol=1e-5)
    assert int(model.transition_downs[0].sa.mlp_bns[0].num_batches_tracked) == 1
    # eval mode: running statistics; the golden eval pass ran after exactly one train-mode pass, before any optimizer step
    model2 = (s3d.PointTransformerCls if cfg['task'] == 'cls' else s3d.PointTransformerSeg)(c)
    model2.load_state_dict(sd, strict=False)
    model2 = model2.to(DEV).train()
    model2.s3d_fps_starts = starts
    with torch.no_grad():
        model2(x.to(DEV))
        ev = model2.eval()(x.to(DEV))
    assert float(np.abs(ev.cpu().numpy() - z['logits_eval']).max()) <= 1e-3


def _variant_cfg(cfg):
    import types
    return types.SimpleNamespace(num_point=cfg['n_points'], num_class=cfg['n_classes'], input_dim=cfg['d_points'],
                                 model=types.SimpleNamespace(nblocks=4, nneighbor=16, transformer_dim=512, head='default',
                                                             transformer_backbone=cfg['backbone'], pretrained=False,
                                                             name=cfg['variant']))


def test_drop_in_variant_module_lwf_step_matches_reference():
    """train_partseg_lwf.py:207-228 on the drop-in module of models/3DViT_LWF: seg_pred = classifier(points);
    img_pred = classifier.forward_images(images); loss = CE + lambda * CE; ONE loss.backward(); optimizer.step()."""
    import simple3d_former_amd as s3d
    z, cfg, sd, x, y, starts = load_point_case('pts_seglwf_tiny_n64_b2')
    img, yi = lwf_images(cfg)
    model = s3d.model_module('3DViT_LWF').PointTransformerSeg(_variant_cfg(cfg))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all('last_pos_embed' in k for k in missing)
    model = model.to(DEV).train()
    model.s3d_fps_starts = starts
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
    opt.zero_grad()
    seg_pred = model(x.to(DEV))
    assert float(np.abs(seg_pred.detach().cpu().numpy() - z['logits']).max()) <= 1e-3
    loss = F.cross_entropy(seg_pred.contiguous().view(-1, 50), y.to(DEV).view(-1))
    img_pred = model.forward_images(img.to(DEV))
    assert float(np.abs(img_pred.detach().cpu().numpy() - z['img_logits']).max()) <= 1e-3
    loss = loss + cfg['lambda_weight'] * F.cross_entropy(img_pred, yi.to(DEV))
    assert abs(float(loss) - float(z['loss'])) <= 1e-3
    loss.backward()
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    assert set(grads) == set(json.loads(str(z['grad_names'])))
    zero_theory = tuple(k for k in grads if k.startswith('transition_') and k.endswith(('mlp_convs.0.bias', 'mlp_convs.1.bias',
                                                                                       'fc1.0.bias', 'fc2.0.bias')))
    check_grads_against_golden(z, grads, rtol=3e-3, atol=3e-6, skip=zero_theory + ('fc1.2.bias', 'fc_pos_embed.2.bias'))
    before = model.blocks[2].attn.qkv.weight.detach().clone()
    opt.step()
    assert float((model.blocks[2].attn.qkv.weight - before).abs().max()) > 0
    with pytest.raises(RuntimeError, match='head applied twice'):
        model(x.to(DEV), type='images')


@pytest.mark.parametrize('name', ['pts_seg1_tiny_n64_b2', 'pts_seg0_tiny_n64_b2'])
def test_drop_in_variant_modules_points_only(name):
    """classifier(points) alone (train_partseg.py:143-152 with model=3DViT_1_layer / 3DViT_0_layer): the 2-D stem / head stay
    without gradient, the FPS starts are drawn per level, a frozen (pretrained-style) stem is honoured by forward_images."""
    import simple3d_former_amd as s3d
    z, cfg, sd, x, y, starts = load_point_case(name)
    model = s3d.model_module(cfg['variant']).PointTransformerSeg(_variant_cfg(cfg))
    model.load_state_dict(sd, strict=False)
    for p in (model.head.weight, model.head.bias, *model.patch_embed.parameters()):     # what pretrained=True does (model.py:285-289)
        p.requires_grad = False
    model = model.to(DEV).train()
    model.s3d_fps_starts = starts
    pred = model(x.to(DEV))
    assert float(np.abs(pred.detach().cpu().numpy() - z['logits']).max()) <= 1e-3
    F.cross_entropy(pred.reshape(-1, 50), y.to(DEV).reshape(-1)).backward()
    assert model.pos_embed.grad is None and model.head.weight.grad is None and model.patch_embed.proj.weight.grad is None
    g = model.new_head.weight.grad
    ref = float(z['gnorm/new_head.weight'])
    assert g is not None and abs(float(g.double().norm()) - ref) <= 0.03 * ref
    model.zero_grad()
    img, yi = lwf_images(cfg)
    F.cross_entropy(model.forward_images(img.to(DEV)), yi.to(DEV)).backward()
    assert model.head.weight.grad is None and model.patch_embed.proj.weight.grad is None       # frozen
    assert model.pos_embed.grad is not None and float(model.pos_embed.grad.abs().max()) > 0   # pos_embed stays trainable (:287)
    assert model.new_head.weight.grad is None and model.fc1[0].weight.grad is None            # not in the image graph
    model.s3d_fps_starts = None                                                               # random starts: one per level, in range
    with torch.no_grad():
        out = model.eval()(x.to(DEV))
    assert tuple(out.shape) == (cfg['batch'], cfg['n_points'], 50) and bool(torch.isfinite(out).all())


def test_drop_in_module_honours_bn_momentum_adjust():
    """train_partseg.py:97-99,130: classifier.apply(bn_momentum_adjust) changes every BatchNorm's momentum per epoch."""
    import simple3d_former_amd as s3d
    z, cfg, sd, x, y, starts = load_point_case('pts_seg1_tiny_n64_b2')
    model = s3d.model_module('3DViT_1_layer').PointTransformerSeg(_variant_cfg(cfg))
    model.load_state_dict(sd, strict=False)
    model = model.to(DEV).train()
    model.s3d_fps_starts = starts

    def bn_momentum_adjust(m, momentum):
        if isinstance(m, torch.nn.BatchNorm2d) or isinstance(m, torch.nn.BatchNorm1d):
            m.momentum = momentum

    model = model.apply(lambda m: bn_momentum_adjust(m, 0.5))
    with torch.no_grad():
        model(x.to(DEV))
    k = 'transition_ups.0.fc2.2.running_mean'
    batch_mean = (z['stat/' + k] - 0.9 * sd[k].numpy()) / 0.1            # the fixture ran with the default momentum 0.1
    np.testing.assert_allclose(dict(model.named_buffers())[k].cpu().numpy(), 0.5 * sd[k].numpy() + 0.5 * batch_mean, rtol=1e-3, atol=1e-4)


def test_point_dp_trainer_two_halves_graphs_and_rccl_path():
    """PointDataParallelTrainer on one GPU: [forward, CE, backward_top] | RCCL | [backward_bottom] | RCCL | [SGD] captured as three
    HIP graphs with forced all-reduces, vs the plain eager train_step -- for the two-level model and a variant."""
    import os
    import torch.distributed as dist
    from simple3d_former_amd.parallel import PointDataParallelTrainer
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29519')
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        for variant, task, dp, nc in (('3DViT', 'cls', 6, 40), ('3DViT_1_layer', 'seg', 22, 50)):
            kw = dict(backbone='deit_tiny_patch16_224', n_points=64, d_points=dp, n_classes=nc)
            sd = po.init_state_dict(backbone=kw['backbone'], n_classes=nc, d_points=dp, seed=3, variant=variant)
            x, y, starts = po.synthetic_points(4, 64, dp, nc, task, seed=4, variant=variant)
            x, y, sts = x.to(DEV), y.to(DEV), tuple(s.to(DEV) for s in starts)
            ref = PointEngine(task=task, device=DEV, variant=variant, **kw); ref.load_state_dict(sd)
            eng = PointEngine(task=task, device=DEV, variant=variant, **kw); eng.load_state_dict(sd)
            tr = PointDataParallelTrainer(eng, use_graphs=True, force_collectives=True)
            split = eng.arena.offsets['cls_token']
            assert tr.slices == [(split, eng.arena.g.numel()), (0, split)] and eng.grad_scale == 1.0
            p0 = ref.arena.p.clone()
            for step in range(3):
                l_ref = float(ref.train_step(x, y, sts))
                l_dp = float(tr.step(x, y, sts))
                assert abs(l_ref - l_dp) <= 2e-3, f'{variant} step {step}: {l_ref} vs {l_dp}'
            upd = float((ref.arena.p - p0).abs().max())
            d = float((eng.arena.p - ref.arena.p).abs().max())
            # two NON-deterministic runs (fp32 atomics in different orders, train-mode BatchNorm on four clouds amplifying them): measured
            # 0.5 - 2.1 % of the largest update -- round 4's 2 % bar failed once in a full-suite run of round 6, and `pytest -x` in front
            # of 80 other tests is no place for a coin flip.  A wrong average / a missed bucket shows as ~100 %.
            assert d <= 0.05 * upd + 1e-6, f'{variant}: replicas differ by {d:.3e} (largest update {upd:.3e})'
            # running statistics follow the parameters, which may differ between two runs by the order of the fp32 atomics in the
            # wgrad / scatter kernels (bounded above at 5 % of an update): compare with a matching, not an absolute 1e-4, bar
            for a, b in zip(eng.bn_buffers(), ref.bn_buffers()):
                assert float((a - b).abs().max()) <= 1e-3 * (1.0 + float(b.abs().max()))
    finally:
        from tests._util import teardown_process_group
        teardown_process_group()


@pytest.mark.parametrize('B,N,S,C,ch', [(2, 40, 10, 8, 16), (3, 64, 64, 48, 96), (2, 256, 64, 96, 192), (1, 128, 32, 192, 384),
                                          (1, 64, 16, 384, 768)])     # ch > 512: two channel passes in the backward kernel
def test_group_project_fwd_bwd(B, N, S, C, ch):
    """s3d_group_project_*: conv0([xyz_rel | feats[idx]]) = Pf[idx] + xyz_rel . Wx^T + b vs the reference formulation (grouped
    rows through a [ch][3 + C] weight, sample_and_group + the first Conv2d of PointNetSetAbstraction) in fp64."""
    g = torch.Generator().manual_seed(B * N + C)
    K = 16
    xyz = torch.rand(B, N, 3, generator=g); feats = torch.randn(B, N, C, generator=g)
    new_xyz = xyz[:, torch.randperm(N, generator=g)[:S]].contiguous()
    idx = po.knn_indices(new_xyz, xyz, K)
    W = torch.randn(ch, 3 + C, generator=g) / (3 + C) ** 0.5; bias = torch.randn(ch, generator=g)
    Wd, fd = W.double().requires_grad_(True), feats.double().requires_grad_(True)
    bd = bias.double().requires_grad_(True)
    grouped = torch.cat([po.index_points(xyz.double(), idx) - new_xyz.double()[:, :, None], po.index_points(fd, idx)], dim=-1)
    ref = grouped.reshape(B * S * K, -1) @ Wd.t() + bd
    Pf = (feats.double().reshape(B * N, C) @ W.double()[:, 3:].t()).float()
    dev = lambda t: t.to(DEV)
    xyz_d, nx_d, idx_d, W_d, b_d, Pf_d = dev(xyz), dev(new_xyz), dev(idx.to(torch.int32)), dev(W), dev(bias), dev(Pf)
    x = torch.empty(B * S * K, ch, device=DEV)
    a = L.fill(L.S3dGroupProjArgs(), xyz=xyz_d, new_xyz=nx_d, idx=idx_d, B=B, N=N, S=S, K=K, C=ch, W=W_d, ldw=3 + C, bias=b_d,
               Pf=Pf_d, ldp=ch, x=x, ldx=ch)
    sums = torch.zeros(2 * ch, dtype=torch.float64, device=DEV)         # optional: BatchNorm statistics of x, accumulated on the fly
    L.fill(a, sums=sums)
    L.check(L.lib().s3d_group_project_fwd(ctypes.byref(a), L.current_stream()), 'gp fwd')
    assert rel_err(x, ref.detach()) < 2e-6
    assert rel_err(sums[:ch], x.double().sum(0)) < 1e-9 and rel_err(sums[ch:], (x.double() ** 2).sum(0)) < 1e-9
    dx = torch.randn(B * S * K, ch, generator=g).to(torch.bfloat16)
    ref.backward(dx.double())
    dx_d = dev(dx)
    # transposed neighbour lists: for every point the ascending entries e = s*K + j that reference it
    inv_off = torch.empty(B, N + 1, dtype=torch.int32, device=DEV); inv_rows = torch.empty(B, S * K, dtype=torch.int32, device=DEV)
    L.check(L.lib().s3d_neighbor_csr(L.ptr(idx_d), B, N, S, K, L.ptr(inv_off), L.ptr(inv_rows), L.current_stream()), 'csr')
    flat = idx.reshape(B, S * K)
    for b in range(B):
        order = torch.sort(flat[b], stable=True).indices                 # ascending point, ties in entry order
        assert torch.equal(inv_rows[b].cpu().long(), order)
        assert torch.equal(inv_off[b].cpu().long(), torch.cat([torch.zeros(1, dtype=torch.long), torch.bincount(flat[b], minlength=N).cumsum(0)]))
    dPf = torch.full((B * N, ch), float('nan'), device=DEV)              # written, not accumulated
    dW = torch.zeros(ch, 3 + C, device=DEV); db = torch.zeros(ch, device=DEV)
    L.fill(a, dx=dx_d, lddx=ch, dPf=dPf, dW=dW, dbias=db, inv_off=inv_off, inv_rows=inv_rows)
    L.check(L.lib().s3d_group_project_bwd(ctypes.byref(a), L.current_stream()), 'gp bwd')
    assert rel_err(dW[:, :3], Wd.grad[:, :3]) < 1e-4 and float(dW[:, 3:].abs().max()) == 0.0
    assert rel_err(db, bd.grad) < 1e-4
    # dPf: d(loss)/d(Pf) -> the per-point GEMMs' inputs; check through the chain rule: dfeats = dPf Wf, dWf = dPf^T feats
    assert rel_err(dPf.double().cpu() @ W.double()[:, 3:], fd.grad.reshape(B * N, C)) < 1e-4
    assert rel_err(dPf.double().cpu().t() @ feats.double().reshape(B * N, C), Wd.grad[:, 3:]) < 1e-4
    with pytest.raises(RuntimeError, match='multiple of 4'):
        L.check(L.lib().s3d_group_project_fwd(ctypes.byref(L.fill(a, C=ch - 1)), L.current_stream()), 'gp fwd')


def test_pipelined_steps_equal_plain_steps():
    """train_step_pipelined (geometry of the next batch prepared on the side stream during the current step, two geometry sets)
    == the plain train_step sequence on alternating batches, eagerly and through the two captured graphs."""
    kw = dict(backbone='deit_tiny_patch16_224', n_points=64, d_points=22, n_classes=50)
    sd = po.init_state_dict(backbone=kw['backbone'], n_classes=50, d_points=22, seed=3)
    batches = []
    for seed in (4, 5):
        x, y, starts = po.synthetic_points(3, 64, 22, 50, 'seg', seed=seed)
        batches.append((x.to(DEV), y.to(DEV), tuple(s.to(DEV) for s in starts)))
    ref = PointEngine(task='seg', device=DEV, **kw); ref.load_state_dict(sd)
    want = [float(ref.train_step(*batches[i % 2])) for i in range(4)]
    eng = PointEngine(task='seg', device=DEV, **kw); eng.load_state_dict(sd)
    eng.prepare_geometry(batches[0][0], batches[0][2], 0)
    got = []
    for i in range(4):
        (x, y, st), (nx, _, nst) = batches[i % 2], batches[(i + 1) % 2]
        got.append(float(eng.train_step_pipelined(x, y, st, nx, nst, i % 2)))
    assert max(abs(a - b) for a, b in zip(want, got)) <= 2e-3, (want, got)
    assert float((eng.arena.p - ref.arena.p).abs().max()) <= 2e-3
    # captured: two graphs over static buffers, alternating
    eng2 = PointEngine(task='seg', device=DEV, **kw); eng2.load_state_dict(sd)
    xs = [batches[0][0].clone(), batches[1][0].clone()]; ys = [batches[0][1].clone(), batches[1][1].clone()]
    sts = [tuple(t.clone() for t in batches[0][2]), tuple(t.clone() for t in batches[1][2])]
    snap = [t.clone() for t in eng2.train_state()]
    graphs, loss = eng2.capture_train_step_pipelined(xs, ys, sts)
    # the capture's warm-up steps must leave no trace: parameters, momentum buffer, step flag, BatchNorm running statistics
    assert all(torch.equal(a, b) for a, b in zip(eng2.train_state(), snap)), 'capture warm-up leaked into the training state'
    eng2.prepare_geometry(xs[0], sts[0], 0)
    got2 = []
    for i in range(4):
        graphs[i % 2].replay()
        got2.append(float(loss))
    assert max(abs(a - b) for a, b in zip(want, got2)) <= 2e-3, (want, got2)


def test_dp_trainer_pipelined_geometry_equals_plain_steps():
    """PointDataParallelTrainer.prime / step_pipelined (two buffer sets, [top | bottom] graphs per set, geometry of the next batch on
    the side stream inside the top graph) == the plain train_step sequence on alternating batches."""
    from simple3d_former_amd.parallel import PointDataParallelTrainer
    kw = dict(backbone='deit_tiny_patch16_224', n_points=64, d_points=22, n_classes=50)
    sd = po.init_state_dict(backbone=kw['backbone'], n_classes=50, d_points=22, seed=3)
    batches = []
    for seed in (4, 5):
        x, y, starts = po.synthetic_points(3, 64, 22, 50, 'seg', seed=seed)
        batches.append((x.to(DEV), y.to(DEV), tuple(s.to(DEV) for s in starts)))
    ref = PointEngine(task='seg', device=DEV, **kw); ref.load_state_dict(sd)
    want = [float(ref.train_step(*batches[i % 2])) for i in range(5)]
    eng = PointEngine(task='seg', device=DEV, **kw); eng.load_state_dict(sd)
    tr = PointDataParallelTrainer(eng)
    tr.prime(*batches[0])
    got = [float(tr.step_pipelined(*batches[(i + 1) % 2])) for i in range(5)]
    assert max(abs(a - b) for a, b in zip(want, got)) <= 2e-3, (want, got)
    assert float((eng.arena.p - ref.arena.p).abs().max()) <= 2e-3


def test_captured_graph_follows_lr_and_bn_momentum_schedules():
    """train_partseg.py:121-130 decays the learning rate and the BatchNorm momentum every epoch; both live in device memory
    (PointEngine.hyper), so a graph captured once keeps following them -- and capturing mid-training applies no update."""
    kw = dict(backbone='deit_tiny_patch16_224', d_points=6, n_classes=5)
    sd = po.init_state_dict(seed=3, **kw)
    eng = PointEngine(task='cls', device=DEV, n_points=64, **kw); eng.load_state_dict(sd)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 64, 6, generator=g).to(DEV); y = torch.randint(0, 5, (2,), generator=g).to(DEV)
    sts = tuple(torch.randint(0, n, (2,), generator=g).to(DEV) for n in (64, 64))
    before = [t.clone() for t in eng.train_state()]
    graph, loss = eng.capture_train_step(x, y, sts)
    assert all(torch.equal(a, b) for a, b in zip(eng.train_state(), before)), 'capture warm-up leaked into the training state'
    eng.set_lr(0.0)                                   # lr = 0: a replay must leave the parameters alone
    p0 = eng.arena.p.clone()
    graph.replay(); torch.cuda.synchronize()
    assert torch.equal(eng.arena.p, p0)
    rm0 = eng.bn_buffers()[0].clone()
    eng.set_bn_momentum(0.0)                          # momentum = 0: running statistics frozen
    graph.replay(); torch.cuda.synchronize()
    rm1 = eng.bn_buffers()[0].clone()
    graph.replay(); torch.cuda.synchronize()
    assert torch.equal(eng.bn_buffers()[0], rm1)
    eng.set_lr(0.01); eng.set_bn_momentum(0.1)
    graph.replay(); torch.cuda.synchronize()
    assert not torch.equal(eng.arena.p, p0) and not torch.equal(eng.bn_buffers()[0], rm1)
    assert eng.lr == 0.01 and eng.bn_momentum == 0.1 and eng.grad_scale == 1.0


@pytest.mark.parametrize('B,N,C', [(3, 64, 48), (128, 1024, 48), (2, 100, 192), (2, 50, 320)])
def test_mean_points_and_broadcast(B, N, C):
    """x.mean(1) (models/3DViT/model.py:325) and its backward (broadcast / N); C > 256 walks the column chunks."""
    g = torch.Generator().manual_seed(B + N + C)
    x = torch.randn(B, N, C, generator=g)
    xd = x.to(DEV)
    out = torch.empty(B, C, device=DEV)
    L.check(L.lib().s3d_mean_points(L.ptr(xd), B, N, C, L.ptr(out), L.current_stream()), 'mean_points')
    assert rel_err(out, x.double().mean(1)) < 1e-5
    d = torch.randn(B, C, generator=g).to(DEV)
    y = torch.empty(B * N, C, device=DEV)
    L.check(L.lib().s3d_bcast_rows(L.ptr(d), N, C, ctypes.c_long(B * N), ctypes.c_float(1.0 / N), L.ptr(y), L.current_stream()), 'bcast')
    assert rel_err(y.view(B, N, C), (d.cpu() / N)[:, None, :].expand(B, N, C)) < 1e-6


def test_geometry_on_the_main_stream_gives_the_same_result(monkeypatch):
    """S3D_POINT_GEOM_STREAM=0 (FPS / kNN inline on the current stream) and the side-stream default must agree bit for bit: the
    same kernels on the same inputs, only their placement differs."""
    from simple3d_former_amd import point_engine as pe_mod
    z, cfg, sd, x, y, starts = load_point_case('pts_seg_tiny_n64_b2')
    outs = []
    for on in (True, False):
        monkeypatch.setattr(pe_mod, 'GEOM_STREAM', on)
        eng = PointEngine(backbone=cfg['backbone'], n_points=64, d_points=22, n_classes=50, task='seg', device=DEV)
        eng.load_state_dict(sd)
        sts = tuple(s.to(DEV) for s in starts)
        logits = eng.forward(x.to(DEV), sts).clone()
        eng.cross_entropy(cfg['batch'], y.to(DEV)); eng.zero_grad(); eng.backward(cfg['batch'])
        outs.append((logits, [eng.workspace(cfg['batch']).td[i].idx.clone() for i in range(2)], eng.arena.grad('fc1.0.weight').clone()))
        assert (eng._side is not None) == on if hasattr(eng, '_side') else not on
    assert torch.equal(outs[0][0], outs[1][0])
    assert all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))
    # the backward is not bit-reproducible run to run (fp32 atomics in the split-K wgrads; a flipped bf16 rounding downstream): same bar
    # as between two data-parallel replicas
    assert float((outs[0][2] - outs[1][2]).abs().max()) <= 2e-2 * float(outs[0][2].abs().max())


@pytest.mark.parametrize('rows,D,C', [(128, 48, 50), (4100, 96, 50), (70000, 48, 13), (33, 192, 40)])
def test_am_softmax_row_head_kernels(rows, D, C):
    """s3d_l2norm_rows_fwd / _bwd + s3d_am_weight_fwd / _bwd around fp64 matmuls == AMSoftmaxLayer.forward and its autograd
    (models/3DViT/model.py:134-142) on [rows][D] features, W [D][C]."""
    g = torch.Generator().manual_seed(31)
    x = torch.randn(rows, D, generator=g) * 3
    W = torch.randn(D, C, generator=g)
    dl = torch.randn(rows, C, generator=g)
    xr, Wr = x.double().requires_grad_(True), W.double().requires_grad_(True)
    xn_ref = xr / torch.norm(xr, p=2, dim=1, keepdim=True).clamp(min=1e-12)
    wn_ref = Wr / torch.norm(Wr, p=2, dim=0, keepdim=True).clamp(min=1e-12)
    logits_ref = xn_ref @ wn_ref * 30.0
    logits_ref.backward(dl.double())
    lib, s = L.lib(), L.current_stream()
    xd, Wd = x.to(DEV), W.to(DEV)
    inv = torch.empty(rows, dtype=torch.float32, device=DEV)
    hi = torch.zeros(rows, D, dtype=torch.bfloat16, device=DEV); lo = torch.zeros_like(hi)
    L.check(lib.s3d_l2norm_rows_fwd(L.ptr(xd), ctypes.c_long(D), ctypes.c_long(rows), D, L.ptr(inv), L.ptr(hi), L.ptr(lo), ctypes.c_long(D), s), 'l2norm fwd')
    assert rel_err(hi.float() + lo.float(), xn_ref.detach()) < 2e-5
    assert rel_err(inv, 1.0 / xr.detach().norm(dim=1)) < 1e-6
    Wl = torch.zeros(C, D, dtype=torch.float32, device=DEV); inv_w = torch.empty(C, dtype=torch.float32, device=DEV)
    L.check(lib.s3d_am_weight_fwd(L.ptr(Wd), D, C, ctypes.c_float(30.0), L.ptr(Wl), D, L.ptr(inv_w), s), 'am weight fwd')
    assert rel_err(Wl, 30.0 * wn_ref.detach().t()) < 1e-6
    # logits / gradients through fp64 matmuls of the kernels' own outputs (the GEMMs themselves have their own tests)
    xn = (hi.float() + lo.float()).double().cpu()
    assert rel_err(xn @ Wl.double().cpu().t(), logits_ref.detach()) < 2e-5
    dxn = (dl.double() @ Wl.double().cpu()).float().to(DEV)                 # Linear dgrad
    dWl = (dl.double().t() @ xn_ref.detach()).float().to(DEV)               # Linear wgrad [C][D]
    dx = torch.empty(rows, D, dtype=torch.float32, device=DEV)
    L.check(lib.s3d_l2norm_rows_bwd(L.ptr(dxn), ctypes.c_long(D), L.ptr(xd), ctypes.c_long(D), L.ptr(inv), ctypes.c_long(rows), D, L.ptr(dx), ctypes.c_long(D), s), 'l2norm bwd')
    assert rel_err(dx, xr.grad) < 1e-5
    L.check(lib.s3d_l2norm_rows_bwd(L.ptr(dxn), ctypes.c_long(D), L.ptr(xd), ctypes.c_long(D), L.ptr(inv), ctypes.c_long(rows), D, L.ptr(dxn), ctypes.c_long(D), s), 'l2norm bwd in place')
    assert torch.equal(dxn, dx)
    dW = torch.zeros(D, C, dtype=torch.float32, device=DEV)
    L.check(lib.s3d_am_weight_bwd(L.ptr(dWl), D, L.ptr(Wd), L.ptr(inv_w), D, C, ctypes.c_float(30.0), L.ptr(dW), s), 'am weight bwd')
    assert rel_err(dW, Wr.grad) < 1e-5


def test_drop_in_seg_module_with_am_softmax_head_and_cls_error():
    """cfg.model.head == 'AMSoftmax' (models/3DViT/model.py:427-428) through the drop-in PointTransformerSeg vs the reference golden;
    the classification model with that head dies in the reference's own forward (:135 unpacks a 3-D shape) -> ValueError here."""
    import types
    import simple3d_former_amd as s3d
    z, cfg, sd, x, y, starts = load_point_case('pts_seg_tiny_n64_am_b2')
    mk = lambda: types.SimpleNamespace(num_point=cfg['n_points'], num_class=cfg['n_classes'], input_dim=cfg['d_points'],
                                       model=types.SimpleNamespace(nblocks=4, nneighbor=16, transformer_dim=512, head='AMSoftmax',
                                                                   transformer_backbone=cfg['backbone'], pretrained=False, name='3DViT'))
    model = s3d.PointTransformerSeg(mk())
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and 'head.W' not in missing
    model = model.to(DEV)
    model.s3d_fps_starts = starts
    pred = model(x.to(DEV))
    assert float(np.abs(pred.detach().cpu().numpy() - z['logits']).max()) <= 1e-3
    loss = F.cross_entropy(pred.reshape(-1, cfg['n_classes']), y.to(DEV).reshape(-1))
    assert abs(float(loss) - float(z['loss'])) <= 1e-3
    loss.backward()
    g = dict(model.named_parameters())['head.W'].grad
    assert g is not None and abs(float(g.double().norm()) - float(z['gnorm/head.W'])) <= 3e-2 * float(z['gnorm/head.W'])
    with pytest.raises(ValueError, match='fails in the reference itself'):
        s3d.PointTransformerCls(mk())


@pytest.mark.parametrize('dev_hyper', [False, True])
def test_sgd_momentum_matches_torch_and_refreshes_planes(dev_hyper):
    """s3d_sgd_step / s3d_sgd_step_dev vs torch.optim.SGD(lr=0.01, momentum=0.9) (train_cls.py:91) over four steps, with the 1 / world
    gradient scale of the data-parallel trainer; the split-bf16 planes follow the parameters, the gradients come back zeroed."""
    g = torch.Generator().manual_seed(17)
    n = 40960
    p0 = torch.randn(n, generator=g)
    q = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([q], lr=0.01, momentum=0.9)
    p = p0.clone().to(DEV); buf = torch.zeros(n, device=DEV); grad = torch.zeros(n, device=DEV)
    hi = torch.zeros(n, dtype=torch.bfloat16, device=DEV); lo = torch.zeros_like(hi)
    steps = torch.zeros(1, dtype=torch.int32, device=DEV)
    hyper = torch.tensor([0.01, 0.9, 0.5], dtype=torch.float32, device=DEV)
    lib, s = L.lib(), L.current_stream()
    for step in range(4):
        gr = torch.randn(n, generator=g)
        q.grad = 0.5 * gr.clone()
        opt.step()
        grad.copy_(gr)
        if dev_hyper:
            L.check(lib.s3d_sgd_step_dev(L.ptr(p), L.ptr(grad), L.ptr(buf), L.ptr(hi), L.ptr(lo), ctypes.c_long(n), L.ptr(hyper), L.ptr(steps), s), 'sgd dev')
        else:
            L.check(lib.s3d_sgd_step(L.ptr(p), L.ptr(grad), L.ptr(buf), L.ptr(hi), L.ptr(lo), ctypes.c_long(n), ctypes.c_float(0.01), ctypes.c_float(0.9),
                                     ctypes.c_float(0.5), L.ptr(steps), s), 'sgd')
        assert rel_err(p, q.detach()) < 1e-6, f'step {step}'
        assert float(grad.abs().max()) == 0.0
        assert rel_err(hi.float() + lo.float(), q.detach()) < 2e-5
    assert int(steps) == 4


def test_point_trained_state_fixture_from_the_reference_sgd():
    """VERDICT r04 item 6, point path: the HIP training step (forward incl. FPS / kNN, CE, backward, SGD + momentum) against the fixture the
    REFERENCE PointTransformerCls produced under torch.optim.SGD(lr = 0.01, momentum = 0.9) (train_cls.py:91,117-123;
    tests/golden/make_golden_points_trained.py): 80 train-mode steps on a learnable batch set.  Train-mode BatchNorm on 8 clouds + momentum is a
    chaotic map (two fp32 implementations drift to ~1e-2 of the loss, tests/test_oracle_points.py), so: the first steps pin the optimizer,
    the whole curve is bounded, and the held-out eval-mode decisions must agree where the reference's top-2 gap is clear of the drift."""
    z = np.load(f'{GOLDEN}/trained_pts_cls_tiny_n64_sgd80.npz')
    cfg = json.loads(str(z['cfg']))
    sd = po.init_state_dict(backbone=cfg['backbone'], n_classes=cfg['n_classes'], d_points=cfg['d_points'], seed=9)
    eng = PointEngine(backbone=cfg['backbone'], n_points=cfg['n_points'], d_points=cfg['d_points'], n_classes=cfg['n_classes'], task='cls', device=DEV)
    eng.load_state_dict(sd)
    data = [po.synthetic_class_points(cfg['batch'], cfg['n_points'], cfg['labels'], seed=600 + i) for i in range(cfg['n_batches'])]
    data = [(x.to(DEV), y.to(DEV), tuple(s.to(DEV) for s in st)) for x, y, st in data]
    dev = []
    for step in range(cfg['steps']):
        x, y, st = data[step % len(data)]
        loss = float(eng.train_step(x, y, st))
        ref = float(z['losses'][step])
        dev.append(abs(loss - ref) / max(abs(ref), 0.05))
        if step < 5:
            assert abs(loss - ref) <= 5e-3 * max(1.0, abs(ref)), f'step {step}: HIP loss {loss:.5f} vs reference {ref:.5f}'
    xh, yh, sth = po.synthetic_class_points(cfg['held_batch'], cfg['n_points'], cfg['labels'], seed=999)
    logits = eng.forward(xh.to(DEV), tuple(s.to(DEV) for s in sth), training=False).cpu().numpy()
    clear = z['held_top2_gap'] > 1.5
    agree = int((logits.argmax(1)[clear] == z['held_argmax'][clear]).sum())
    print(f'point trained-state fixture: relative loss deviation by step (every 10th) {[round(d, 4) for d in dev[::10]]}, worst {max(dev):.3f}; '
          f'held-out decisions equal on {agree}/{int(clear.sum())} clear samples, logits within {float(np.abs(logits - z["held_logits"]).max()):.2f}')
    assert max(dev) <= 0.10, max(dev)            # measured 0.035 (the two fp32 CPU implementations: 0.011)
    assert agree >= int(clear.sum()) - 1
