"""Training-trajectory parity (the "cls-acc parity" half of BASELINE.json's metric): 50 fused HIP training steps of the cfg-2
model (deit_small + VoxelEmbed 32^3, 40 classes) at batch 8 against the CPU oracle's fp32 forward / autograd backward / Adam on
the SAME batches (train_cls_voxel.py:275-288), in deterministic mode (s3d_set_deterministic: no split-K atomics) so that the
run is reproducible bit for bit."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import simple3d_former_amd as s3d
    from simple3d_former_amd import _lib as L

from oracle import voxel_oracle as vo

DEV = 'cuda'
KW = dict(backbone='deit_small_patch16_224', embed_layer='VoxelEmbed', voxel_size=32, cell=6, patch=5, n_classes=40)
FKW = dict(backbone=KW['backbone'], embed_layer='VoxelEmbed', cell=6, patch=5)


@pytest.fixture
def deterministic():
    lib = L.lib()
    lib.s3d_set_deterministic(1)
    yield
    lib.s3d_set_deterministic(0)


def _engine(sd):
    eng = s3d.VoxelEngine(device=DEV, lr=1e-3, **KW)           # README recipe: Adam, lr 1e-3
    eng.load_state_dict(sd)
    return eng


def test_deterministic_mode_makes_the_training_step_bitwise_reproducible(deterministic):
    sd = vo.init_state_dict(seed=9, exercise_all=True, **KW)
    x, y = vo.synthetic_batch(8, 32, 40, seed=11)
    runs = []
    for _ in range(2):
        eng = _engine(sd)
        losses = [float(eng.train_step(x.to(DEV), y.to(DEV))) for _ in range(6)]
        runs.append((losses, eng.arena.p.clone()))
    assert runs[0][0] == runs[1][0], 'losses differ between two identical runs'
    assert torch.equal(runs[0][1], runs[1][1]), 'parameters differ between two identical runs'


def test_update_in_slices_beside_the_backward_equals_the_single_update(deterministic):
    """s3d_adam_begin + s3d_adam_apply per arena slice on a second stream (VoxelEngine.update_slices) == one s3d_adam_step behind
    the backward, bit for bit, eagerly and as a captured graph."""
    sd = vo.init_state_dict(seed=9, exercise_all=True, **KW)
    x, y = vo.synthetic_batch(8, 32, 40, seed=11)
    x, y = x.to(DEV), y.to(DEV)
    runs = []
    for slices, graph in ((0, False), (3, False), (4, True)):
        eng = _engine(sd)
        eng.update_slices = slices
        if graph:
            g, sx, sy, loss = eng.capture_train_step(8)
            sx.copy_(x); sy.copy_(y)
            losses = []
            for _ in range(5):
                g.replay()
                losses.append(float(loss))
        else:
            losses = [float(eng.train_step(x, y)) for _ in range(5)]
        torch.cuda.synchronize()
        runs.append((losses, eng.arena.p.clone(), eng.arena.hi.clone(), eng.arena.g.clone()))
    for r in runs[1:]:
        assert r[0] == runs[0][0]
        assert torch.equal(r[1], runs[0][1]) and torch.equal(r[2], runs[0][2]) and not r[3].any()


@pytest.mark.parametrize('B', [8, 64])
def test_optimizer_update_inside_the_backward_launches_equals_the_single_update(deterministic, B):
    """S3dAdamFill (csrc/adam_fill.h): the Adam update of block i + 1's GEMM parameters rides on block i's backward launches as filler
    workgroups, the rest is one s3d_adam_apply_ranges launch == s3d_adam_step behind the backward, bit for bit (parameters, moments,
    weight planes, zeroed gradients), eagerly and as a captured graph.  B = 64: the LDS-DMA pair kernels / LayerNorm / attention
    backward carry the shares; B = 8: the register-staged pair kernels cannot, their shares are drained as plain launches."""
    sd = vo.init_state_dict(seed=9, exercise_all=True, **KW)
    x, y = vo.synthetic_batch(B, 32, 40, seed=11)
    x, y = x.to(DEV), y.to(DEV)
    runs = []
    for fill, graph in ((False, False), (True, False), (True, True)):
        eng = _engine(sd)
        eng.adam_fill = fill
        if graph:
            g, sx, sy, loss = eng.capture_train_step(B)
            sx.copy_(x); sy.copy_(y)
            losses = []
            for _ in range(4):
                g.replay()
                losses.append(float(loss))
        else:
            losses = [float(eng.train_step(x, y)) for _ in range(4)]
        torch.cuda.synchronize()
        if fill:
            st = eng.adam_fill_stats
            # 11 of 12 blocks' GEMM parameters on the paired launches; on the dgrad chain (round 5) a block's share rides on the grouped wgrad
            # launch AFTER the one that computes its gradients, so the last group of blocks is left to the final launch: 9 of 12
            assert st['filled'] + st['rest'] == eng.arena.numel and st['filled'] > 0.70 * eng.arena.numel, st
        runs.append((losses, eng.arena.p.clone(), eng.arena.m.clone(), eng.arena.v.clone(), eng.arena.hi.clone(), eng.arena.lo.clone(), eng.arena.g.clone()))
        assert eng.optimizer_state()['step'] == 4
    for r in runs[1:]:
        assert r[0] == runs[0][0]
        for a, b in zip(r[1:6], runs[0][1:6]):
            assert torch.equal(a, b)
        assert not r[6].any()


def test_fifty_step_trajectory_tracks_the_oracle(deterministic):
    steps, B, nb = 50, 8, 5
    sd = vo.init_state_dict(seed=9, **KW)                       # the reference's own initialisation
    batches = [vo.synthetic_batch(B, 32, 40, seed=100 + i) for i in range(nb)]       # five batches, cycled (ten "epochs")
    x_held, y_held = vo.synthetic_batch(32, 32, 40, seed=999)
    eng = _engine(sd)
    names = vo.used_param_names(sd)
    ref = {k: v.clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(ref[k]) for k in names}
    v = {k: torch.zeros_like(ref[k]) for k in names}
    worst = 0.0
    got_curve, ref_curve = [], []
    for step in range(1, steps + 1):
        x, y = batches[(step - 1) % nb]
        loss = float(eng.train_step(x.to(DEV), y.to(DEV)))
        _, loss_ref, grads = vo.loss_and_grads(ref, x, y, **FKW)
        for k, g in grads.items():
            vo.adam_step(ref[k], g, m[k], v[k], step)
        loss_ref = float(loss_ref)
        got_curve.append(loss); ref_curve.append(loss_ref)
        rel = abs(loss - loss_ref) / max(abs(loss_ref), 1e-6)
        worst = max(worst, rel)
        assert rel <= 5e-3, f'step {step}: HIP loss {loss:.5f} vs oracle {loss_ref:.5f} (rel {rel:.3e})'
    assert ref_curve[-1] < 0.9 * ref_curve[0], f'the oracle run did not train: {ref_curve[0]:.3f} -> {ref_curve[-1]:.3f}'
    # held-out synthetic batch: same class decisions, hence the same accuracy
    logits = eng.forward(x_held.to(DEV)).cpu()
    with torch.no_grad():
        ref_logits = vo.forward(ref, x_held, **FKW)
    acc = float((logits.argmax(1) == y_held).float().mean())
    acc_ref = float((ref_logits.argmax(1) == y_held).float().mean())
    top2 = ref_logits.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 5e-2                    # decisions that are not inside the trajectory noise
    assert int(clear.sum()) >= 24, 'held-out batch has too few clear-cut decisions to be meaningful'
    assert torch.equal(logits.argmax(1)[clear], ref_logits.argmax(1)[clear]), 'class decisions differ on clear-cut held-out samples'
    assert abs(acc - acc_ref) <= 1.0 / 32 + 1e-9, (acc, acc_ref)
    print(f'50-step trajectory: worst relative loss deviation {worst:.3e}; loss {ref_curve[0]:.3f} -> {ref_curve[-1]:.3f}; '
          f'held-out accuracy {acc:.3f} (oracle {acc_ref:.3f}), {int(clear.sum())}/32 clear-cut decisions all equal')


def test_trained_state_fixture_from_the_reference_optimizer():
    """the HIP fused training step against a fixture produced by the REFERENCE model trained with the reference's own
    optimizer (torch.optim.Adam(lr = 1e-3), train_cls_voxel.py:195,277-288; tests/golden/make_golden_trained.py): 60 steps on a fixed learnable
    batch set in cfg-1 geometry.  The loss trajectory is reproduced step by step and the held-out class decisions -- 6 distinct classes,
    logits that depend on the input -- are identical wherever the reference's top-2 gap exceeds 2e-3 (all 32 here)."""
    import json
    import numpy as np
    from tests._util import GOLDEN
    z = np.load(f'{GOLDEN}/trained_cfg1_small_v30_adam60.npz')
    cfg = json.loads(str(z['cfg']))
    kw = {k: cfg[k] for k in ('backbone', 'embed_layer', 'voxel_size', 'cell', 'patch', 'n_classes', 'pos_embedding', 'head')}
    sd = vo.init_state_dict(seed=9, exercise_all=False, portable=True, **kw)
    eng = s3d.VoxelEngine(device=DEV, lr=cfg['lr'], **kw)
    eng.load_state_dict(sd)
    data = [vo.synthetic_class_batch(cfg['batch'], cfg['voxel_size'], cfg['cell'], cfg['labels'], seed=500 + i) for i in range(cfg['n_batches'])]
    data = [(x.to(DEV), y.to(DEV)) for x, y in data]
    worst = 0.0
    for step in range(cfg['steps']):
        x, y = data[step % len(data)]
        loss = float(eng.train_step(x, y))
        ref = float(z['losses'][step])
        rel = abs(loss - ref) / max(abs(ref), 1e-6)
        worst = max(worst, rel)
        assert rel <= 5e-3, f'step {step}: HIP loss {loss:.5f} vs reference {ref:.5f} (rel {rel:.2e})'
    xh, yh = vo.synthetic_class_batch(cfg['held_batch'], cfg['voxel_size'], cfg['cell'], cfg['labels'], seed=999)
    logits = eng.forward(xh.to(DEV)).cpu().numpy()
    clear = z['held_top2_gap'] > 2e-3
    assert int(clear.sum()) >= 24 and len(set(z['held_argmax'].tolist())) >= 6
    np.testing.assert_array_equal(logits.argmax(1)[clear], z['held_argmax'][clear])
    err = float(np.abs(logits - z['held_logits']).max())
    print(f'trained-state fixture: worst relative loss deviation over 60 steps {worst:.2e}; held-out logits within {err:.2e}; '
          f'{int(clear.sum())}/32 decisions ({len(set(z["held_argmax"].tolist()))} classes) equal')
    assert err <= 0.05 * float(np.abs(z['held_logits']).max()), err          # 60 plain-bf16 backward steps apart: the trajectory bound, not the 1e-3 forward bar
