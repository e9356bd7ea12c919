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
            # launch AFTER the one that computes its gradients, so the last group of four blocks is left to the final launch: 8 of 12
            # (all twelve blocks are on the chain since the last block runs dense there)
            assert st['filled'] + st['rest'] == eng.arena.numel and st['filled'] > 0.60 * eng.arena.numel, st
        runs.append((losses, eng.arena.p.clone(), eng.arena.m.clone(), eng.arena.v.clone(), eng.arena.hi.clone(), eng.arena.lo.clone(), eng.arena.g.clone()))
        assert eng.optimizer_state()['step'] == 4
    for r in runs[1:]:
        assert r[0] == runs[0][0]
        for a, b in zip(r[1:6], runs[0][1:6]):
            assert torch.equal(a, b)
        assert not r[6].any()


def test_trajectory_tracks_the_oracle(deterministic):
    """50 fused HIP steps against the oracle's forward / autograd / adam_step on the same batches (the stable-regime fixture below pins 400
    steps against the reference itself)."""
    steps, B, nb = 50, 8, 5
    sd = vo.init_state_dict(seed=9, **KW)                       # the reference's own initialisation
    batches = [vo.synthetic_batch(B, 32, 40, seed=100 + i) for i in range(nb)]       # five batches, cycled (ten "epochs")
    x_held, y_held = vo.synthetic_batch(32, 32, 40, seed=999)
    eng = _engine(sd)
    got_curve = [float(eng.train_step(*(t.to(DEV) for t in batches[(step - 1) % nb]))) for step in range(1, steps + 1)]
    names = vo.used_param_names(sd)
    ref = {k: v.clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(ref[k]) for k in names}
    v = {k: torch.zeros_like(ref[k]) for k in names}
    ref_curve = []
    for step in range(1, steps + 1):
        x, y = batches[(step - 1) % nb]
        _, loss_ref, grads = vo.loss_and_grads(ref, x, y, **FKW)
        for k, g in grads.items():
            vo.adam_step(ref[k], g, m[k], v[k], step)
        ref_curve.append(float(loss_ref))
    with torch.no_grad():
        ref_logits = vo.forward(ref, x_held, **FKW)
    worst = 0.0
    for step, (loss, loss_ref) in enumerate(zip(got_curve, ref_curve), 1):
        rel = abs(loss - loss_ref) / max(abs(loss_ref), 1e-6)
        worst = max(worst, rel)
        assert rel <= 5e-3, f'step {step}: HIP loss {loss:.5f} vs oracle {loss_ref:.5f} (rel {rel:.3e})'
    assert ref_curve[-1] < 0.9 * ref_curve[0], f'the oracle run did not train: {ref_curve[0]:.3f} -> {ref_curve[-1]:.3f}'
    # held-out synthetic batch: same class decisions, hence the same accuracy
    logits = eng.forward(x_held.to(DEV)).cpu()
    acc = float((logits.argmax(1) == y_held).float().mean())
    acc_ref = float((ref_logits.argmax(1) == y_held).float().mean())
    top2 = ref_logits.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 5e-2                    # decisions that are not inside the trajectory noise
    assert int(clear.sum()) >= 24, 'held-out batch has too few clear-cut decisions to be meaningful'
    assert torch.equal(logits.argmax(1)[clear], ref_logits.argmax(1)[clear]), 'class decisions differ on clear-cut held-out samples'
    assert abs(acc - acc_ref) <= 1.0 / 32 + 1e-9, (acc, acc_ref)
    print(f'{steps}-step trajectory: worst relative loss deviation {worst:.3e}; loss {ref_curve[0]:.3f} -> {ref_curve[-1]:.3f}; '
          f'held-out accuracy {acc:.3f} (oracle {acc_ref:.3f}), {int(clear.sum())}/32 clear-cut decisions all equal')


def _trained_fixture_run(precise):
    import json
    import numpy as np
    from tests._util import GOLDEN
    z = np.load(f'{GOLDEN}/trained_cfg1_small_v30_adam60.npz')
    cfg = json.loads(str(z['cfg']))
    kw = {k: cfg[k] for k in ('backbone', 'embed_layer', 'voxel_size', 'cell', 'patch', 'n_classes', 'pos_embedding', 'head')}
    sd = vo.init_state_dict(seed=9, exercise_all=False, portable=True, **kw)
    eng = s3d.VoxelEngine(device=DEV, lr=cfg['lr'], precise_backward=precise, **kw)
    eng.load_state_dict(sd)
    data = [vo.synthetic_class_batch(cfg['batch'], cfg['voxel_size'], cfg['cell'], cfg['labels'], seed=500 + i) for i in range(cfg['n_batches'])]
    data = [(x.to(DEV), y.to(DEV)) for x, y in data]
    losses = []
    for step in range(cfg['steps']):
        x, y = data[step % len(data)]
        losses.append(float(eng.train_step(x, y)))
    xh, yh = vo.synthetic_class_batch(cfg['held_batch'], cfg['voxel_size'], cfg['cell'], cfg['labels'], seed=999)
    logits = eng.forward(xh.to(DEV)).cpu().numpy()
    rel = np.abs(np.array(losses) - z['losses']) / np.maximum(np.abs(z['losses']), 1e-6)
    return z, rel, np.array(losses), logits, yh.numpy()


def test_trained_state_fixture_from_the_reference_optimizer():
    """the HIP fused training step against a fixture produced by the REFERENCE model trained with the reference's own
    optimizer (torch.optim.Adam(lr = 1e-3), train_cls_voxel.py:195,277-288; tests/golden/make_golden_trained.py): 60 steps on a fixed learnable
    batch set in cfg-1 geometry, split-precision backward (the parity mode).  Adam at lr 1e-3 from the initialisation is an unstable regime
    (the reference's loss goes 3.65, 2.54, 4.18, 3.92 in its first steps): a perturbation of 1e-7 grows to 2e-4 over the 60 steps (the fp32
    CPU oracle, tests/test_oracle_golden.py), the split-precision HIP backward ends within 6e-3 -- every step, all 60 -- and the held-out
    class decisions (6 distinct classes, logits that depend on the input) are identical wherever the reference's top-2 gap exceeds 2e-3
    (all 32 here)."""
    import numpy as np
    z, rel, _, logits, _ = _trained_fixture_run(precise=True)
    assert float(rel[:20].max()) <= 1e-3, f'first 20 steps: worst relative loss deviation {float(rel[:20].max()):.2e}'
    assert float(rel.max()) <= 1.5e-2, f'worst relative loss deviation {float(rel.max()):.2e} at step {int(rel.argmax())}'
    clear = z['held_top2_gap'] > 2e-3
    assert int(clear.sum()) >= 24 and len(set(z['held_argmax'].tolist())) >= 6
    np.testing.assert_array_equal(logits.argmax(1)[clear], z['held_argmax'][clear])
    err = float(np.abs(logits - z['held_logits']).max())
    print(f'trained-state fixture, split-precision backward: worst relative loss deviation {float(rel[:20].max()):.2e} (steps 0-19) / {float(rel.max()):.2e} (all 60); '
          f'held-out logits within {err:.2e}; {int(clear.sum())}/32 decisions ({len(set(z["held_argmax"].tolist()))} classes) equal')
    assert err <= 0.05 * float(np.abs(z['held_logits']).max()), err


def test_trained_state_fixture_default_backward_trains_alike():
    """The same fixture with the DEFAULT backward (plain bf16 MFMA operands: what bench.py times).  Its gradients carry ~1e-2 of rounding
    noise, which this unstable early phase amplifies (measured: 1.5 % at step 3, +-10 - 20 % from step 22 on, single late steps up to 90 %;
    the deterministic mode walks yet another path, and so does the split-precision backward with the learning rate changed by 1e-3:
    33 %, tools/r5/traj_dev.py) -- so the trajectory is pinned while it can be (steps 0 - 2: optimizer semantics, 5e-3; steps 0 - 19: 4e-2)
    and the run has to train: the mean loss of the last ten steps (1.05 - 1.64 over runs against the reference's 1.22, from 3.65).
    The step-60 held-out snapshot is printed, not asserted: 0.22 - 0.53 over thirteen runs against 0.72 for the reference and 0.53 - 0.72 for
    its lr-perturbed siblings -- 32 samples of a run whose loss still swings by +-30 % per step."""
    import numpy as np
    z, rel, losses, logits, yh = _trained_fixture_run(precise=False)
    tail, tail_ref = float(losses[-10:].mean()), float(z['losses'][-10:].mean())
    acc, acc_ref = float((logits.argmax(1) == yh).mean()), float((z['held_argmax'] == yh).mean())
    print(f'trained-state fixture, default backward: steps 0-2 within {float(rel[:3].max()):.2e}, steps 0-19 within {float(rel[:20].max()):.2e}, all within '
          f'{float(rel.max()):.2e}; last-ten-step loss {tail:.3f} (reference {tail_ref:.3f}); held-out accuracy {acc:.3f} (reference {acc_ref:.3f})')
    assert float(rel[:3].max()) <= 5e-3, f'steps 0 - 2: {rel[:3]}'
    assert float(rel[:20].max()) <= 4e-2, f'steps 0 - 19: worst {float(rel[:20].max()):.2e}'
    assert tail <= 0.6 * float(z['losses'][0]), (tail, tail_ref)


# ----------------------------------------------------------------------------------------------------------------------------------
# Round 6: class-accuracy parity of the BENCHED mode as an asserted fact.
def _stable_fixture_runs(backward, seeds=None):
    import json
    import numpy as np
    from tests._util import GOLDEN
    z = np.load(f'{GOLDEN}/trained_stable_cfg1_small_v30_adam400.npz')
    cfg = json.loads(str(z['cfg']))
    kw = {k: cfg[k] for k in ('backbone', 'embed_layer', 'voxel_size', 'cell', 'patch', 'n_classes', 'pos_embedding', 'head')}
    dk = dict(base=cfg['density_base'], step=cfg['density_step'])
    data = [vo.synthetic_class_batch(cfg['batch'], cfg['voxel_size'], cfg['cell'], cfg['labels'], seed=500 + i, **dk) for i in range(cfg['n_batches'])]
    data = [(x.to(DEV), y.to(DEV)) for x, y in data]
    xh, yh = vo.synthetic_class_batch(cfg['held_batch'], cfg['voxel_size'], cfg['cell'], cfg['labels'], seed=999, **dk)
    assert np.array_equal(yh.numpy(), z['held_target'])
    xh = xh.to(DEV)
    T = cfg['tail']
    out = []
    for seed in (seeds or cfg['seeds']):
        sd = vo.init_state_dict(seed=seed, exercise_all=False, portable=True, **kw)
        eng = s3d.VoxelEngine(device=DEV, lr=cfg['lr'], backward=backward, **kw)
        eng.load_state_dict(sd)
        losses, accs = [], []
        for step in range(cfg['steps']):
            x, y = data[step % len(data)]
            losses.append(eng.train_step(x, y).clone())
            if step + 1 in cfg['checkpoints']:
                accs.append(float((eng.forward(xh).argmax(1).cpu() == yh).float().mean()))
        losses = torch.stack(losses).cpu().numpy().astype(np.float64)
        ref = z[f'losses_{seed}']
        rel = np.abs(losses - ref) / np.maximum(np.abs(ref), 1e-6)
        out.append(dict(seed=seed, acc=float(np.mean(accs)), tail=float(np.median(losses[-T:])), rel=rel,
                        ref_acc=float(z[f'held_acc_{seed}'].mean()), ref_tail=float(np.median(ref[-T:]))))
    return cfg, out


def test_benched_backward_reproduces_the_reference_trained_accuracy_and_final_loss():
    """The mode bench.py times (plain-bf16 backward, non-deterministic launch structure) against the REFERENCE model trained by the reference's
    own torch.optim.Adam in a stable regime (tests/golden/make_golden_trained.py stable: lr 3e-5 = the README recipe under its per-epoch
    warm-up, 400 steps, 12 classes, 256 held-out samples at six checkpoints, TWELVE reference seeds; train_cls_voxel.py:195,277-288,315-329).
    Per-seed outcomes are chaotic (the reference's own seeds spread by +-0.037 in accuracy; the same seed under two fp32 implementations
    differs as much), so the criterion is distributional -- the reference's own seed-to-seed spread:
      * every seed: the first 20 steps track the reference of that seed to 1e-3 (measured 1.9 - 2.6e-4 over six repetitions of the twelve seeds,
        tools/r6/stable_margins.py), the first 50 to 5e-3 (measured 1.0 - 1.6e-3), the first 100 to 0.3 (6 - 10e-2: the divergence is under way);
      * every seed's held-out accuracy inside [min - 2 sigma, max + 2 sigma] of the reference's twelve;
      * the MEAN accuracy over the seeds within 3 standard errors of the difference of the reference's mean (~0.045);
      * the geometric-mean final loss (median of the last 40 steps) within 3 standard errors (in log) of the reference's.
    What sits behind the bars: 40 seeds per mode on an MI355X (profiles/r06_backward_precision_ablation.txt) -- accuracy 0.6738 +- 0.0059
    (this mode), 0.6762 +- 0.0053 (split-precision backward), 0.6775 +- 0.0052 (deterministic mode), 0.6795 +- 0.0058 (the reference's
    arithmetic in fp32): no mode is distinguishable from another."""
    import numpy as np
    cfg, runs = _stable_fixture_runs('bf16')
    acc, ref_acc = np.array([r['acc'] for r in runs]), np.array([r['ref_acc'] for r in runs])
    lt, ref_lt = np.log([r['tail'] for r in runs]), np.log([r['ref_tail'] for r in runs])
    n = len(runs)
    sig = float(ref_acc.std(ddof=1))
    se_acc = float(np.sqrt(ref_acc.var(ddof=1) / n + acc.var(ddof=1) / n))
    se_lt = float(np.sqrt(ref_lt.var(ddof=1) / n + lt.var(ddof=1) / n))
    print(f'stable-regime fixture, benched backward, {n} seeds: held-out accuracy {acc.mean():.4f} (reference {ref_acc.mean():.4f}, its seed-to-seed '
          f'sigma {sig:.3f}, range {ref_acc.min():.3f} .. {ref_acc.max():.3f}; ours {acc.min():.3f} .. {acc.max():.3f}); final loss geo-mean '
          f'{np.exp(lt.mean()):.4f} (reference {np.exp(ref_lt.mean()):.4f}); steps 0-19 within {max(float(r["rel"][:20].max()) for r in runs):.1e}, '
          f'0-99 within {max(float(r["rel"][:100].max()) for r in runs):.1e}')
    assert n >= 12
    for r in runs:
        assert float(r['rel'][:20].max()) <= 1e-3, (r['seed'], float(r['rel'][:20].max()))
        assert float(r['rel'][:50].max()) <= 5e-3, (r['seed'], float(r['rel'][:50].max()))
        assert float(r['rel'][:100].max()) <= 0.3, (r['seed'], float(r['rel'][:100].max()))
        assert ref_acc.min() - 2 * sig <= r['acc'] <= ref_acc.max() + 2 * sig, (r['seed'], r['acc'], ref_acc.min(), ref_acc.max(), sig)
    assert abs(acc.mean() - ref_acc.mean()) <= 3 * se_acc, (acc.mean(), ref_acc.mean(), se_acc)
    assert abs(lt.mean() - ref_lt.mean()) <= 3 * se_lt, (np.exp(lt.mean()), np.exp(ref_lt.mean()), se_lt)


def test_split_backward_tracks_the_reference_seed_by_seed_on_the_stable_fixture():
    """The gradient-parity mode (backward='split') on four of the fixture's seeds: here the trajectory itself is pinned -- the first 100 steps
    within 2e-3 of the reference of that seed (measured <= 5e-4), the same distributional bars as above on the outcome."""
    import numpy as np
    cfg, runs = _stable_fixture_runs('split', seeds=[9, 11, 12, 13])
    for r in runs:
        assert float(r['rel'][:20].max()) <= 2e-5, (r['seed'], float(r['rel'][:20].max()))
        assert float(r['rel'][:100].max()) <= 2e-3, (r['seed'], float(r['rel'][:100].max()))
        assert abs(r['acc'] - r['ref_acc']) <= 0.12, (r['seed'], r['acc'], r['ref_acc'])
    print('stable-regime fixture, split backward: ' + '; '.join(f'seed {r["seed"]}: steps 0-99 within {float(r["rel"][:100].max()):.1e}, accuracy {r["acc"]:.3f} '
                                                               f'(reference {r["ref_acc"]:.3f})' for r in runs))
