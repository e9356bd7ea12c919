"""Shared helpers for the parity tests (golden loading, parameter regeneration)."""
import json
import os

import numpy as np
import torch

from oracle import voxel_oracle as vo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
MODEL_KEYS = ('backbone', 'embed_layer', 'voxel_size', 'cell', 'patch', 'n_classes', 'pos_embedding', 'head')
FWD_KEYS = ('backbone', 'embed_layer', 'cell', 'patch', 'pos_embedding')

VOXEL_CASES = ['cfg1_small_v30_b8', 'cfg2_small_v32_b4', 'tiny_v12_default_b3', 'tiny_v12_noavg_default_b2',
               'tiny_v12_naive_b2', 'small_v30_amsoftmax_b4', 'tiny_v12_group_b3', 'cfg3_base_v128_group_b1']


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    cfg = json.loads(str(z['cfg']))
    return z, cfg


def rebuild_inputs(cfg, z=None):
    """Regenerates the exact parameter dict + synthetic batch the golden generator used and,
    when the fixture is given, proves it via the stored parameter fingerprint."""
    sd = vo.init_state_dict(seed=9, exercise_all=True, portable=True, **{k: cfg[k] for k in MODEL_KEYS})
    if z is not None:
        keys = sorted(sd)
        fp = np.array([[float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in keys])
        np.testing.assert_allclose(fp, z['fingerprint'], rtol=1e-12, atol=1e-12,
                                   err_msg='regenerated parameters differ from the golden run (RNG drift?)')
    x, y = vo.synthetic_batch(cfg['batch'], cfg['voxel_size'], cfg['n_classes'], seed=9, portable=True)
    return sd, x, y


def fwd_kwargs(cfg):
    return {k: cfg[k] for k in FWD_KEYS}


def check_grads_against_golden(z, grads, rtol, atol, names=None, skip=()):
    """grads: {name: tensor}.  For every parameter compares (a) the gradient norm, (b) the RMS error over the sampled
    entries and (c) the worst sampled entry, all relative to the rms of the reference gradient:
        |norm - ref| <= 10*rtol*ref,   rms_err <= 10*rtol*rms,   max_err <= 50*rtol*rms   (+ atol)
    (rtol=3e-3 -> 3 % / 3 % / 15 %: the plain-bf16 backward bar; rtol=1e-4 for the fp32 oracle).
    Small tensors (<= 4096 entries) are also compared in full: rms with the same bound, the worst of ALL entries with
    60*rtol -- the error of a plain-bf16 backward is noise with sigma ~ the rms bound, and the largest of 4096 draws sits at
    ~4.1 sigma, i.e. AT 40*rtol when the rms error is at its own bound (observed: 0.118 .. 0.135 of the gradient rms on the
    n = 1024 / 2048 point fixtures, with either formulation of the set-abstraction backward).  The ~100 sampled entries get
    50*rtol.  ADVICE r02 asked whether that allowance covers atomic-order noise or the bf16 rounding of dy1 in the TransitionDown
    backward: the point goldens now run in deterministic mode (tests/test_gpu_points.py: deterministic_reductions) and the first
    layer's weight gradient (fc1.0.weight of pts_seg_tiny_n2048_b1, every error of the backward pass summed over B*N rows) still
    sits at 0.128 of the gradient rms there -- it is rounding of the bf16 backward operands, not run-to-run noise, so 40*rtol
    (0.12) is simply below what a plain-bf16 backward delivers on that tensor."""
    gold_names = json.loads(str(z['grad_names']))
    worst = 0.0
    for k in (names or gold_names):
        assert k in grads, f'missing gradient for {k}'
        if k in skip:
            continue
        g = grads[k].detach().float().cpu().flatten()
        ref_norm = float(z['gnorm/' + k])
        idx = torch.from_numpy(z['gidx/' + k])
        got = g[idx].numpy().astype(np.float64)
        ref = z['gval/' + k].astype(np.float64)
        scale = max(ref_norm / max(g.numel(), 1) ** 0.5, 1e-12)     # rms of the reference grad
        diff = np.abs(got - ref)
        rms_err, max_err = float(np.sqrt((diff ** 2).mean())), float(diff.max())
        worst = max(worst, max_err / scale)
        assert rms_err <= atol + 10 * rtol * scale, f'{k}: sampled grad rms err {rms_err:.3e} (grad rms {scale:.3e})'
        assert max_err <= atol + 50 * rtol * scale, f'{k}: sampled grad max err {max_err:.3e} (grad rms {scale:.3e})'
        assert abs(float(g.double().norm()) - ref_norm) <= atol * max(g.numel(), 1) ** 0.5 + rtol * ref_norm * 10 + 1e-12, \
            f'{k}: grad norm {float(g.norm()):.6e} vs {ref_norm:.6e}'
        if ('gfull/' + k) in z.files:
            full = grads[k].detach().float().cpu().numpy().reshape(z['gfull/' + k].shape).astype(np.float64)
            d = np.abs(full - z['gfull/' + k])
            assert float(np.sqrt((d ** 2).mean())) <= atol + 10 * rtol * scale, f'{k}: full-tensor rms err'
            assert float(d.max()) <= atol + 60 * rtol * scale, f'{k}: full-tensor max err {d.max():.3e} (rms {scale:.3e})'
    return worst


def check_grads_against_oracle(grads, ref_grads, rtol, atol=1e-8, tile=64, loose=None):
    """Every gradient tensor of the HIP path against the CPU oracle's autograd gradient of the SAME step, in full (no sampling).
    With rms = the rms of the reference tensor:
      (a) |norm - ref norm| <= 10*rtol * ref norm;
      (b) rms error over ALL entries <= 10*rtol * rms;
      (c) rms error of every tile x tile block (of the 2-D view) <= 25*rtol * max(rms, rms of the reference block): a wrong or missing
          GEMM tile / k-slice / bias segment is a LOCAL error that a whole-tensor rms can hide, a per-block rms cannot;
      (d) every entry: |err| <= 80*rtol * max(rms, |ref entry|) -- the error of a plain-bf16 backward is rounding noise with
          sigma <= the rms bound, the largest of 6e5 draws sits at ~5 sigma, and tensors of mixed scale (the class-token row of a
          positional embedding, the xyz columns of a set-abstraction convolution) carry noise in proportion to the LOCAL magnitude;
      (e) no bias: alpha = <got, ref> / <ref, ref> within 5*rtol + 4 * (rms err / rms) / sqrt(entries) of 1 -- zero-mean rounding
          noise leaves alpha alone, missing rows / a wrong scale / a dropped k-slice move it.
    loose = {name: factor}: tensors whose bars are multiplied by `factor` (the caller says why).
    Returns {name: (rms_err / rms, worst block, worst entry, alpha)}; all violations are reported together."""
    out, bad = {}, []
    missing = [k for k in ref_grads if k not in grads]
    assert not missing, f'no gradient for {missing}'
    pad = torch.nn.functional.pad
    for k, ref_t in ref_grads.items():
        f = (loose or {}).get(k, 1.0)
        ref = ref_t.detach().double().cpu().reshape(-1)
        got = grads[k].detach().double().cpu().reshape(-1)
        assert got.numel() == ref.numel(), f'{k}: {got.numel()} vs {ref.numel()} entries'
        rms = max(float(ref.pow(2).mean().sqrt()), 1e-30)
        d = got - ref
        rms_err = float(d.pow(2).mean().sqrt())
        cols = ref_t.shape[-1] if ref_t.dim() >= 2 else ref.numel()
        d2, r2 = d.reshape(-1, cols), ref.reshape(-1, cols)
        R, C = d2.shape
        pr, pc = (-R) % tile if R > 1 else 0, (-C) % tile
        tr = tile if R > 1 else 1
        blocks = lambda t: pad(t, (0, pc, 0, pr)).reshape((R + pr) // tr, tr, (C + pc) // tile, tile).sum((1, 3))
        cnt = blocks(torch.ones_like(d2)).clamp_min(1)
        blk_err, blk_ref = (blocks(d2 ** 2) / cnt).sqrt(), (blocks(r2 ** 2) / cnt).sqrt()
        blk = float((blk_err / blk_ref.clamp_min(rms)).max())
        worst = float((d.abs() / ref.abs().clamp_min(rms)).max())
        alpha = float((got * ref).sum() / (ref * ref).sum().clamp_min(1e-60))
        out[k] = (rms_err / rms, blk, worst, alpha)
        if rms_err > atol + 10 * rtol * f * rms:
            bad.append(f'{k}: grad rms err {rms_err:.3e} = {rms_err / rms:.4f} of the grad rms (bar {10 * rtol * f:.3f})')
        if abs(float(got.norm()) - float(ref.norm())) > atol * ref.numel() ** 0.5 + 10 * rtol * f * float(ref.norm()):
            bad.append(f'{k}: grad norm {float(got.norm()):.6e} vs {float(ref.norm()):.6e}')
        if blk > 25 * rtol * f + atol / rms:
            bad.append(f'{k}: worst {tile}x{tile} block rms err {blk:.4f} of its scale (bar {25 * rtol * f:.3f})')
        if worst > 80 * rtol * f + atol / rms:
            bad.append(f'{k}: worst entry err {worst:.4f} of its scale (bar {80 * rtol * f:.3f})')
        if abs(alpha - 1) > 5 * rtol * f + 4 * (rms_err / rms) / ref.numel() ** 0.5 + atol / rms:
            bad.append(f'{k}: regression coefficient alpha = {alpha:.5f} (rms err {rms_err / rms:.4f}, {ref.numel()} entries)')
    top = sorted(out.items(), key=lambda kv: -kv[1][0])[:6]
    print('   largest gradient errors (rms / worst block / worst entry / alpha): ' +
          '; '.join(f'{k} {v[0]:.4f}/{v[1]:.4f}/{v[2]:.3f}/{v[3]:.4f}' for k, v in top))
    assert not bad, '\n'.join(bad)
    return out


def release_graphs(trainer):
    """Drops a data-parallel trainer's captured HIP graphs (they may hold captured collectives)."""
    cap = getattr(trainer, '_cap', None)
    if isinstance(cap, dict):
        cap.clear()


def teardown_process_group(*holders):
    """destroy_process_group() of a single-process RCCL group at the end of a GPU test, in an order that cannot abort the interpreter: trainers
    that hold HIP graphs with CAPTURED collectives are released first and the device is idle.  (Once in ~10 full-suite runs of round 6 the
    teardown of such a group raised SIGABRT inside destroy_process_group -- with `pytest -x` that is the end of the whole GPU suite, not of a
    test.)  `holders`: dicts / lists whose contents (trainers, engines) should be dropped before the group goes."""
    import gc
    import torch.distributed as dist
    for h in holders:
        h.clear()
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if dist.is_initialized():
        dist.destroy_process_group()
