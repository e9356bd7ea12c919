"""GPU (MI355X) end-to-end parity of the voxel path against (a) the golden fixtures captured from the reference and
(b) the CPU oracle, plus size-independent properties at BASELINE.json's full cfg-2 size.

Bars (BASELINE.md section 4 / north_star): class indices bit-exact, logits and loss within 1e-3 (absolute) of the
fp32 reference; gradients (plain-bf16 backward) within 3e-2 of the reference gradient rms."""
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import simple3d_former_amd as s3d
    from simple3d_former_amd import _lib as L

from oracle import voxel_oracle as vo
from tests._util import release_graphs, teardown_process_group, MODEL_KEYS, check_grads_against_golden, check_grads_against_oracle, fwd_kwargs, load_case, rebuild_inputs

DEV = 'cuda'
LOGIT_TOL = 1e-3
DEFAULT_CASES = ['cfg1_small_v30_b8', 'cfg2_small_v32_b4', 'tiny_v12_default_b3', 'tiny_v12_noavg_default_b2',
                 'tiny_v12_naive_b2', 'small_v30_amsoftmax_b4',
                 'tiny_v12_group_b3', 'cfg3_base_v128_group_b1']     # group_embed (eval-mode dropout), cfg-3 real geometry


def make_engine(cfg, sd, **kw):
    eng = s3d.VoxelEngine(device=DEV, **{k: cfg[k] for k in MODEL_KEYS}, **kw)
    eng.load_state_dict(sd)
    return eng


@pytest.mark.parametrize('name', DEFAULT_CASES)
def test_engine_matches_reference_golden(name):
    z, cfg = load_case(name)
    sd, x, y = rebuild_inputs(cfg, z)
    eng = make_engine(cfg, sd)
    B = cfg['batch']
    logits = eng.forward(x.to(DEV)).cpu()
    err = float(np.abs(logits.numpy() - z['logits']).max())
    assert err <= LOGIT_TOL, f'logits max abs err {err:.3e} vs reference (bar {LOGIT_TOL})'
    np.testing.assert_array_equal(logits.argmax(1).numpy(), z['argmax'])          # class indices bit-exact
    assert float(z['top2_gap'].min()) > 2 * LOGIT_TOL, 'fixture top-2 gap too small for argmax to be meaningful'
    loss = float(eng.cross_entropy(B, y.to(DEV)))
    assert abs(loss - float(z['loss'])) <= LOGIT_TOL
    eng.zero_grad()
    eng.backward(B)
    grads = {k: eng.arena.grad(k) for k in eng.shapes}
    assert set(grads) == set(json.loads(str(z['grad_names'])))
    worst = check_grads_against_golden(z, grads, rtol=3e-3, atol=1e-7)
    print(f'{name}: logits err {err:.2e}, worst sampled grad err / rms {worst:.3f}')


@pytest.mark.parametrize('name', DEFAULT_CASES)
def test_split_precision_backward_matches_reference_gradients_tightly(name):
    """The tight check of the backward ALGORITHM.  The shipped backward runs on plain-bf16 operands, so against the reference's fp32
    gradients it can only be held to its rounding noise (3 % rms / 15 % worst entry above) -- a small systematic error would pass.
    VoxelEngine(precise_backward=True) runs the SAME launch sequence (capi.hip block_bwd_split: the chain and the epilogues of
    block_bwd, the LayerNorm / token / head kernels unchanged) with every dgrad / wgrad as a three-MFMA split product on hi + lo
    operands without split-K, the attention backward in fp32 and every intermediate gradient as a hi + lo pair: the reference
    gradients (train_cls_voxel.py:282-287, captured from the reference itself) must then be met with the fp32 oracle's own bar,
    check_grads_against_golden(rtol=1e-4) -- 30x tighter.  All eight voxel fixtures: default positional embedding, AM-softmax head,
    group_embed (encoder layer + two passes over the shared blocks) incl. the real cfg-3 geometry."""
    z, cfg = load_case(name)
    sd, x, y = rebuild_inputs(cfg, z)
    eng = make_engine(cfg, sd, precise_backward=True)
    B = cfg['batch']
    logits = eng.forward(x.to(DEV)).cpu()
    assert float(np.abs(logits.numpy() - z['logits']).max()) <= LOGIT_TOL
    eng.cross_entropy(B, y.to(DEV))
    eng.zero_grad()
    eng.backward(B)
    grads = {k: eng.arena.grad(k) for k in eng.shapes}
    worst = check_grads_against_golden(z, grads, rtol=1e-4, atol=1e-7)
    print(f'{name}: split-precision backward, worst sampled grad err / rms {worst:.2e}')


def test_plain_bf16_mode_is_less_accurate_but_close():
    """split=False is the plain-bf16 forward (one MFMA per product): ~1e-2 logit error, which is why the default
    forward is split-bf16."""
    z, cfg = load_case('cfg2_small_v32_b4')
    sd, x, y = rebuild_inputs(cfg, z)
    eng = make_engine(cfg, sd, split=False)
    err = float(np.abs(eng.forward(x.to(DEV)).cpu().numpy() - z['logits']).max())
    assert err < 5e-2
    print('plain bf16 logits err', err)


def test_drop_in_module_training_step_matches_oracle():
    """pred = model(voxel); loss = F.cross_entropy(pred, y); loss.backward(); torch.optim.Adam.step()
    (train_cls_voxel.py:277-288) on the drop-in module vs the CPU oracle."""
    cfg = dict(backbone='deit_tiny_patch16_224', embed_layer='VoxelEmbed', voxel_size=12, cell=4, patch=3, n_classes=10,
               pos_embedding='default', head='default', batch=4)
    sd = vo.init_state_dict(seed=3, exercise_all=True, **{k: cfg[k] for k in MODEL_KEYS})
    x, y = vo.synthetic_batch(4, 12, 10, seed=5)
    model = s3d.Feature3D_ViT2D_V2(embed_layer=s3d.VoxelEmbed(voxel_size=12, cell_size=4, patch_size=3, embed_dim=192),
                                   n_classes=10, transformer_backbone='deit_tiny_patch16_224', pretrained=False,
                                   pos_embedding='default')
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    ref_sd = {k: v.clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v2 = {k: torch.zeros_like(v) for k, v in sd.items()}
    for step in range(1, 4):
        opt.zero_grad()
        pred = model(x.to(DEV))
        loss = F.cross_entropy(pred, y.to(DEV))
        loss.backward()
        opt.step()
        logits_ref, loss_ref, grads_ref = vo.loss_and_grads(ref_sd, x, y, **fwd_kwargs(cfg))
        # step 1 sees identical parameters -> the 1e-3 bar.  Later steps follow Adam updates whose first move is
        # +-lr*sign(g) per parameter, so near-zero bf16 gradients legitimately flip and trajectories drift apart.
        tol = LOGIT_TOL if step == 1 else 3e-2
        assert float((pred.detach().cpu() - logits_ref).abs().max()) <= tol, f'step {step}'
        assert abs(float(loss) - float(loss_ref)) <= tol
        for k, g in grads_ref.items():
            vo.adam_step(ref_sd[k], g, m[k], v2[k], step)
    # unused 2-D stem / head parameters never receive a gradient
    assert model.pos_embed.grad is None and model.head.weight.grad is None and model.patch_embed.proj.weight.grad is None
    assert model.blocks[0].attn.qkv.weight.grad is not None
    # state_dict round trip keeps the reference key set
    assert set(model.state_dict().keys()) == set(sd.keys())


def test_fused_train_step_matches_oracle_and_graph_replay():
    cfg = dict(backbone='deit_tiny_patch16_224', embed_layer='VoxelEmbed', voxel_size=12, cell=4, patch=3, n_classes=10,
               pos_embedding='default', head='default', batch=6)
    sd = vo.init_state_dict(seed=4, exercise_all=True, **{k: cfg[k] for k in MODEL_KEYS})
    x, y = vo.synthetic_batch(6, 12, 10, seed=6)
    eng = make_engine(cfg, sd)
    eng2 = make_engine(cfg, sd)
    graph, sx, sy, gloss = eng2.capture_train_step(6)
    sx.copy_(x.to(DEV)); sy.copy_(y.to(DEV))
    ref_sd = {k: v.clone() for k, v in sd.items() if k in eng.shapes}
    m = {k: torch.zeros_like(v) for k, v in ref_sd.items()}
    v2 = {k: torch.zeros_like(v) for k, v in ref_sd.items()}
    full = dict(sd)
    for step in range(1, 5):
        loss = float(eng.train_step(x.to(DEV), y.to(DEV)))
        graph.replay()
        gl = float(gloss)
        full.update(ref_sd)
        _, loss_ref, grads_ref = vo.loss_and_grads(full, x, y, **fwd_kwargs(cfg))
        assert abs(loss - float(loss_ref)) <= (LOGIT_TOL if step == 1 else 3e-2), f'step {step}: {loss} vs {float(loss_ref)}'
        assert abs(gl - loss) <= 2e-3, f'graph replay loss {gl} vs eager {loss}'   # fp32-atomic wgrad order differs
        for k, g in grads_ref.items():
            vo.adam_step(ref_sd[k], g, m[k], v2[k], step)
    # after 4 Adam steps the parameters moved by ~4*lr each; compare the update direction with the oracle's
    got = eng.state_dict()
    num = den = 0.0
    for k in ref_sd:
        d_got = (got[k].cpu() - sd[k]).flatten().double()
        d_ref = (ref_sd[k] - sd[k]).flatten().double()
        num += float((d_got * d_ref).sum()); den += float(d_got.norm() * d_ref.norm())
    assert num / den > 0.9, f'update cosine {num / den:.4f}'


def test_cfg2_full_size_properties():
    """BASELINE cfg-2 (deit_small + VoxelEmbed 32^3, batch 64): size-independent checks."""
    kw = dict(backbone='deit_small_patch16_224', embed_layer='VoxelEmbed', voxel_size=32, cell=6, patch=5, n_classes=40)
    sd = vo.init_state_dict(seed=9, **kw)                       # the reference's own init (voxel_pos_embed zeros ...)
    x, y = vo.synthetic_batch(64, 32, 40, seed=9)
    eng = s3d.VoxelEngine(device=DEV, **kw)
    eng.load_state_dict(sd)
    xd, yd = x.to(DEV), y.to(DEV)
    logits = eng.forward(xd).clone()
    assert torch.isfinite(logits).all()
    # (1) run-to-run determinism of the forward (no atomics on that path): bitwise
    assert torch.equal(logits, eng.forward(xd))
    # (2) batch independence: each sample's logits do not depend on its batch-mates
    half = eng.forward(xd[:32].contiguous()).clone()
    assert float((half - logits[:32]).abs().max()) <= 1e-5
    # (3) batch-permutation equivariance
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(0)).to(DEV)
    assert float((eng.forward(xd[perm].contiguous()) - logits[perm]).abs().max()) <= 1e-5
    # (4) parity with the CPU oracle on a slice (full-size weights, 8 samples)
    with torch.no_grad():
        ref = vo.forward(sd, x[:8], backbone=kw['backbone'], embed_layer='VoxelEmbed', cell=6, patch=5)
    assert float((logits[:8].cpu() - ref).abs().max()) <= LOGIT_TOL
    assert torch.equal(logits[:8].cpu().argmax(1), ref.argmax(1))
    # (5) gradient linearity: batch-mean gradient == mean of the two half-batch gradients
    eng.forward(xd); eng.cross_entropy(64, yd); eng.zero_grad(); eng.backward(64)
    g_full = eng.arena.g.clone()
    g_half = torch.zeros_like(g_full)
    for sl in (slice(0, 32), slice(32, 64)):
        eng.forward(xd[sl].contiguous()); eng.cross_entropy(32, yd[sl].contiguous()); eng.zero_grad(); eng.backward(32)
        g_half += 0.5 * eng.arena.g
    rel = float((g_full - g_half).norm() / g_full.norm())
    assert rel < 2e-2, f'gradient linearity rel err {rel:.3e}'
    # (6) a few fused steps reduce the loss on a fixed batch
    eng.zero_grad()
    l0 = float(eng.train_step(xd, yd))
    for _ in range(10):
        l1 = float(eng.train_step(xd, yd))
    assert l1 < l0, f'loss did not decrease: {l0} -> {l1}'


@pytest.mark.parametrize('init', ['reference', 'exercise_all'])
def test_cfg2_full_size_parity(init):
    """BASELINE configs[1] at its FULL size -- deit_small + VoxelEmbed 32^3, batch 64 (1664 token rows) -- in the DEFAULT mode, i.e.
    exactly the launches bench.py times (fused block forward, gemm_pair_dmat_kernel dgrad || wgrad pairs with split-K atomics,
    attn_bwd_small_kernel, class-rows-only last block, fused loss end): all 64 x 40 logits, every clear-cut class decision, the
    loss and EVERY gradient tensor against the CPU oracle's forward + autograd of the same batch (train_cls_voxel.py:277-287).
    'reference' = the reference's own initialisation (zero positional embedding, unit LayerNorm gains: what the bench runs),
    'exercise_all' = every zero / one-initialised tensor perturbed so that bias / affine / positional paths carry signal."""
    kw = dict(backbone='deit_small_patch16_224', embed_layer='VoxelEmbed', voxel_size=32, cell=6, patch=5, n_classes=40)
    sd = vo.init_state_dict(seed=9, exercise_all=(init == 'exercise_all'), **kw)
    x, y = vo.synthetic_batch(64, 32, 40, seed=9)
    assert not L.lib().s3d_get_deterministic()
    eng = s3d.VoxelEngine(device=DEV, **kw)
    eng.load_state_dict(sd)
    xd, yd = x.to(DEV), y.to(DEV)
    ref_logits, ref_loss, ref_grads = vo.loss_and_grads(sd, x, y, backbone=kw['backbone'], embed_layer='VoxelEmbed', cell=6, patch=5)
    # (1) the inference entry points
    logits = eng.forward(xd).cpu()
    err = float((logits - ref_logits).abs().max())
    assert err <= LOGIT_TOL, f'logits max abs err {err:.3e}'
    top2 = ref_logits.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2 * LOGIT_TOL
    assert int(clear.sum()) >= 56, 'too few clear-cut decisions for the argmax check to mean anything'
    assert torch.equal(logits.argmax(1)[clear], ref_logits.argmax(1)[clear])
    # (2) the training entry point bench.py times: fused loss end + backward, default (atomic split-K) dispatch
    eng.zero_grad()
    loss = float(eng.forward_loss(xd, yd))
    assert abs(loss - float(ref_loss)) <= LOGIT_TOL, (loss, float(ref_loss))
    assert float((eng.workspace(64).logits.cpu() - ref_logits).abs().max()) <= LOGIT_TOL
    eng.backward(64)
    grads = {k: eng.arena.grad(k) for k in eng.shapes}
    assert set(grads) == set(ref_grads)
    stats = check_grads_against_oracle(grads, ref_grads, rtol=3e-3)
    worst = max(stats.items(), key=lambda kv: kv[1][0])
    print(f'cfg-2 B=64 [{init}]: logits err {err:.2e}, loss err {abs(loss - float(ref_loss)):.2e}, worst grad rms err / rms '
          f'{worst[1][0]:.4f} ({worst[0]}), worst block {max(v[1] for v in stats.values()):.4f}, worst entry {max(v[2] for v in stats.values()):.3f}')
    # (3) the unfused chain (forward + cross_entropy + backward from d(logits)) lands on the same gradients
    eng.zero_grad()
    eng.forward(xd); eng.cross_entropy(64, yd); eng.backward(64)
    check_grads_against_oracle({k: eng.arena.grad(k) for k in eng.shapes}, ref_grads, rtol=3e-3)


@pytest.mark.parametrize('weighted', [False, True])
def test_fused_loss_end_equals_the_six_kernel_chain(weighted):
    """s3d_head_loss_fused (final norm -> head -> cross entropy -> d(logits) -> d(feat) -> final-norm backward + single-writer batch
    reductions) against s3d_layernorm_fwd + s3d_head_fwd + s3d_cross_entropy + s3d_head_bwd + s3d_layernorm_bwd on the same model."""
    cfg = dict(backbone='deit_tiny_patch16_224', embed_layer='VoxelEmbed', voxel_size=12, cell=4, patch=3, n_classes=10,
               pos_embedding='default', head='default', batch=7)
    sd = vo.init_state_dict(seed=21, exercise_all=True, **{k: cfg[k] for k in MODEL_KEYS})
    x, y = vo.synthetic_batch(7, 12, 10, seed=22)
    w = (torch.rand(10, generator=torch.Generator().manual_seed(3)) + 0.5).to(DEV) if weighted else None
    ref, eng = make_engine(cfg, sd), make_engine(cfg, sd)
    ref.forward(x.to(DEV)); l_ref = float(ref.cross_entropy(7, y.to(DEV), w)); ref.zero_grad(); ref.backward(7)
    eng.forward_features(x.to(DEV)); l_got = float(eng.head_loss(7, y.to(DEV), w)); eng.zero_grad()
    # zero_grad wiped the head / norm gradients head_loss had accumulated: run it again on the clean arena, then the blocks
    eng.head_loss(7, y.to(DEV), w); eng.backward(7)
    wr, wg = ref.workspace(7), eng.workspace(7)
    assert abs(l_ref - l_got) <= 1e-6 * max(1.0, abs(l_ref))
    assert float((wr.logits - wg.logits).abs().max()) <= 1e-5 and float((wr.dlogits - wg.dlogits).abs().max()) <= 1e-7
    assert float((wr.feat - wg.feat).abs().max()) <= 1e-5
    for k in ('voxel_head.weight', 'voxel_head.bias', 'norm.weight', 'norm.bias'):
        a, b = ref.arena.grad(k), eng.arena.grad(k)
        assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max())), k
    rel = float((ref.arena.g - eng.arena.g).norm() / ref.arena.g.norm())
    assert rel < 2e-2, rel                                       # the blocks below see the same d(x_final) up to bf16 / atomics noise


def test_fused_layernorm_epilogue_equals_the_standalone_kernels():
    """ln_fuse=True: norm2 / the next block's norm1 are computed by the last-arriving tile of every row band inside the attn.proj /
    mlp.fc2 GEMM launches (in-launch hand-off through write-through stores + an agent-scope ticket).  Same row arithmetic as the
    LayerNorm kernel -> the logits must agree to rounding, at the full cfg-2 size (52 bands x 12 tiles per launch) and repeatedly
    (a stale hand-off would show as a sporadic mismatch)."""
    kw = dict(backbone='deit_small_patch16_224', embed_layer='VoxelEmbed', voxel_size=32, cell=6, patch=5, n_classes=40)
    sd = vo.init_state_dict(seed=9, exercise_all=True, **kw)
    x, y = vo.synthetic_batch(64, 32, 40, seed=9)
    xd = x.to(DEV)
    plain = s3d.VoxelEngine(device=DEV, ln_fuse=False, **kw); plain.load_state_dict(sd)
    fused = s3d.VoxelEngine(device=DEV, ln_fuse=True, **kw); fused.load_state_dict(sd)
    want = plain.forward(xd).clone()
    for _ in range(20):
        got = fused.forward(xd)
        assert float((got - want).abs().max()) <= 5e-5            # FMA-contraction-level differences only
    wsf, wsp = fused.workspace(64).blocks, plain.workspace(64).blocks
    assert float((wsf.stats[:-1] - wsp.stats[:-1]).abs().max()) <= 1e-5           # mean / rstd of every LayerNorm (the last block's norm2
                                                                                  # runs on the class rows only: compact statistics)
    assert int(wsf.ln_tickets.abs().sum()) == 0                                   # tickets are left zero for the next launch
    fused.cross_entropy(64, y.to(DEV)); fused.zero_grad(); fused.backward(64)
    plain.cross_entropy(64, y.to(DEV)); plain.zero_grad(); plain.backward(64)
    rel = float((fused.arena.g - plain.arena.g).norm() / plain.arena.g.norm())
    assert rel < 2e-2, rel


@pytest.mark.parametrize('mode', ['event_graph', 'segment_graphs', 'event_graph_bf16_wire', 'auto', 'auto_fallback'])
def test_dp_trainer_segmented_graphs_and_rccl_path(mode):
    """DataParallelTrainer on one GPU with a (forced) RCCL all-reduce of each of 3 gradient buckets, vs the plain eager fused step.
    event_graph (opt-in: an event-record node costs more than the graph boundary it replaces on this runtime): ONE natively assembled graph with an event behind every backward segment, collectives launched from a
    side stream on those events (s3d_graph_marker / s3d_graph_events_at_markers); segment_graphs: one graph per segment, collectives in between."""
    import os
    import torch.distributed as dist
    from simple3d_former_amd.parallel import DataParallelTrainer
    cfg = dict(backbone='deit_tiny_patch16_224', embed_layer='VoxelEmbed', voxel_size=12, cell=4, patch=3, n_classes=10,
               pos_embedding='default', head='default', batch=5)
    sd = vo.init_state_dict(seed=7, exercise_all=True, **{k: cfg[k] for k in MODEL_KEYS})
    x, y = vo.synthetic_batch(5, 12, 10, seed=8)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29517')
    dist.init_process_group('nccl', rank=0, world_size=1)
    tr = None
    try:
        ref = make_engine(cfg, sd)
        eng = make_engine(cfg, sd)
        if mode == 'auto':      # the default: a child-process preflight on the same rank / device decides; with RCCL it passes -> captured collectives
            tr = DataParallelTrainer(eng, n_buckets=3, use_graphs=True, force_collectives=True)
            assert tr.preflight is not None and tr.preflight[0], tr.preflight
            assert tr.collectives_mode() == 'captured in the step graph'
        elif mode == 'auto_fallback':       # the preflight child fails (forced): the trainer must fall back to segment graphs, not hang or raise
            os.environ['S3D_PREFLIGHT_FORCE_FAIL'] = '1'
            try:
                tr = DataParallelTrainer(eng, n_buckets=3, use_graphs=True, force_collectives=True)
            finally:
                del os.environ['S3D_PREFLIGHT_FORCE_FAIL']
            assert tr.preflight is not None and not tr.preflight[0] and 'exit code 5' in tr.preflight[1], tr.preflight
            assert tr.collectives_mode() == 'host-launched between graph segments'
        else:
            tr = DataParallelTrainer(eng, n_buckets=3, use_graphs=True, force_collectives=True, event_graph=mode != 'segment_graphs',
                                     graph_collectives=False, wire='bf16' if mode.endswith('bf16_wire') else 'fp32')
            assert tr.collectives_mode() == ('host-launched between graph segments' if mode == 'segment_graphs' else 'host-launched on graph events')
        assert len(tr.slices) == 3 and tr.segments == [(11, 6), (5, 3), (2, 0)]
        for step in range(4):
            l_ref = float(ref.train_step(x.to(DEV), y.to(DEV)))
            l_dp = float(tr.step(x.to(DEV), y.to(DEV)))
            assert abs(l_ref - l_dp) <= 2e-3, f'step {step}: {l_ref} vs {l_dp}'
        d = (eng.arena.p - ref.arena.p).abs().max()
        assert ('fwd_bwd' in tr._cap) == (mode.startswith('event_graph')) and ('whole' in tr._cap) == (mode == 'auto')
        assert (len(tr._cap['graphs']) == 3) == (mode in ('segment_graphs', 'auto_fallback'))
        assert float(d) <= 8.5e-3          # bound 2*steps*lr: Adam moves +-lr per step and fp32-atomic ordering may flip near-zero grads
    finally:
        release_graphs(tr)                  # the HIP graphs (with captured collectives in 'auto') go before the communicator
        del tr
        teardown_process_group()


@pytest.mark.parametrize('mode', ['eager', 'phase_graphs', 'whole_graph', 'rccl_auto', 'group_embed'])
def test_sharded_trainer_on_one_gpu_equals_the_plain_step(mode):
    """ShardedDataParallelTrainer (reduce-scatter -> Adam on the local shard -> all-gather overlapped with the next forward in block ranges) at
    one rank, where a shard is the whole bucket and the collectives move nothing: in DETERMINISTIC mode the parameters after four steps are
    those of the engine's own fused step (same kernels on the same data; only LayerNorm-gradient partials are reduced per segment) -- eagerly,
    as one graph per phase with the side-stream work launched from the host, as ONE captured graph, with forced 1-rank RCCL reduce-scatter /
    all-gather behind the child-process preflight (rccl_auto), and for group_embed (two passes over the shared blocks)."""
    import os
    import torch.distributed as dist
    from simple3d_former_amd import _lib as L
    from simple3d_former_amd.parallel import ShardedDataParallelTrainer
    group = mode == 'group_embed'
    cfg = dict(backbone='deit_tiny_patch16_224', embed_layer='VoxelEmbed_no_average' if group else 'VoxelEmbed', voxel_size=12, cell=4, patch=3,
               n_classes=10, pos_embedding='group_embed' if group else 'default', head='default', batch=5)
    sd = vo.init_state_dict(seed=7, exercise_all=True, **{k: cfg[k] for k in MODEL_KEYS})
    x, y = vo.synthetic_batch(5, 12, 10, seed=8)
    x, y = x.to(DEV), y.to(DEV)
    lib = L.lib()
    lib.s3d_set_deterministic(1)
    rccl = mode == 'rccl_auto'
    if rccl:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29519')
        dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        ref, eng = make_engine(cfg, sd), make_engine(cfg, sd)
        if rccl:
            tr = ShardedDataParallelTrainer(eng, bucket_blocks=[5, 4, 2, 1], force_collectives=True)
            assert tr.preflight is not None and tr.preflight[0], tr.preflight
            assert tr.collectives_mode() == 'reduce-scatter + all-gather, captured in the step graph'
        else:
            tr = ShardedDataParallelTrainer(eng, bucket_blocks=[5, 4, 2, 1], use_graphs=mode != 'eager',
                                            graph_collectives=mode in ('whole_graph', 'group_embed'))
        assert tr.segments == [(11, 7), (6, 3), (2, 1), (0, 0)] and tr.fwd_ranges == [(0, 0), (1, 2), (3, 6), (7, 11)]
        assert tr.slices[0][1] == eng.arena.numel and tr.slices[-1][0] == 0 and tr.shards == tr.slices
        for step in range(4):
            l_ref = float(ref.train_step(x, y))
            l_dp = float(tr.step(x, y))
            assert abs(l_ref - l_dp) <= 1e-5 * max(1.0, abs(l_ref)), f'step {step}: {l_ref} vs {l_dp}'
        assert tr.pending
        tr.sync_parameters()
        torch.cuda.synchronize()
        if mode != 'eager':
            assert ('whole' in tr._cap) == (mode in ('whole_graph', 'rccl_auto', 'group_embed'))
        d = float((eng.arena.p - ref.arena.p).abs().max())
        assert d <= 2e-6, d                               # (LayerNorm partials are summed per segment: the last bits of those gradients may differ)
        assert torch.equal(eng.arena.hi, ref.arena.hi) or d > 0
        hi, lo = eng.arena.hi.clone(), eng.arena.lo.clone()
        eng.refresh_weight_planes()                       # the planes the gather phase wrote ARE the split of the parameters
        assert torch.equal(hi, eng.arena.hi) and torch.equal(lo, eng.arena.lo)
        assert not eng.arena.g.any()
    finally:
        lib.s3d_set_deterministic(0)
        if rccl:
            release_graphs(locals().get('tr'))
            teardown_process_group()


def test_group_embed_training_mode_dropout_matches_oracle():
    """model.train() semantics of group_embed: dropout(0.1) at the four sites of nn.TransformerEncoderLayer, with the
    counter-based mask shared by the HIP kernels and the oracle (torch's RNG stream itself cannot be reproduced)."""
    cfg = dict(backbone='deit_tiny_patch16_224', embed_layer='VoxelEmbed_no_average', voxel_size=12, cell=4, patch=3, n_classes=10,
               pos_embedding='group_embed', head='default', batch=3)
    sd = vo.init_state_dict(seed=5, exercise_all=True, portable=True, **{k: cfg[k] for k in MODEL_KEYS})
    x, y = vo.synthetic_batch(3, 12, 10, seed=6, portable=True)
    eng = make_engine(cfg, sd)
    eng.set_dropout(0.1, seed=1234)
    logits = eng.forward(x.to(DEV)).cpu()
    kw = dict(fwd_kwargs(cfg), training=True, dropout_p=0.1, hash_seed=1234)
    logits_ref, loss_ref, grads_ref = vo.loss_and_grads(sd, x, y, **kw)
    with torch.no_grad():
        logits_eval = vo.forward(sd, x, **fwd_kwargs(cfg))
    assert float((logits_ref - logits_eval).abs().max()) > 1e-2           # the masks really change the output
    assert float((logits - logits_ref).abs().max()) <= LOGIT_TOL
    loss = float(eng.cross_entropy(3, y.to(DEV)))
    assert abs(loss - float(loss_ref)) <= LOGIT_TOL
    eng.zero_grad(); eng.backward(3)
    for k, g in grads_ref.items():
        got = eng.arena.grad(k).cpu().double()
        ref = g.double()
        rms = float(ref.pow(2).mean().sqrt())
        err = float((got - ref).pow(2).mean().sqrt())
        assert err <= 3e-2 * rms + 1e-7, f'{k}: grad rms err {err:.3e} vs rms {rms:.3e}'
    # a different seed gives different masks; seed advance per fused step
    eng.set_dropout(0.1, seed=99)
    assert float((eng.forward(x.to(DEV)).cpu() - logits).abs().max()) > 1e-3
    s0 = int(eng.dropout_seed)
    eng.train_step(x.to(DEV), y.to(DEV))
    assert int(eng.dropout_seed) == s0 + 1


# ---------------------------------------------------------------------------------------------- 2-D branch / LwF
def _lwf_fixture():
    z, cfg = load_case('tiny_v12_lwf_b2')
    sd, x, y = rebuild_inputs(cfg, z)
    img = (vo.portable_uniform((cfg['batch'], 3, 224, 224), 9, 7001) * 2 - 1).float()
    yi = (vo.portable_uniform((cfg['batch'],), 9, 7002) * 1000).long()
    return z, cfg, sd, x, y, img, yi


def test_forward_images_and_lwf_gradients_match_reference_golden():
    """forward_images (vit_3d_2d_pretrain.py:435-451) and d(CE_voxel + 0.1 * CE_image)/d(params) (train_cls_voxel.py:250-267)."""
    z, cfg, sd, x, y, img, yi = _lwf_fixture()
    eng = make_engine(cfg, sd, image_branch=True)
    B = cfg['batch']
    lv = eng.forward(x.to(DEV)).cpu()
    li = eng.images.forward(img.to(DEV)).cpu()
    err_v = float(np.abs(lv.numpy() - z['logits']).max())
    err_i = float(np.abs(li.numpy() - z['img_logits']).max())
    assert err_v <= LOGIT_TOL and err_i <= LOGIT_TOL, f'logits err voxel {err_v:.3e} image {err_i:.3e}'
    np.testing.assert_array_equal(li.argmax(1).numpy(), z['img_argmax'])
    assert float(z['img_top2_gap'].min()) > 2 * LOGIT_TOL
    loss_v = float(eng.cross_entropy(B, y.to(DEV)))
    loss_i = float(eng.images.cross_entropy(B, yi.to(DEV), grad_scale=cfg['lambda_weight']))
    assert abs(loss_v - float(z['loss_voxel'])) <= LOGIT_TOL and abs(loss_i - float(z['loss_image'])) <= LOGIT_TOL
    assert abs(loss_v + cfg['lambda_weight'] * loss_i - float(z['loss'])) <= LOGIT_TOL
    eng.zero_grad()
    eng.backward(B)
    eng.images.backward(B)                                    # accumulates on top of the voxel gradients
    grads = {k: eng.arena.grad(k) for k in eng.shapes}
    assert set(grads) == set(json.loads(str(z['grad_names'])))
    worst = check_grads_against_golden(z, grads, rtol=3e-3, atol=1e-7)
    print(f'lwf: logits err voxel {err_v:.2e} image {err_i:.2e}, worst sampled grad err / rms {worst:.3f}')


def test_drop_in_module_lwf_step_and_frozen_stem():
    """model(voxel) + model.forward_images(images) under autograd, as the LwF loop uses them; with the reference's freezing of
    the 2-D stem / head (pretrained checkpoints) those parameters get no gradient while the shared blocks still do."""
    z, cfg, sd, x, y, img, yi = _lwf_fixture()
    model = s3d.Feature3D_ViT2D_V2(embed_layer=s3d.VoxelEmbed(voxel_size=12, cell_size=4, patch_size=3, embed_dim=192),
                                   n_classes=10, transformer_backbone='deit_tiny_patch16_224', pretrained=False,
                                   pos_embedding='default')
    model.load_state_dict(sd, strict=True)
    model.to(DEV)
    pred, img_pred = model(x.to(DEV)), model.forward_images(img.to(DEV))
    assert float((img_pred.detach().cpu() - torch.from_numpy(z['img_logits'])).abs().max()) <= LOGIT_TOL
    loss = F.cross_entropy(pred, y.to(DEV)) + cfg['lambda_weight'] * F.cross_entropy(img_pred, yi.to(DEV))
    assert abs(float(loss) - float(z['loss'])) <= LOGIT_TOL
    loss.backward()
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    assert set(grads) == set(json.loads(str(z['grad_names'])))
    check_grads_against_golden(z, grads, rtol=3e-3, atol=1e-7)
    with pytest.raises(AssertionError, match="doesn't match model"):
        model.forward_images(torch.zeros(1, 3, 192, 192, device=DEV))
    # frozen 2-D stem / head
    for p in model.parameters():
        p.grad = None
    for k in ('head.weight', 'head.bias', 'pos_embed', 'patch_embed.proj.weight', 'patch_embed.proj.bias'):
        dict(model.named_parameters())[k].requires_grad = False
    model._engine.images.frozen = True
    F.cross_entropy(model.forward_images(img.to(DEV)), yi.to(DEV)).backward()
    named = dict(model.named_parameters())
    assert all(named[k].grad is None for k in ('head.weight', 'pos_embed', 'patch_embed.proj.weight'))
    assert float(named['blocks.0.attn.qkv.weight'].grad.abs().sum()) > 0 and float(named['cls_token'].grad.abs().sum()) > 0
    g = model._engine.arena
    assert float(g.grad('head.weight').abs().sum()) == 0 and float(g.grad('patch_embed.proj.weight').abs().sum()) == 0


def test_lwf_train_step_matches_oracle_adam_step():
    z, cfg, sd, x, y, img, yi = _lwf_fixture()
    eng = make_engine(cfg, sd, image_branch=True)
    eng.set_optimizer(lr=1e-3)
    loss, lv, li = eng.lwf_train_step(x.to(DEV), y.to(DEV), img.to(DEV), yi.to(DEV), lambda_weight=cfg['lambda_weight'])
    assert abs(float(loss) - float(z['loss'])) <= LOGIT_TOL
    _, _, _, grads = vo.lwf_loss_and_grads(sd, x, y, img, yi, lambda_weight=cfg['lambda_weight'], **fwd_kwargs(cfg))
    after = eng.state_dict()
    moved = 0
    for k, g in grads.items():
        want, m0, v0 = sd[k].clone(), torch.zeros_like(sd[k]), torch.zeros_like(sd[k])
        vo.adam_step(want, g, m0, v0, 1, lr=1e-3)
        # first Adam step = -lr * sign(g) wherever |g| >> eps: compare where the oracle gradient is not tiny
        mask = g.abs() > 1e-6
        if mask.any():
            d = (after[k].cpu() - want.reshape(after[k].shape))[mask.reshape(after[k].shape)].abs().max()
            assert float(d) <= 2.1e-3, f'{k}: {float(d):.3e}'          # a sign flip of a ~0 gradient moves 2*lr
            moved += 1
    assert moved > 100
    assert float(eng.arena.g.abs().sum()) == 0.0                # Adam zeroed the gradients


def test_attention_visualisation_hooks_fire_with_reference_semantics():
    """visualize_attention_map_voxel.py:120-146: forward hooks on model.blocks[i].attn that rebuild softmax(q k^T * scale) from
    the hook input with the module's .qkv / .num_heads / .scale."""
    z, cfg = load_case('tiny_v12_default_b3')
    sd, x, y = rebuild_inputs(cfg, z)
    model = s3d.Feature3D_ViT2D_V2(embed_layer=s3d.VoxelEmbed(voxel_size=12, cell_size=4, patch_size=3, embed_dim=192),
                                   n_classes=10, transformer_backbone='deit_tiny_patch16_224', pretrained=False,
                                   pos_embedding='default')
    model.load_state_dict(sd, strict=True)
    model.to(DEV).eval()
    activation = {}

    def get_attn_softmax(name):
        def hook(m, inp, out):
            with torch.no_grad():
                t = inp[0]
                B, N, C = t.shape
                qkv = m.qkv(t).reshape(B, N, 3, m.num_heads, C // m.num_heads).permute(2, 0, 3, 1, 4)
                activation[name] = ((qkv[0] @ qkv[1].transpose(-2, -1)) * m.scale).softmax(dim=-1)
                activation[name + '/out'] = out
        return hook

    for idx, blk in enumerate(model.blocks.children()):
        blk.attn.register_forward_hook(get_attn_softmax(f'attn{idx}'))
    with torch.no_grad():
        model(x.to(DEV))
    assert len([k for k in activation if not k.endswith('/out')]) == 12
    # oracle: the same maps from the fp32 restatement
    kw = fwd_kwargs(cfg)
    t = vo.voxel_embed(x, sd['voxel_embed.proj.conv3d_1.weight'], sd['voxel_embed.proj.conv3d_1.bias'], cfg['cell'])
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat((sd['cls_token'].expand(x.shape[0], -1, -1), t), dim=1) + sd['voxel_pos_embed']
    H = vo.BACKBONES[kw['backbone']]['num_heads']
    for i in range(12):
        pre = f'blocks.{i}.'
        xn = vo.layer_norm(t, sd[pre + 'norm1.weight'], sd[pre + 'norm1.bias'])
        B, N, C = xn.shape
        qkv = vo.linear(xn, sd[pre + 'attn.qkv.weight'], sd[pre + 'attn.qkv.bias']).reshape(B, N, 3, H, C // H).permute(2, 0, 3, 1, 4)
        ref = ((qkv[0] @ qkv[1].transpose(-2, -1)) * (C // H) ** -0.5).softmax(-1)
        got = activation[f'attn{i}'].cpu()
        assert got.shape == ref.shape == (B, H, N, N)
        assert float((got - ref).abs().max()) < 2e-3, f'block {i}'
        if i == 0:      # hook output = the attention branch's contribution attn(norm1(x))
            want = vo.attention(xn, sd, pre + 'attn.', H)
            assert float((activation['attn0/out'].cpu() - want).abs().max()) < 1e-3
        t = vo.vit_block(t, sd, i, H)
