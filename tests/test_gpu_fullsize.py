"""Size-independent properties at BASELINE.json's full sizes for configs[2..4] (cfg-2 has test_cfg2_full_size_properties in
test_gpu_model.py): the kernels that only run at scale -- the cooperative long-sequence attention, 128x128 split-K wgrads, packed
pass-1 attention at 47 k groups, the 2.1 M-row BatchNorm / neighbourhood kernels -- are exercised at the sizes bench.py times.

Invariants are chosen per model: group_embed attends ACROSS the samples of a batch (vit_3d_2d_pretrain.py:472-480), so batch
independence / sub-batch gradient linearity do not hold there (permutation equivariance and linearity in d(logits) do); the point
models couple samples through train-mode BatchNorm, so per-sample checks run in eval mode (running statistics)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import simple3d_former_amd as s3d
    from simple3d_former_amd.point_engine import PointEngine

from oracle import point_oracle as po
from oracle import voxel_oracle as vo
from tests._util import check_grads_against_oracle

DEV = 'cuda'
LOGIT_TOL = 1e-3


def test_cfg3_encoder_attention_forward_repeats_bit_for_bit_launch_after_launch():
    """The seq-first encoder attention at the cfg-3 geometry (15 x 4 heads, 12544 keys, hd 192, weight dropout with the stored mask), 48 launches:
    every output, log-sum-exp and mask word equals the first launch's.  Round 6 shipped a software-pipelined forward whose prologue scored key
    tile 0 from a buffer that a wave one tile ahead was already refilling with K(2): one launch in ~40 came back with 32 - 96 rows off by 1e-4
    (first workgroups of a launch only, where one wave's Q rows can arrive microseconds late) -- profiles/r06_attn_prologue_race.txt."""
    from simple3d_former_amd import ops
    Bb, H, hd, N = 15, 4, 192, 64 * 196
    D = H * hd
    g = torch.Generator(device=DEV).manual_seed(6)
    hi, lo = ops.split_bf16(torch.randn(Bb * N, 3 * D, generator=g, device=DEV) * 0.5)
    seed = torch.tensor([4321], dtype=torch.int64, device=DEV)
    T = (N + 31) // 32
    mbuf = torch.zeros(Bb * H * T * T * 32, dtype=torch.int32, device=DEV)
    side = torch.cuda.Stream()
    for flag in (1, 0):                                            # what the encoder layer launches (one bf16 plane of P), and the full split
        first = None
        for r in range(24):
            if r % 3 == 1:                                         # unrelated traffic on a second stream: different cache / TLB state at launch
                with torch.cuda.stream(side): torch.randn(1 << 24, device=DEV).sum()
            o_hi, o_lo, lse = ops.attention_fwd(hi, lo, Bb, H, N, D, 1, Bb, split=True, drop=(0.1, seed, 0), drop_mask=mbuf, p_single_plane=flag)
            if first is None: first = (o_hi.clone(), o_lo.clone(), lse.clone(), mbuf.clone())
            else:
                for name, a, b in zip(('out_hi', 'out_lo', 'lse', 'mask'), (o_hi, o_lo, lse, mbuf), first):
                    assert torch.equal(a, b), f'launch {r} (p_single_plane {flag}): {int((a != b).sum())} elements of {name} differ from the first launch'
    torch.cuda.synchronize()


@pytest.mark.parametrize('B', [16, 64])
def test_cfg3_full_geometry_properties(B):
    """BASELINE cfg-3 geometry (deit_base H=3, VoxelEmbed_no_average 128^3, cell 9, patch 14, group_embed, 55 classes).
    B = 16: L = 16 * 196 = 3136 keys per (token, head) in the seq-first encoder layer (cooperative attention kernels), 47 040 pass-1
    sequences of 15 tokens (packed pairs), M = 47 040 rows in every pass-1 GEMM (128x128 tiles, split-K wgrads).
    B = 64: the size bench.py --config cfg3 times -- L = 12 544 keys, 188 160 pass-1 rows (128x256 / 256x128 fat tiles), 12 608
    pass-2 rows (the 'long from 8192 rows' dispatch), ~100 GB of saved activations."""
    kw = dict(backbone='deit_base_patch16_224', embed_layer='VoxelEmbed_no_average', voxel_size=128, cell=9, patch=14, n_classes=55)
    sd = vo.init_state_dict(seed=9, pos_embedding='group_embed', exercise_all=True, **kw)
    x, y = vo.synthetic_batch(B, 128, 55, seed=9)
    eng = s3d.VoxelEngine(device=DEV, pos_embedding='group_embed', **kw)
    eng.load_state_dict(sd)
    xd, yd = x.to(DEV), y.to(DEV)
    logits = eng.forward(xd).clone()
    assert torch.isfinite(logits).all()
    # (1) the forward has no atomics: bitwise run-to-run
    assert torch.equal(logits, eng.forward(xd))
    # (2) permuting the samples permutes the logits (attention over the B*196 groups is permutation-equivariant; only the
    #     summation order inside the flash tiles changes)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).to(DEV)
    lp = eng.forward(xd[perm].contiguous()).clone()
    assert float((lp - logits[perm]).abs().max()) <= 2e-4
    # (3) parity with the CPU oracle (same full-size weights) on a 2-sample batch: the cross-sample attention makes a SLICE of the
    #     batch incomparable, so the engine runs the 2-sample batch too (once: ~2 TFLOP of fp32 CPU work per sample)
    if B == 16:
        l2 = eng.forward(xd[:2].contiguous()).clone().cpu()
        with torch.no_grad():
            ref = vo.forward(sd, x[:2], backbone=kw['backbone'], embed_layer=kw['embed_layer'], cell=9, patch=14, pos_embedding='group_embed')
        assert float((l2 - ref).abs().max()) <= LOGIT_TOL, float((l2 - ref).abs().max())
        assert torch.equal(l2.argmax(1), ref.argmax(1))
    # (4) the backward is linear in d(logits): gradients of 2 * dlogits == 2 * gradients (within the split-K atomic / bf16 noise)
    eng.forward(xd); eng.cross_entropy(B, yd)
    ws = eng.workspace(B)
    d1 = ws.dlogits.clone()
    eng.zero_grad(); eng.backward(B, d1)
    g1 = eng.arena.g.clone()
    assert torch.isfinite(g1).all() and float(g1.norm()) > 0
    eng.zero_grad(); eng.backward(B, 2 * d1)
    rel = float((eng.arena.g - 2 * g1).norm() / (2 * g1).norm())
    assert rel < 2e-2, f'backward not linear in dlogits: rel err {rel:.3e}'
    # every used parameter receives a gradient (shared blocks: both passes accumulate)
    for k in ('group_embed.self_attn.in_proj_weight', 'blocks.0.attn.qkv.weight', 'blocks.11.mlp.fc2.weight', 'voxel_embed.proj.conv3d_1.weight',
              'group_pos_embed', 'voxel_pos_embed'):
        assert float(eng.arena.grad(k).abs().max()) > 0, k
    # (5) training-mode dropout(0.1) + a few fused steps reduce the loss on a fixed batch
    eng.set_dropout(0.1, seed=3)
    eng.zero_grad()
    l0 = float(eng.train_step(xd, yd))
    for _ in range(6):
        l1 = float(eng.train_step(xd, yd))
    assert l1 < l0, f'loss did not decrease: {l0} -> {l1}'


def _point_fullsize(task, n_points, d_points, n_classes, B, n_slice):
    backbone = 'deit_tiny_patch16_224'
    sd = po.init_state_dict(backbone=backbone, n_classes=n_classes, d_points=d_points, seed=9)
    x, y, starts = po.synthetic_points(B, n_points, d_points, n_classes, task, seed=9)
    eng = PointEngine(backbone=backbone, n_points=n_points, d_points=d_points, n_classes=n_classes, task=task, device=DEV)
    eng.load_state_dict(sd)
    xd, yd, sts = x.to(DEV), y.to(DEV), tuple(s.to(DEV) for s in starts)
    # ---- eval mode (running statistics): samples are independent
    le = eng.forward(xd, sts, training=False).clone()
    assert torch.isfinite(le).all()
    assert torch.equal(le, eng.forward(xd, sts, training=False)), 'eval-mode forward is not bitwise reproducible'
    ws = eng.workspace(B)
    fps_idx = [t.fps_idx.clone().cpu() for t in ws.td]
    knn_idx = [t.idx.clone().cpu() for t in ws.td]
    half = eng.forward(xd[:B // 2].contiguous(), tuple(s[:B // 2].contiguous() for s in sts), training=False).clone()
    assert float((half - le[:B // 2]).abs().max()) <= 1e-5, 'batch independence (eval mode)'
    # parity with the CPU oracle on a slice: logits + bit-exact FPS / kNN indices
    xs, ss = x[:n_slice], tuple(s[:n_slice] for s in starts)
    with torch.no_grad():
        ref = po.forward(sd, xs, task=task, backbone=backbone, starts=ss, training=False)
    got = le[:n_slice].cpu()
    assert float((got - ref).abs().max()) <= LOGIT_TOL, float((got - ref).abs().max())
    if task == 'cls':
        assert torch.equal(got.argmax(-1), ref.argmax(-1))
    else:                                               # per-point argmax: allow ties inside the tolerance band only
        top2 = ref.topk(2, dim=-1).values
        clear = (top2[..., 0] - top2[..., 1]) > 2 * LOGIT_TOL
        assert torch.equal(got.argmax(-1)[clear], ref.argmax(-1)[clear])
    xyz = xs[..., :3]
    f0 = po.farthest_point_sample(xyz, eng.S[0], ss[0])
    assert torch.equal(fps_idx[0][:n_slice].long(), f0), 'FPS indices (level 0) differ from the reference algorithm'
    nx0 = po.index_points(xyz, f0)
    assert torch.equal(knn_idx[0][:n_slice].long(), po.knn_indices(nx0, xyz, 16)), 'kNN indices (level 0)'
    f1 = po.farthest_point_sample(nx0, eng.S[1], ss[1])
    assert torch.equal(fps_idx[1][:n_slice].long(), f1), 'FPS indices (level 1)'
    # ---- train mode: batch statistics couple the samples; check reproducibility within the fp64-atomic noise, linearity of the
    # backward in d(logits), and that SGD steps reduce the loss on a fixed batch
    lt = eng.forward(xd, sts).clone()
    lt2 = eng.forward(xd, sts).clone()
    assert float((lt - lt2).abs().max()) <= 1e-4
    eng.cross_entropy(B, yd)
    d1 = ws.dlogits.clone()
    eng.zero_grad(); eng.backward(B)
    g1 = eng.arena.g.clone()
    assert torch.isfinite(g1).all() and float(g1.norm()) > 0
    eng.forward(xd, sts)
    ws.dlogits.copy_(2 * d1)
    eng.zero_grad(); eng.backward(B)
    rel = float((eng.arena.g - 2 * g1).norm() / (2 * g1).norm())
    assert rel < 2e-2, f'backward not linear in dlogits: rel err {rel:.3e}'
    eng.zero_grad()
    l0 = float(eng.train_step(xd, yd, sts))
    for _ in range(8):
        l1 = float(eng.train_step(xd, yd, sts))
    assert l1 < l0, f'loss did not decrease: {l0} -> {l1}'


def test_cfg4_full_size_properties():
    """BASELINE cfg-4: PointTransformerCls, 1024 points x 6 channels, 40 classes, batch 128 (2.1 M grouped rows in TransitionDown 0)."""
    _point_fullsize('cls', 1024, 6, 40, 128, n_slice=4)


def test_cfg5_full_size_properties():
    """BASELINE cfg-5: PointTransformerSeg, 2048 points x 22 channels, 50 parts, batch 32 (513-token sequences, 65 536 head rows)."""
    _point_fullsize('seg', 2048, 22, 50, 32, n_slice=2)


def _point_fullsize_parity(task, n_points, d_points, n_classes, B, rtol, loose=None):
    """One TRAINING-mode step at the benched size in the DEFAULT dispatch (split-K atomic wgrads, column sums from the GEMM epilogues,
    2.1 M-row BatchNorm kernels, cooperative long-sequence attention): logits, loss, BatchNorm batch statistics and EVERY gradient
    tensor against the CPU oracle's forward + autograd of the same batch (train_cls.py:119-123 / train_partseg.py:143-150)."""
    backbone = 'deit_tiny_patch16_224'
    sd = po.init_state_dict(backbone=backbone, n_classes=n_classes, d_points=d_points, seed=9)
    x, y, starts = po.synthetic_points(B, n_points, d_points, n_classes, task, seed=9)
    eng = PointEngine(backbone=backbone, n_points=n_points, d_points=d_points, n_classes=n_classes, task=task, device=DEV)
    eng.load_state_dict(sd)
    xd, yd, sts = x.to(DEV), y.to(DEV), tuple(s.to(DEV) for s in starts)
    ref_logits, ref_loss, ref_grads, ref_stats = po.loss_and_grads(sd, x, y, task=task, backbone=backbone, starts=starts)
    logits = eng.forward(xd, sts).cpu().reshape(ref_logits.shape)
    err = float((logits - ref_logits).abs().max())
    assert err <= LOGIT_TOL, f'logits max abs err {err:.3e}'
    top2 = ref_logits.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 2 * LOGIT_TOL
    assert float(clear.float().mean()) > 0.8
    assert torch.equal(logits.argmax(-1)[clear], ref_logits.argmax(-1)[clear])
    loss = float(eng.cross_entropy(B, yd))
    assert abs(loss - float(ref_loss)) <= LOGIT_TOL, (loss, float(ref_loss))
    eng.zero_grad()
    eng.backward(B)
    grads = {k: eng.arena.grad(k) for k in eng.shapes}
    # biases that feed a train-mode BatchNorm (and norm.bias, which reaches the loss only through one) have a theoretically ZERO
    # gradient: both sides return rounding noise there (tests/test_gpu_points.py)
    zero_theory = ('mlp_convs.0.bias', 'mlp_convs.1.bias', 'fc1.0.bias', 'fc2.0.bias', 'norm.bias')
    skip = {k for k in ref_grads if k.endswith(zero_theory) and (k.startswith('transition_') or k == 'norm.bias')} | {'fc1.2.bias', 'fc_pos_embed.2.bias'}
    stats = check_grads_against_oracle(grads, {k: v for k, v in ref_grads.items() if k not in skip}, rtol=rtol, atol=3e-7, loose=loose)
    worst = max(stats.items(), key=lambda kv: kv[1][0])
    print(f'{task} B={B}: logits err {err:.2e}, worst grad rms err / rms {worst[1][0]:.4f} ({worst[0]}), worst entry / local scale {max(v[2] for v in stats.values()):.3f}')


def test_cfg4_full_size_parity():
    """Gradient bar 5 % rms (rtol 5e-3), no loosened tensor.  Until round 4 the bar was 10 %: the gradient that leaves a train-mode
    BatchNorm sums to ZERO over the rows of every channel, its bf16 copy (S3dBnArgs::dx) did not, and everything that multiplies it by
    a column with a non-zero MEAN -- the feature columns of the set-abstraction convolution, the relu(.) inputs of fc1.2 /
    fc_pos_embed.2, the all-ones column of a bias -- picked up mean * (random walk of 2 M rounding residues) where the exact product
    cancels (tools/probes/point_grad_noise_probe.py, profiles/r04_point_grad_noise.txt): 6.9 % on fc_pos_embed.0.bias, 4.9 % (27 % in the
    worst 64 x 64 block) on TransitionDown 0's conv-0 weight.  Round 5: the apply kernel carries the rounding residue from element to
    element (points.hip: bn_bwd_apply_vec_kernel), the stored column sums miss zero by half an ulp per workgroup -> 2.4 % / 1.7 % (5.4 %
    worst block) on those two.  What is left on top is cls_token (4.3 - 4.7 %, unchanged): the sum of 128 per-cloud class-row gradients
    that went through twelve plain-bf16 block backwards (models/3DViT/model.py:325: every token row carries 1/257 of the pooled
    gradient) -- rounding noise of the backward itself, alpha = 0.998."""
    _point_fullsize_parity('cls', 1024, 6, 40, 128, rtol=5e-3)


def test_cfg5_full_size_parity():
    _point_fullsize_parity('seg', 2048, 22, 50, 32, rtol=2e-3)       # 2 % rms bar (4 % until the BatchNorm dx residue carry of round 5); worst 0.9 %


@pytest.mark.parametrize('B,dropout', [(4, 0.0), (4, 0.1), (7, 0.1)])
def test_cfg3_reduced_batch_training_step_matches_oracle(B, dropout):
    """BASELINE cfg-3 in its real geometry (deit_base H=3, 128^3 grid, cell 9, patch 14, group_embed) at batch 4: 11 760 pass-1 token
    rows (the 'long' GEMM dispatch from 8192 rows: 128x256 / 256x128 tiles, 128x128 split-K wgrads), 784 keys per (position, head) in
    the seq-first encoder layer (cooperative long-sequence attention kernels), packed 15-token pairs in pass 1 -- i.e. the kernel
    instantiations of the benched batch-64 step -- one full training-mode step against the CPU oracle: logits, loss, every gradient.
    dropout = 0.1: nn.TransformerEncoderLayer's training mode with the counter-based mask the oracle shares.
    Batch 7 (round 5, VERDICT r04 'weak' item 2): 20 580 pass-1 rows = 81 row tiles of 256 -- the qkv and fc1 forward launches and every long
    dgrad (qkv / fc1 / fc2 * gelu' / proj) then take the 256 x 256 eight-wave tiles (gemm_nt_fat_kernel<BF16_BIAS | GELU>, gemm_nn_fat_kernel<F32 |
    DGELU | BF16_BIAS>: their workgroups fill 95 % of the rounds they occupy; checked with the launch-coverage hooks, tools/r5/cov_b7.py) -- a
    MODEL-level comparison with the oracle for the kernels that are a third of the benched batch-64 step, not only the operator-level tests.
    (The residual forward launches -- proj, fc2: N = 768 -- only reach that tile from ~38 k rows = batch 13, beyond what the CPU oracle
    finishes in a test; they stay covered at operator level, test_gemm_fat_forward_tile.)"""
    kw = dict(backbone='deit_base_patch16_224', embed_layer='VoxelEmbed_no_average', voxel_size=128, cell=9, patch=14, n_classes=55)
    sd = vo.init_state_dict(seed=9, pos_embedding='group_embed', exercise_all=True, **kw)
    x, y = vo.synthetic_batch(B, 128, 55, seed=9)
    eng = s3d.VoxelEngine(device=DEV, pos_embedding='group_embed', **kw)
    eng.load_state_dict(sd)
    okw = dict(backbone=kw['backbone'], embed_layer=kw['embed_layer'], cell=9, patch=14, pos_embedding='group_embed')
    if dropout > 0:
        eng.set_dropout(dropout, seed=77)
        okw.update(training=True, dropout_p=dropout, hash_seed=77)
    ref_logits, ref_loss, ref_grads = vo.loss_and_grads(sd, x, y, **okw)
    xd, yd = x.to(DEV), y.to(DEV)
    eng.zero_grad()
    loss = float(eng.forward_loss(xd, yd))
    logits = eng.workspace(B).logits.cpu()
    err = float((logits - ref_logits).abs().max())
    assert err <= LOGIT_TOL, f'logits max abs err {err:.3e}'
    assert abs(loss - float(ref_loss)) <= LOGIT_TOL
    eng.backward(B)
    grads = {k: eng.arena.grad(k) for k in eng.shapes}
    assert set(grads) == set(ref_grads)
    stats = check_grads_against_oracle(grads, ref_grads, rtol=3e-3)
    worst = max(stats.items(), key=lambda kv: kv[1][0])
    print(f'cfg-3 B={B} dropout {dropout}: logits err {err:.2e}, worst grad rms err / rms {worst[1][0]:.4f} ({worst[0]}), worst entry {max(v[2] for v in stats.values()):.3f}')


def test_cfg3_forward_at_batch_13_reaches_the_fat_residual_tile_and_matches_oracle():
    """VERDICT r05 'weak' item 4: `gemm_nt_fat_kernel<RESID>` -- the forward attn.proj / mlp.fc2 launches of the benched cfg-3 step, its dominant
    kernel -- is dispatched from ~38 k token rows only, i.e. from batch 13 (38 220 pass-1 rows).  One FORWARD of the real geometry at batch 13
    against the oracle's forward (the oracle's backward at that size is out of a test's reach; the forward alone is ~9 TFLOP of fp32 CPU work,
    a dozen seconds on sixteen threads): all 13 x 55 logits within 1e-3, class decisions equal, and the launch-coverage hooks confirm that the
    256 x 256 residual tile ran inside this comparison."""
    from simple3d_former_amd import _lib as L
    from tests import _cov
    B = 13
    kw = dict(backbone='deit_base_patch16_224', embed_layer='VoxelEmbed_no_average', voxel_size=128, cell=9, patch=14, n_classes=55)
    sd = vo.init_state_dict(seed=9, pos_embedding='group_embed', exercise_all=True, **kw)
    x, y = vo.synthetic_batch(B, 128, 55, seed=9)
    eng = s3d.VoxelEngine(device=DEV, pos_embedding='group_embed', **kw)
    eng.load_state_dict(sd)
    logits = eng.forward(x.to(DEV)).cpu()
    launched = _cov.collect(L.lib())                     # (conftest enabled the hooks: this test is in _cov.ORACLE_COMPARED)
    fat_resid = 'gemm_nt_fat:' + str(300000000000 + 256 * 100000000 + 256 * 100000 + 100 + 2)       # gemm.hip: KEY of the 256 x 256 NT tile, EPI_RESID = 2
    assert launched.get(fat_resid, 0) >= 24, f'the 256 x 256 residual tile did not run at batch {B}: {sorted(k for k in launched if k.startswith("gemm_nt"))}'
    with torch.no_grad():
        ref = vo.forward(sd, x, backbone=kw['backbone'], embed_layer=kw['embed_layer'], cell=9, patch=14, pos_embedding='group_embed')
    err = float((logits - ref).abs().max())
    assert err <= LOGIT_TOL, f'logits max abs err {err:.3e}'
    top2 = ref.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2 * LOGIT_TOL
    assert torch.equal(logits.argmax(1)[clear], ref.argmax(1)[clear])
    print(f'cfg-3 forward at batch {B}: logits err {err:.2e}; {launched[fat_resid]} launches of gemm_nt_fat_kernel<RESID>; {int(clear.sum())}/{B} clear decisions equal')
