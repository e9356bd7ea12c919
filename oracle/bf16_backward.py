"""TEST INFRASTRUCTURE ONLY -- the voxel oracle's training step with the HIP backward's ROUNDING POINTS restated on the CPU.

Why: the HIP backward runs its GEMMs / attention on plain-bf16 operands, so against the reference's fp32
gradients it can only be held to the bf16 noise floor (3 % rms / 15 % worst entry, tests/_util.py) -- bars that a small systematic
error (a dropped term) would pass.  Here the SAME algorithm as oracle/voxel_oracle.py (timm Block / Attention / Mlp as restated in
oracle/timm_shim/timm/models/vision_transformer.py:41-78, the tokenizers of models/embed_layer_3d_modality.py:150-209, the heads of
models/vit_3d_2d_pretrain.py:39-56,364, F.cross_entropy of train_cls_voxel.py:282-285) is differentiated by hand-written backward
functions that round exactly where the kernels round:

  * every Linear backward reads the incoming gradient as bf16 (the bf16 copy the LayerNorm backward / the previous epilogue wrote),
    the weight's hi plane and the saved activation's hi plane; products accumulate in fp32 (here: fp64); db = column sums of the
    bf16 gradient (gemm.hip wgrad, capi.hip block_bwd);
  * mlp.fc2 dgrad -> * gelu'(bf16 pre-activation) -> bf16 (EPI_DGELU); attn.proj dgrad -> bf16 (d att); fc1 / qkv dgrads stay fp32;
  * attention backward (attention.hip attn_bwd_small_kernel and siblings): q, k, v, dO as bf16, delta = sum(dO * O) and
    P = exp(q k^T * scale - lse) in fp32, dS = bf16(P * (dP - delta) * scale), P as bf16 for dV, dQ / dK / dV stored as bf16;
  * tokenizer wgrad: alpha * bf16(dy)^T A (the folded patch operand is exact in bf16); LayerNorm backward, residual adds, token /
    positional gradients, heads and the loss are fp32.

With round=False every function is the exact derivative: tests/test_oracle_golden.py checks those gradients against autograd on
voxel_oracle (and thereby against the reference goldens it is pinned to), so the hand-written backward is itself pinned.  With
round=True the HIP gradients must agree to ~1e-3 of the gradient rms (accumulation order + the odd rounding-boundary flip) --
tests/test_gpu_model.py::test_backward_matches_the_rounding_faithful_oracle -- 30x tighter than the fp32 comparison allows.
Default positional embedding only (the group_embed encoder layer and the point path keep the fp32 bars)."""
import torch
import torch.nn.functional as F

from . import voxel_oracle as vo


def _r(t, on):
    """fp32 -> bf16 (rne) -> back, the way f2bf / v_cvt_pk_bf16_f32 round; identity when the emulation is off."""
    return t.float().to(torch.bfloat16).to(t.dtype) if on else t


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, on, round_dx):
        ctx.save_for_backward(x, w)
        ctx.on, ctx.round_dx = on, round_dx
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        on = ctx.on
        dyb = _r(dy, on)
        dx = dyb @ _r(w, on)
        if ctx.round_dx:
            dx = _r(dx, on)
        d2, x2 = dyb.reshape(-1, dyb.shape[-1]), _r(x, on).reshape(-1, x.shape[-1])
        return dx, d2.t() @ x2, d2.sum(0), None, None


class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, on):
        ctx.save_for_backward(h)
        ctx.on = on
        return F.gelu(h)

    @staticmethod
    def backward(ctx, dy):
        (h,) = ctx.saved_tensors
        hb = _r(h, ctx.on)                                   # the saved pre-activation is bf16 (S3dBlockActs::hpre)
        cdf = 0.5 * (1.0 + torch.erf(hb * 0.7071067811865476))
        pdf = torch.exp(-0.5 * hb * hb) * 0.3989422804014327
        return _r(dy * (cdf + hb * pdf), ctx.on), None


class _Attention(torch.autograd.Function):
    """q, k, v: [B, H, N, hd] -> softmax(q k^T * scale) v."""
    @staticmethod
    def forward(ctx, q, k, v, scale, on):
        s = (q @ k.transpose(-2, -1)) * scale
        lse = torch.logsumexp(s, dim=-1, keepdim=True)
        o = torch.exp(s - lse) @ v
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.scale, ctx.on = scale, on
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        on, scale = ctx.on, ctx.scale
        dob, qh, kh, vh = _r(do, on), _r(q, on), _r(k, on), _r(v, on)
        delta = (dob * o).sum(-1, keepdim=True)
        p = torch.exp((qh @ kh.transpose(-2, -1)) * scale - lse)
        dp = dob @ vh.transpose(-2, -1)
        ds = _r(p * (dp - delta) * scale, on)
        dq = _r(ds @ kh, on)
        dv = _r(_r(p, on).transpose(-2, -1) @ dob, on)
        dk = _r(ds.transpose(-2, -1) @ qh, on)
        return dq, dk, dv, None, None


class _RoundGrad(torch.autograd.Function):
    """Identity whose gradient is read as bf16 (the tokenizer wgrad's dy operand)."""
    @staticmethod
    def forward(ctx, x, on):
        ctx.on = on
        return x.clone()

    @staticmethod
    def backward(ctx, dy):
        return _r(dy, ctx.on), None


def _block(x, sd, i, H, on):
    p = f'blocks.{i}.'
    B, N, D = x.shape
    hd = D // H
    xn = vo.layer_norm(x, sd[p + 'norm1.weight'], sd[p + 'norm1.bias'])
    qkv = _Linear.apply(xn, sd[p + 'attn.qkv.weight'], sd[p + 'attn.qkv.bias'], on, False)
    qkv = qkv.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    o = _Attention.apply(qkv[0], qkv[1], qkv[2], hd ** -0.5, on)
    att = o.transpose(1, 2).reshape(B, N, D)
    x = x + _Linear.apply(att, sd[p + 'attn.proj.weight'], sd[p + 'attn.proj.bias'], on, True)      # d(att) is stored as bf16
    xn = vo.layer_norm(x, sd[p + 'norm2.weight'], sd[p + 'norm2.bias'])
    h = _Gelu.apply(_Linear.apply(xn, sd[p + 'mlp.fc1.weight'], sd[p + 'mlp.fc1.bias'], on, False), on)
    return x + _Linear.apply(h, sd[p + 'mlp.fc2.weight'], sd[p + 'mlp.fc2.bias'], on, False)        # rounded after gelu', in _Gelu


def forward(sd, x, *, backbone, embed_layer, cell, patch=None, pos_embedding='default', round=True):
    if pos_embedding not in (None, 'default'):
        raise NotImplementedError('the rounding-faithful backward covers the default positional embedding')
    cfg = vo.BACKBONES[backbone]
    depth, H = cfg['depth'], cfg['num_heads']
    ck = 'voxel_embed.proj.conv2d_1' if embed_layer == 'VoxelNaiveProjection' else 'voxel_embed.proj.conv3d_1'
    w, b = sd[ck + '.weight'], sd[ck + '.bias']
    if embed_layer == 'VoxelEmbed':
        t = F.conv3d(x, w, None, stride=cell).mean(dim=4)
    elif embed_layer == 'VoxelNaiveProjection':
        t = F.conv2d(torch.clamp(x.sum(dim=4), min=0, max=1), w, None, stride=cell)
    elif embed_layer == 'VoxelEmbed_no_average':
        t = F.conv3d(x, w, None, stride=cell)
    else:
        raise ValueError(embed_layer)
    t = _RoundGrad.apply(t, round) + b.reshape((1, -1) + (1,) * (t.dim() - 2))
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat((sd['cls_token'].expand(x.shape[0], -1, -1), t), dim=1) + sd['voxel_pos_embed']
    for i in range(depth):
        t = _block(t, sd, i, H, round)
    feat = vo.layer_norm(t, sd['norm.weight'], sd['norm.bias'])[:, 0]
    return vo.voxel_head(feat, sd)


def loss_and_grads(sd, x, target, weight=None, round=True, **kw):
    """(logits, loss, {name: grad}) in fp64 with the HIP backward's rounding points (round=True) or exactly (round=False)."""
    names = vo.used_param_names(sd, kw.get('pos_embedding', 'default'))
    leaf = {k: v.detach().double().clone().requires_grad_(k in names) for k, v in sd.items()}
    logits = forward(leaf, x.double(), round=round, **kw)
    loss = F.cross_entropy(logits, target, weight=None if weight is None else weight.double())
    grads = torch.autograd.grad(loss, [leaf[k] for k in names], allow_unused=True)
    return logits.detach(), loss.detach(), {k: g for k, g in zip(names, grads) if g is not None}
