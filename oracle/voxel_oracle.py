"""TEST INFRASTRUCTURE ONLY -- CPU (PyTorch fp32) restatement of the Simple3D-Former voxel
hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package must never route through it.

Written as pure functions over a flat ``{state_dict key: tensor}`` mapping (the reference's
own key names), so the same parameter dict drives the reference
model (golden generation), this oracle, and the HIP engine.

What each function follows (all paths relative to /root/reference):
  voxel_embed               models/embed_layer_3d_modality.py:150-177  (Conv3d k=s=c, mean over dim 4)
  voxel_embed_no_average    models/embed_layer_3d_modality.py:43-70    (Conv3d only)
  voxel_naive_projection    models/embed_layer_3d_modality.py:182-209  (clamp(sum_z) -> Conv2d)
  am_softmax_head           models/vit_3d_2d_pretrain.py:39-56
  vit_block / attention / mlp   timm==0.3.2 (un-vendored; requirements.txt:6).  In-tree evidence:
                            visualize_attention_map_voxel.py:120-140, models/vip_3d.py:25-41,
                            ctor args models/vit_3d_2d_pretrain.py:279-325 (LayerNorm eps 1e-6,
                            qkv_bias, mlp_ratio 4, deit_base built with num_heads=3)
  group_encoder_layer       torch.nn.TransformerEncoderLayer as constructed at
                            models/vit_3d_2d_pretrain.py:381 (d_model=D, dim_feedforward=D, nhead=4,
                            post-norm, ReLU, batch_first=False, eps 1e-5) and fed at :479
  forward_features          models/vit_3d_2d_pretrain.py:453-496 (default / no_embed / group_embed)
  forward                   models/vit_3d_2d_pretrain.py:523-526
  cross_entropy             train_cls_voxel.py:282-285
  adam_step                 torch.optim.Adam defaults as used at train_cls_voxel.py:195

PINNING: the reference ships no tests or golden vectors.  This oracle is
pinned against outputs of the reference itself, captured in this container by
tests/golden/make_golden.py (reference files imported unmodified on top of oracle/timm_shim);
tests/test_oracle_golden.py replays them.  The timm arithmetic underneath is "parity
unpinned" by the reference (timm is absent); it is cross-checked against independent
implementations in tests/test_oracle_timm.py.
"""
import math

import torch
import torch.nn.functional as F

BACKBONES = {  # models/vit_3d_2d_pretrain.py:279-325 (note: base is built with 3 heads)
    'deit_tiny_patch16_224': dict(embed_dim=192, depth=12, num_heads=3),
    'deit_small_patch16_224': dict(embed_dim=384, depth=12, num_heads=6),
    'deit_base_patch16_224': dict(embed_dim=768, depth=12, num_heads=3),
    'deit_base_distilled_patch16_224': dict(embed_dim=768, depth=12, num_heads=3),
    'vit_base_patch16_224_21k': dict(embed_dim=768, depth=12, num_heads=3),
}
LN_EPS = 1e-6          # partial(nn.LayerNorm, eps=1e-6), vit_3d_2d_pretrain.py:287
GROUP_LN_EPS = 1e-5    # nn.TransformerEncoderLayer default layer_norm_eps
GROUP_HEADS = 4        # vit_3d_2d_pretrain.py:381


def _bf16(t, on):
    """Optional emulation of the HIP path's bf16 MFMA operand rounding (used by tests to
    predict the bf16 error budget; off for the fp32 oracle proper)."""
    return t.to(torch.bfloat16).to(torch.float32) if on else t


def linear(x, w, b=None, bf16=False):
    y = _bf16(x, bf16) @ _bf16(w, bf16).t()
    return y if b is None else y + b


# ----------------------------------------------------------------------------- tokenizers
def voxel_embed(x, w, b, cell):
    """[B,1,V,V,V] -> [B,D,P,P]: non-overlapping Conv3d then mean over the z-patch axis."""
    return F.conv3d(x, w, b, stride=cell).mean(dim=4)


def voxel_embed_folded(x, w, b, cell):
    """Algebraically identical form used by the HIP tokenizer (z-fold first, then one
    patch GEMM with K=c^3): mean_z(conv(x)) == conv applied to the z-folded grid / P."""
    B, _, V, _, _ = x.shape
    c = cell
    P = (V - c) // c + 1
    g = x[:, 0, :P * c, :P * c, :P * c].reshape(B, P, c, P, c, P, c)
    folded = g.sum(dim=5)                                   # [B,P,c,P,c,c]  (sum over pz)
    a = folded.permute(0, 1, 3, 2, 4, 5).reshape(B * P * P, c * c * c)
    out = a @ w.reshape(w.shape[0], -1).t() / P + b         # [B*P*P, D]
    return out.reshape(B, P, P, -1).permute(0, 3, 1, 2)


def voxel_embed_no_average(x, w, b, cell):
    """[B,1,V,V,V] -> [B,D,P,P,P]"""
    return F.conv3d(x, w, b, stride=cell)


def voxel_naive_projection(x, w, b, cell):
    """[B,1,V,V,V] -> clamp(sum over z, 0, 1) [B,1,V,V] -> Conv2d -> [B,D,P,P]"""
    return F.conv2d(torch.clamp(x.sum(dim=4), min=0, max=1), w, b, stride=cell)


# ----------------------------------------------------------------------------- timm block
def layer_norm(x, w, b, eps=LN_EPS):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def attention(x, sd, pre, num_heads, bf16=False):
    B, N, C = x.shape
    hd = C // num_heads
    qkv = linear(x, sd[pre + 'qkv.weight'], sd[pre + 'qkv.bias'], bf16)
    qkv = _bf16(qkv, bf16).reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    p = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
    y = (_bf16(p, bf16) @ v).transpose(1, 2).reshape(B, N, C)
    return linear(_bf16(y, bf16), sd[pre + 'proj.weight'], sd[pre + 'proj.bias'], bf16)


def mlp(x, sd, pre, bf16=False):
    h = F.gelu(linear(x, sd[pre + 'fc1.weight'], sd[pre + 'fc1.bias'], bf16))  # exact erf GELU
    return linear(_bf16(h, bf16), sd[pre + 'fc2.weight'], sd[pre + 'fc2.bias'], bf16)


def vit_block(x, sd, i, num_heads, bf16=False):
    p = f'blocks.{i}.'
    x = x + attention(_bf16(layer_norm(x, sd[p + 'norm1.weight'], sd[p + 'norm1.bias']), bf16),
                      sd, p + 'attn.', num_heads, bf16)
    x = x + mlp(_bf16(layer_norm(x, sd[p + 'norm2.weight'], sd[p + 'norm2.bias']), bf16),
                sd, p + 'mlp.', bf16)
    return x


def run_blocks(x, sd, depth, num_heads, bf16=False):
    for i in range(depth):
        x = vit_block(x, sd, i, num_heads, bf16)
    return layer_norm(x, sd['norm.weight'], sd['norm.bias'])


# ----------------------------------------------------------------------------- group_embed
def hash_keep_mask(shape, seed, site, p, pair_keys=None):
    """Counter-based dropout mask shared with the HIP kernels (common.h: drop_key / drop_mix32 / drop_keep): element i (row-major
    linear index) is kept iff mix32((lo(i) ^ (lo(key) * 0x9E3779B9)) + (mix32(hi(i) ^ hi(key)) ^ lo(key))) >= floor(p * 2^32), with
    key = seed * 0x9E3779B97F4A7C15 + site * 0xD1B54A32D192ED03 + 0x632BE59BD9B4E019 (mod 2^64) and mix32 two rounds of
    xor-shift / multiply.
    pair_keys (default: site == 0, the attention weights [.., query, key]): ONE hash per pair of adjacent keys (2j, 2j + 1) of a query
    row -- the hash of the even key's element index -- and 16 bits of it per decision (low half: even key), compared with
    floor(p * 2^32) >> 16 (common.h: drop_keep_attn / drop_half)."""
    import numpy as np
    if pair_keys is None:
        pair_keys = site == 0
    n = 1
    for d in shape:
        n *= int(d)
    thr = int(p * 4294967296.0)
    key = (seed * 0x9E3779B97F4A7C15 + site * 0xD1B54A32D192ED03 + 0x632BE59BD9B4E019) & _MASK64
    key_hi, key_lo = np.uint32(key >> 32), np.uint32(key & 0xFFFFFFFF)

    def mix32(x):
        x = x ^ (x >> np.uint32(16)); x = x * np.uint32(0x21f0aaad)
        x = x ^ (x >> np.uint32(15)); x = x * np.uint32(0x735a2d97)
        return x ^ (x >> np.uint32(15))

    with np.errstate(over='ignore'):
        idx = np.arange(n, dtype=np.uint64)
        odd = None
        if pair_keys:
            odd = (idx % np.uint64(int(shape[-1]))) & np.uint64(1)    # parity of the key index
            idx = idx - odd                                            # the pair's even key
        lo, hi = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32)
        h = mix32(hi ^ key_hi) ^ key_lo
        z = mix32((lo ^ (key_lo * np.uint32(0x9E3779B9))) + h)     # the key is NOT just an additive offset: no shifted-copy masks
    if pair_keys:
        half = np.where(odd == 1, z >> np.uint32(16), z & np.uint32(0xFFFF)).astype(np.uint64)
        keep = half >= np.uint64(thr >> 16)
        return torch.from_numpy(keep.reshape(tuple(shape)))
    keep = z.astype(np.uint64) >= np.uint64(thr)
    return torch.from_numpy(keep.reshape(tuple(shape)))


def group_encoder_layer(x, sd, dropout_p=0.0, training=False, generator=None, hash_seed=None):
    """Seq-first post-norm encoder layer: x is (L, Nb, D); self-attention runs over axis 0.
    In the reference L = B*P*P and Nb = P+1 (vit_3d_2d_pretrain.py:474-479), i.e. attention
    mixes tokens ACROSS the samples of the batch.  dropout_p>0 with training=True applies the
    four nn.Dropout sites (attention weights, after out_proj, inside FF, after FF)."""
    g = 'group_embed.'
    L, Nb, D = x.shape
    H, hd = GROUP_HEADS, D // GROUP_HEADS

    def drop(t, site):
        """site: 0 attention weights (Nb,H,L,L), 1 after out_proj, 2 after the ReLU, 3 after linear2 (all (L,Nb,*));
        hash_seed selects the counter-based mask shared with the HIP path, otherwise torch's generator is used."""
        if training and dropout_p > 0:
            if hash_seed is not None:
                keep = hash_keep_mask(t.shape, hash_seed, site, dropout_p)
            else:
                keep = torch.rand(t.shape, generator=generator) >= dropout_p
            return t * keep / (1 - dropout_p)
        return t

    qkv = x @ sd[g + 'self_attn.in_proj_weight'].t() + sd[g + 'self_attn.in_proj_bias']
    q, k, v = qkv.split(D, dim=-1)
    # (L, Nb, H, hd) -> (Nb, H, L, hd)
    q = q.reshape(L, Nb, H, hd).permute(1, 2, 0, 3) * hd ** -0.5
    k = k.reshape(L, Nb, H, hd).permute(1, 2, 0, 3)
    v = v.reshape(L, Nb, H, hd).permute(1, 2, 0, 3)
    p = drop((q @ k.transpose(-2, -1)).softmax(dim=-1).contiguous(), 0)
    a = (p @ v).permute(2, 0, 1, 3).reshape(L, Nb, D)
    a = a @ sd[g + 'self_attn.out_proj.weight'].t() + sd[g + 'self_attn.out_proj.bias']
    x = F.layer_norm(x + drop(a, 1), (D,), sd[g + 'norm1.weight'], sd[g + 'norm1.bias'], GROUP_LN_EPS)
    f = drop(F.relu(x @ sd[g + 'linear1.weight'].t() + sd[g + 'linear1.bias']), 2)
    f = f @ sd[g + 'linear2.weight'].t() + sd[g + 'linear2.bias']
    return F.layer_norm(x + drop(f, 3), (D,), sd[g + 'norm2.weight'], sd[g + 'norm2.bias'], GROUP_LN_EPS)


# ----------------------------------------------------------------------------- heads
def am_softmax_head(x, W, s=30.0):
    xn = x / x.norm(p=2, dim=1, keepdim=True).clamp(min=1e-12)
    wn = W / W.norm(p=2, dim=0, keepdim=True).clamp(min=1e-12)
    return (xn @ wn) * s


def voxel_head(feat, sd):
    if 'voxel_head.W' in sd:
        return am_softmax_head(feat, sd['voxel_head.W'])
    return feat @ sd['voxel_head.weight'].t() + sd['voxel_head.bias']


# ----------------------------------------------------------------------------- model
def forward_features(sd, x, *, backbone, embed_layer, cell, patch, pos_embedding='default',
                     bf16=False, training=False, dropout_p=0.1, generator=None, hash_seed=None):
    cfg = BACKBONES[backbone]
    D, depth, H = cfg['embed_dim'], cfg['depth'], cfg['num_heads']
    w, b = sd['voxel_embed.proj.conv3d_1.weight'] if embed_layer != 'VoxelNaiveProjection' else \
        sd['voxel_embed.proj.conv2d_1.weight'], None
    b = sd['voxel_embed.proj.conv3d_1.bias'] if embed_layer != 'VoxelNaiveProjection' else \
        sd['voxel_embed.proj.conv2d_1.bias']
    Bsz = x.shape[0]

    if pos_embedding in (None, 'default', 'no_embed'):
        if embed_layer == 'VoxelEmbed':
            t = voxel_embed(x, w, b, cell)
        elif embed_layer == 'VoxelNaiveProjection':
            t = voxel_naive_projection(x, w, b, cell)
        elif embed_layer == 'VoxelEmbed_no_average':
            t = voxel_embed_no_average(x, w, b, cell)
        else:
            raise ValueError(embed_layer)
        t = t.flatten(2).transpose(1, 2)                                  # [B, n, D]
        t = torch.cat((sd['cls_token'].expand(Bsz, -1, -1), t), dim=1) + sd['voxel_pos_embed']
        return run_blocks(t, sd, depth, H, bf16)[:, 0]

    if pos_embedding == 'group_embed':
        t = voxel_embed_no_average(x, w, b, cell)                          # [B,D,P,P,P]
        P = patch
        t = t.permute(0, 2, 3, 4, 1).reshape(Bsz * P * P, P, D)            # '(b px py) pz c'
        t = torch.cat((sd['group_cls_token'].expand(t.shape[0], -1, -1), t), dim=1)
        t = t + sd['group_pos_embed']
        t = group_encoder_layer(t, sd, dropout_p, training, generator, hash_seed)
        t = run_blocks(t, sd, depth, H, bf16)[:, 0]                        # pass 1
        t = t.reshape(Bsz, P * P, D)
        t = torch.cat((sd['cls_token'].expand(Bsz, -1, -1), t), dim=1) + sd['voxel_pos_embed']
        return run_blocks(t, sd, depth, H, bf16)[:, 0]                     # pass 2 (same weights)

    raise ValueError("Unknown positional embedding scheme!")


def forward(sd, x, **kw):
    return voxel_head(forward_features(sd, x, **kw), sd)


def forward_images(sd, img, *, backbone, bf16=False):
    """Feature3D_ViT2D_V2.forward_images (vit_3d_2d_pretrain.py:435-451): timm PatchEmbed (Conv2d(3, D, 16, stride 16) ->
    flatten(2).transpose(1, 2)), cls concat, + pos_embed, the SAME blocks, norm, 2-D head on the cls row."""
    cfg = BACKBONES[backbone]
    depth, H = cfg['depth'], cfg['num_heads']
    t = F.conv2d(img, sd['patch_embed.proj.weight'], sd['patch_embed.proj.bias'], stride=16).flatten(2).transpose(1, 2)
    t = torch.cat((sd['cls_token'].expand(img.shape[0], -1, -1), t), dim=1) + sd['pos_embed']
    return linear(run_blocks(t, sd, depth, H, bf16)[:, 0], sd['head.weight'], sd['head.bias'])


def lwf_loss_and_grads(sd, x, target, img, img_target, lambda_weight=0.1, **kw):
    """One learning-without-forgetting step's loss (train_cls_voxel.py:250-267): CE(model(voxel)) + lambda * CE(forward_images)."""
    names = sorted(set(used_param_names(sd, kw.get('pos_embedding', 'default'))) |
                   {'patch_embed.proj.weight', 'patch_embed.proj.bias', 'pos_embed', 'head.weight', 'head.bias'})
    leaf = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    lv = forward(leaf, x, **kw)
    li = forward_images(leaf, img, backbone=kw['backbone'])
    loss = F.cross_entropy(lv, target) + lambda_weight * F.cross_entropy(li, img_target)
    grads = torch.autograd.grad(loss, [leaf[k] for k in names], allow_unused=True)
    return lv.detach(), li.detach(), loss.detach(), {k: g for k, g in zip(names, grads) if g is not None}


def cross_entropy(logits, target, weight=None):
    return F.cross_entropy(logits, target, weight=weight)


# ----------------------------------------------------------------------------- training
def used_param_names(sd, pos_embedding='default'):
    """Parameters that receive a gradient from the voxel forward (SURVEY.md section 0 item 4:
    pos_embed, patch_embed.*, head.* are never touched by forward_features)."""
    skip = ('pos_embed', 'patch_embed.', 'head.')
    names = [k for k in sd if not (k == 'pos_embed' or k.startswith(skip[1]) or k.startswith(skip[2]))]
    if pos_embedding != 'group_embed':
        names = [k for k in names if not k.startswith('group_')]
    return names


def loss_and_grads(sd, x, target, weight=None, **kw):
    """Returns (logits, loss, {name: grad}) through torch autograd on the restatement."""
    names = used_param_names(sd, kw.get('pos_embedding', 'default'))
    leaf = {k: (v.detach().clone().requires_grad_(k in names)) for k, v in sd.items()}
    logits = forward(leaf, x, **kw)
    loss = cross_entropy(logits, target, weight)
    grads = torch.autograd.grad(loss, [leaf[k] for k in names], allow_unused=True)
    return logits.detach(), loss.detach(), {k: g for k, g in zip(names, grads) if g is not None}


def adam_step(p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """One torch.optim.Adam update (no weight decay, no amsgrad), in place; `step` is 1-based."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


_MASK64 = (1 << 64) - 1


def portable_uniform(shape, seed, stream):
    """Machine-independent U[0,1) tensor: splitmix64 of (seed, stream, element index) in numpy uint64 arithmetic.
    torch's CPU samplers (normal_/trunc_normal_) differ by an ulp between CPU ISAs (vectorised erfinv), which would
    make golden fixtures unreproducible on the GPU box's host; integer hashing does not."""
    import numpy as np
    n = 1
    for d in shape:
        n *= int(d)
    with np.errstate(over='ignore'):
        key = np.uint64((seed * 0x9E3779B97F4A7C15 + stream * 0xD1B54A32D192ED03 + 0x632BE59BD9B4E019) & _MASK64)
        z = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + key
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return torch.from_numpy(u.reshape(tuple(shape) if len(shape) else ()))


def init_state_dict(*, backbone, embed_layer, voxel_size, cell, patch, n_classes,
                    pos_embedding='default', head='default', seed=9, exercise_all=False, portable=False):
    """Random-init parameter dict with the reference's key names, shapes and init
    distributions (timm _init_weights: Linear trunc_normal std .02 / bias 0, LayerNorm 1/0;
    cls_token/pos_embed trunc_normal .02; voxel_pos_embed / group tokens zeros -- the reference
    never random-initialises them, vit_3d_2d_pretrain.py:370-383; Conv3d / voxel_head Linear /
    group_embed keep torch default init).  Values are NOT bit-identical to the reference's RNG
    stream; golden fixtures carry the reference's own tensors."""
    g = torch.Generator().manual_seed(seed)
    cfg = BACKBONES[backbone]
    D, depth = cfg['embed_dim'], cfg['depth']
    sd = {}

    stream = [0]

    def pu(shape):                       # portable U[0,1), one hash stream per tensor
        stream[0] += 1
        return portable_uniform(shape, seed, stream[0])

    def tn(*shape):
        if portable:                     # same std (.02) as the trunc-normal init, uniform shape
            return ((pu(shape) * 2 - 1) * (0.02 * math.sqrt(3.0))).float()
        t = torch.empty(*shape)
        return torch.nn.init.trunc_normal_(t, std=.02, a=-2., b=2., generator=g)

    def uni(shape, fan_in):
        bound = 1.0 / math.sqrt(fan_in)
        if portable:
            return ((pu(shape) * 2 - 1) * bound).float()
        return (torch.rand(*shape, generator=g) * 2 - 1) * bound

    sd['cls_token'] = tn(1, 1, D)
    sd['pos_embed'] = tn(1, 197, D)
    sd['patch_embed.proj.weight'] = uni((D, 3, 16, 16), 768)
    sd['patch_embed.proj.bias'] = uni((D,), 768)
    for i in range(depth):
        p = f'blocks.{i}.'
        sd[p + 'norm1.weight'] = torch.ones(D); sd[p + 'norm1.bias'] = torch.zeros(D)
        sd[p + 'attn.qkv.weight'] = tn(3 * D, D); sd[p + 'attn.qkv.bias'] = torch.zeros(3 * D)
        sd[p + 'attn.proj.weight'] = tn(D, D); sd[p + 'attn.proj.bias'] = torch.zeros(D)
        sd[p + 'norm2.weight'] = torch.ones(D); sd[p + 'norm2.bias'] = torch.zeros(D)
        sd[p + 'mlp.fc1.weight'] = tn(4 * D, D); sd[p + 'mlp.fc1.bias'] = torch.zeros(4 * D)
        sd[p + 'mlp.fc2.weight'] = tn(D, 4 * D); sd[p + 'mlp.fc2.bias'] = torch.zeros(D)
    sd['norm.weight'] = torch.ones(D); sd['norm.bias'] = torch.zeros(D)
    sd['head.weight'] = tn(1000, D); sd['head.bias'] = torch.zeros(1000)
    c = cell
    if embed_layer == 'VoxelNaiveProjection':
        sd['voxel_embed.proj.conv2d_1.weight'] = uni((D, 1, c, c), c * c)
        sd['voxel_embed.proj.conv2d_1.bias'] = uni((D,), c * c)
    else:
        sd['voxel_embed.proj.conv3d_1.weight'] = uni((D, 1, c, c, c), c ** 3)
        sd['voxel_embed.proj.conv3d_1.bias'] = uni((D,), c ** 3)
    if head == 'AMSoftmax':
        if portable:
            sd['voxel_head.W'] = ((pu((D, n_classes)) * 2 - 1) * math.sqrt(6.0 / (D + n_classes))).float()
        else:
            sd['voxel_head.W'] = torch.randn(D, n_classes, generator=g) * math.sqrt(2.0 / (D + n_classes))
    else:
        sd['voxel_head.weight'] = uni((n_classes, D), D)
        sd['voxel_head.bias'] = uni((n_classes,), D)
    n_tok = {'VoxelEmbed': patch ** 2, 'VoxelNaiveProjection': patch ** 2,
             'VoxelEmbed_no_average': patch ** 3}[embed_layer]
    if pos_embedding == 'group_embed':
        sd['voxel_pos_embed'] = torch.zeros(1, patch ** 2 + 1, D)
        ge = 'group_embed.'
        sd[ge + 'self_attn.in_proj_weight'] = uni((3 * D, D), D) * math.sqrt(3.0) * math.sqrt(2.0) / 2
        sd[ge + 'self_attn.in_proj_bias'] = torch.zeros(3 * D)
        sd[ge + 'self_attn.out_proj.weight'] = uni((D, D), D); sd[ge + 'self_attn.out_proj.bias'] = torch.zeros(D)
        sd[ge + 'linear1.weight'] = uni((D, D), D); sd[ge + 'linear1.bias'] = uni((D,), D)
        sd[ge + 'linear2.weight'] = uni((D, D), D); sd[ge + 'linear2.bias'] = uni((D,), D)
        sd[ge + 'norm1.weight'] = torch.ones(D); sd[ge + 'norm1.bias'] = torch.zeros(D)
        sd[ge + 'norm2.weight'] = torch.ones(D); sd[ge + 'norm2.bias'] = torch.zeros(D)
        sd['group_pos_embed'] = torch.zeros(1, patch + 1, D)
        sd['group_cls_token'] = torch.zeros(1, 1, D)
    else:
        sd['voxel_pos_embed'] = torch.zeros(1, n_tok + 1, D)
    if exercise_all:
        # test-only: make every zero/one-initialised tensor non-trivial so bias / LayerNorm-affine /
        # positional-embedding code paths are actually exercised by parity checks
        for k in sorted(sd):
            t = sd[k]
            if bool((t == 0).all()) or bool((t == 1).all()):
                if portable:
                    sd[k] = t + ((pu(tuple(t.shape)) * 2 - 1) * (0.05 * math.sqrt(3.0))).float()
                else:
                    sd[k] = t + torch.empty_like(t).normal_(0, 0.05, generator=g)
    return sd


def synthetic_class_batch(batch, voxel_size, cell, labels, seed=9, base=0.04, step=0.07):
    """A LEARNABLE portable synthetic batch (trained-state fixtures, tests/golden/make_golden_trained.py): the label is encoded in the
    grid's occupancy -- the i-th entry of `labels` fills the grid at density base + step i (0.04 + 0.07 i by default) -- which a freshly
    initialised model can pick up without positional information (voxel_pos_embed starts at zero, vit_3d_2d_pretrain.py:370-376).
    Labels are drawn uniformly by hash."""
    V = voxel_size
    labels = list(labels)
    pick = (portable_uniform((batch,), seed, 2002) * len(labels)).long().clamp_(max=len(labels) - 1)
    y = torch.tensor(labels, dtype=torch.long)[pick]
    u = portable_uniform((batch, 1, V, V, V), seed, 2001)
    dens = (base + step * pick.double()).view(batch, 1, 1, 1, 1)
    return (u < dens).to(torch.int32).float(), y


def synthetic_batch(batch, voxel_size, n_classes, seed=9, occupancy=0.10, portable=False):
    """SURVEY.md section 8(d) synthetic inputs: seeded 10 %-occupancy binary grid, int32 in the
    dataset (data/modelnet40.py:40), cast to float by the trainer (train_cls_voxel.py:276)."""
    if portable:
        V = voxel_size
        x = (portable_uniform((batch, 1, V, V, V), seed, 1001) < occupancy).to(torch.int32)
        y = (portable_uniform((batch,), seed, 1002) * n_classes).long().clamp_(max=n_classes - 1)
        return x.float(), y
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(batch, 1, voxel_size, voxel_size, voxel_size, generator=g) < occupancy).to(torch.int32)
    y = torch.randint(0, n_classes, (batch,), generator=g)
    return x.float(), y
