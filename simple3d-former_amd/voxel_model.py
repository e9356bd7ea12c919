"""Drop-in model classes for the voxel path.

`VisionTransformer` keeps timm==0.3.2's constructor signature, attribute tree and state_dict keys (SURVEY.md
section 8(b)); `Feature3D_ViT2D_V2` keeps the reference's (models/vit_3d_2d_pretrain.py:275-526).  They are parameter
containers + API surface: `forward` hands the whole forward/backward to the HIP engine (engine.py) through one
torch.autograd.Function, so `pred = model(voxel); loss = F.cross_entropy(pred, y); loss.backward();
optimizer.step()` (train_cls_voxel.py:277-288) works unchanged.  There is no CPU fallback."""
from functools import partial

import torch
from torch import nn

from .engine import BACKBONES, VoxelEngine


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class _NoStandaloneForward(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f'{type(self).__name__} is a parameter container: its arithmetic is fused into the HIP '
                           f'engine and runs when the enclosing model is called')


class Mlp(_NoStandaloneForward):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)


class Attention(_NoStandaloneForward):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)


class Block(_NoStandaloneForward):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        if drop or attn_drop or drop_path:
            raise ValueError('the HIP engine implements the reference configuration (all drop rates 0)')
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer)


class PatchEmbed(nn.Module):
    """2-D image stem kept for checkpoint compatibility (only forward_images uses it; LwF branch, out of scope)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size, self.patch_size = to_2tuple(img_size), to_2tuple(patch_size)
        self.num_patches = (self.img_size[1] // self.patch_size[1]) * (self.img_size[0] // self.patch_size[0])
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., hybrid_backbone=None, norm_layer=nn.LayerNorm):
        super().__init__()
        if hybrid_backbone is not None or drop_rate or attn_drop_rate or drop_path_rate or mlp_ratio != 4 or not qkv_bias:
            raise ValueError('the HIP engine implements the reference configuration: no hybrid backbone, zero drop '
                             'rates, mlp_ratio 4, qkv_bias True')
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.depth, self.num_heads = depth, num_heads
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias, qk_scale, norm_layer=norm_layer)
                                     for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        trunc_normal_(self.pos_embed, std=.02)
        trunc_normal_(self.cls_token, std=.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)


class AMSoftmaxLayer(nn.Module):
    """Parameter container for the AM-softmax head (models/vit_3d_2d_pretrain.py:39-56); math runs in s3d_head_fwd."""

    def __init__(self, in_feats, n_classes, s=30.):
        super().__init__()
        self.s, self.in_feats = s, in_feats
        self.W = nn.Parameter(torch.randn(in_feats, n_classes))
        nn.init.xavier_normal_(self.W, gain=1)


def _fire_attention_hooks(model, bw):
    """visualize_attention_map_voxel.py:120-146 registers forward hooks on model.blocks[i].attn and recomputes the attention
    map from the hook's input (norm1(x)) with the module's own .qkv / .num_heads / .scale.  The blocks run inside the engine,
    so the hooks are fired here with the same (module, (input,), output) triple a torch Block would have produced: input =
    the saved LayerNorm-1 output, output = the attention branch's contribution x_mid - x_in.  No hooks -> no work."""
    for i, blk in enumerate(model.blocks):
        hooks = list(blk.attn._forward_hooks.values())
        if not hooks:
            continue
        with torch.no_grad():
            # norm1(x) recomputed from the saved fp32 block input (the engine keeps only the bf16 high plane per block)
            xn = torch.nn.functional.layer_norm(bw.x[i], (bw.x[i].shape[-1],), blk.norm1.weight, blk.norm1.bias,
                                                blk.norm1.eps).view(bw.Bb, bw.N, -1)
            out = (bw.x_mid[i] - bw.x[i]).view(bw.Bb, bw.N, -1)
        for h in hooks:
            h(blk.attn, (xn,), out)


_IMAGE_ONLY = ('patch_embed.', 'pos_embed', 'head.')                                  # not in the graph of model(voxel)
_VOXEL_ONLY = ('voxel_embed.', 'voxel_pos_embed', 'voxel_head.', 'group_')            # not in the graph of forward_images


GROUP_EMBED_DROPOUT = 0.1        # nn.TransformerEncoderLayer's default, which vit_3d_2d_pretrain.py:381 does not override


class _VoxelForward(torch.autograd.Function):
    """model(voxel) as one autograd node: forward/backward of the entire path run on the HIP engine."""

    @staticmethod
    def forward(ctx, model, x, *params):
        eng = model._engine
        eng.refresh_weight_planes()          # parameters may have been changed by a torch optimizer
        logits = eng.forward(x.contiguous().float()).clone()
        ctx.model, ctx.batch = model, x.shape[0]
        _fire_attention_hooks(model, eng.workspace(x.shape[0]).last)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        eng = model._engine
        eng.zero_grad()
        eng.backward(ctx.batch, dlogits.contiguous().float())
        grads = []
        for k, need in zip(eng.shapes, ctx.needs_input_grad[2:]):       # clones: a second node (forward_images) reuses the arena
            grads.append(eng.arena.grad(k).clone() if need and not k.startswith(_IMAGE_ONLY) else None)
        return (None, None) + tuple(grads)


class _ImageForward(torch.autograd.Function):
    """model.forward_images(images) as one autograd node (2-D branch on the same HIP engine and the same parameters)."""

    @staticmethod
    def forward(ctx, model, x, *params):
        eng = model._engine
        eng.refresh_weight_planes()
        logits = eng.images.forward(x.contiguous().float()).clone()
        ctx.model, ctx.batch = model, x.shape[0]
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        eng = ctx.model._engine
        eng.zero_grad()
        eng.images.backward(ctx.batch, dlogits.contiguous().float())
        grads = []
        for k, need in zip(eng.shapes, ctx.needs_input_grad[2:]):
            grads.append(eng.arena.grad(k).clone() if need and not k.startswith(_VOXEL_ONLY) else None)
        return (None, None) + tuple(grads)


class Feature3D_ViT2D_V2(VisionTransformer):
    """3-D voxel classifier on a 2-D-pretrained ViT backbone (reference: models/vit_3d_2d_pretrain.py:275-526)."""

    _url = {
        'deit_tiny_patch16_224': "https://dl.fbaipublicfiles.com/deit/deit_tiny_patch16_224-a1311bcf.pth",
        'deit_small_patch16_224': "https://dl.fbaipublicfiles.com/deit/deit_small_patch16_224-cd65a155.pth",
        'deit_base_patch16_224': "https://dl.fbaipublicfiles.com/deit/deit_base_patch16_224-b5f2ef4d.pth",
        'deit_base_distilled_patch16_224': "https://dl.fbaipublicfiles.com/deit/deit_base_distilled_patch16_224-df68dfff.pth",
        'vit_base_patch16_224_21k': "./3rd_party/B_16.pth",
    }

    def __init__(self, n_classes=10, embed_layer=None, data_shape=None, transformer_backbone='deit_base_patch16_224',
                 pretrained=True, pos_embedding=None, **kwargs):
        if transformer_backbone not in BACKBONES:
            raise ValueError("Unknown transformer backbone name!")
        cfg = BACKBONES[transformer_backbone]
        self.transformer_backbone, self.pretrained = transformer_backbone, pretrained
        super().__init__(patch_size=16, embed_dim=cfg['embed_dim'], depth=cfg['depth'], num_heads=cfg['num_heads'],
                         mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6))
        self.url = self._url[transformer_backbone]
        self.dist_token = None
        self.n_classes = n_classes
        print(self.transformer_backbone)
        self._load_backbone_weight(kwargs.get('pretrained_path'))
        self.voxel_embed = embed_layer
        self.voxel_head = AMSoftmaxLayer(self.embed_dim, n_classes) if kwargs.get('head') == 'AMSoftmax' \
            else nn.Linear(self.embed_dim, n_classes)
        self.head_type = 'AMSoftmax' if kwargs.get('head') == 'AMSoftmax' else 'default'
        self.pos_embed_type = pos_embedding
        if pos_embedding is None or pos_embedding == "default":
            # zeros and never random-initialised in the reference (trunc_normal_ is applied to pos_embed instead)
            self.voxel_pos_embed = nn.Parameter(torch.zeros(1, embed_layer.num_patches + 1, self.embed_dim))
            trunc_normal_(self.pos_embed, std=.02)
        elif pos_embedding == "group_embed":
            self.voxel_pos_embed = nn.Parameter(torch.zeros(1, embed_layer.patch_size ** 2 + 1, self.embed_dim))
            trunc_normal_(self.pos_embed, std=.02)
            self.group_embed = nn.TransformerEncoderLayer(d_model=self.embed_dim, dim_feedforward=self.embed_dim, nhead=4)
            self.group_pos_embed = nn.Parameter(torch.zeros(1, embed_layer.patch_size + 1, self.embed_dim))
            self.group_cls_token = nn.Parameter(torch.zeros(1, 1, self.embed_dim))
        elif pos_embedding in ("no_embed", "weight_sharing"):
            raise NotImplementedError(
                f"pos_embedding={pos_embedding!r}: 'no_embed' raises AttributeError in the reference's own forward "
                f"(voxel_pos_embed is never created) and 'weight_sharing' is outside the BASELINE configs")
        else:
            raise ValueError("Unknown positional embedding scheme!")
        self._engine = None

    def _load_backbone_weight(self, path):
        if not self.pretrained:
            return
        if path is None:
            raise RuntimeError(f'pretrained=True needs the DeiT checkpoint {self.url}; there is no network here -- pass '
                               f'pretrained_path=<local .pth> (same file format) or pretrained=False')
        ckpt = torch.load(path, map_location='cpu')
        sd = ckpt['model'] if 'model' in ckpt else ckpt
        own = self.state_dict()
        sd = {k: v for k, v in sd.items() if k in own}
        if 'distilled' in self.transformer_backbone and 'pos_embed' in sd:
            sd['pos_embed'] = sd['pos_embed'][:, 1:, :]
        self.load_state_dict(sd, strict=False)
        self.head.weight.requires_grad = False
        self.head.bias.requires_grad = False
        self.pos_embed.requires_grad = False
        for p in self.patch_embed.parameters():
            p.requires_grad = False

    # ------------------------------------------------------------------ engine plumbing
    def s3d_engine(self, device=None):
        """Builds (once per device) the HIP engine and re-points the used nn.Parameters into its flat arena."""
        device = torch.device(device) if device is not None else next(self.parameters()).device
        if self._engine is not None and self._engine.device == device:
            return self._engine
        te = self.voxel_embed
        eng = VoxelEngine(backbone=self.transformer_backbone, embed_layer=type(te).__name__,
                          voxel_size=te.voxel_size[0], cell=te.cell_size[0], patch=te.patch_size,
                          n_classes=self.n_classes, head=self.head_type, device=device, image_branch=True,
                          pos_embedding='group_embed' if self.pos_embed_type == 'group_embed' else 'default')
        eng.images.frozen = bool(self.pretrained)             # __load_backbone_weight freezes the 2-D stem / head
        own = dict(self.named_parameters())
        eng.load_state_dict({k: own[k].detach() for k in eng.shapes})
        for k in eng.shapes:                      # same Parameter objects (optimizers keep working), new storage
            own[k].data = eng.arena.param(k)
        self._engine = eng
        return eng

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError('Feature3D_ViT2D_V2 runs on the HIP engine: move the model and the voxel batch to the '
                               'MI355X (no CPU fallback; the CPU reference lives in oracle/)')
        eng = self.s3d_engine(x.device)
        if eng.group:
            # nn.TransformerEncoderLayer(dropout=0.1) inside group_embed (vit_3d_2d_pretrain.py:381): active under model.train(),
            # identity under model.eval(); every training forward draws fresh masks
            eng.set_dropout(GROUP_EMBED_DROPOUT if self.training else 0.0)
            if self.training:
                eng.advance_dropout_seed()
        own = dict(self.named_parameters())
        return _VoxelForward.apply(self, x, *[own[k] for k in eng.shapes])

    def forward_images(self, x):
        """The 2-D branch (vit_3d_2d_pretrain.py:435-451): PatchEmbed -> cls + pos_embed -> the same blocks -> norm -> head."""
        if not x.is_cuda:
            raise RuntimeError('Feature3D_ViT2D_V2 runs on the HIP engine: move the model and the images to the MI355X')
        eng = self.s3d_engine(x.device)
        own = dict(self.named_parameters())
        return _ImageForward.apply(self, x, *[own[k] for k in eng.shapes])
