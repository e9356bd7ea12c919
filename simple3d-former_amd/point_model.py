"""Drop-in PointTransformerCls / PointTransformerSeg (reference: models/3DViT/model.py:144-337 and :341-535).

Same constructor (`Model(cfg)` with the hydra cfg fields the reference reads at :203-205,230 -- num_point, num_class,
input_dim, model.{nneighbor, transformer_backbone, pretrained, head} -- and the same write-back of cfg.embed_dim, :221),
same attribute tree and state_dict keys (incl. the parameters the reference creates but never uses: pos_embed,
patch_embed = PointEmbed, sa.last_pos_embed).  forward() runs the HIP PointEngine through one autograd.Function, so
`pred = classifier(points); loss = criterion(pred, target); loss.backward(); optimizer.step()` (train_cls.py:117-123,
train_partseg.py:143-152) works unchanged.  No CPU fallback.

`model_module(name)` stands in for `importlib.import_module('models.{}.model'.format(args.model.name))` (train_partseg.py:74,
train_partseg_lwf.py) for the four model directories: 3DViT (Cls + Seg) and the part-segmentation variants 3DViT_1_layer,
3DViT_0_layer, 3DViT_LWF, which keep timm's 2-D stem / `head`, name the point head `new_head`, take
`forward(x, type='points')` and add `forward_images(images)` (models/3DViT_1_layer/model.py:294-354)."""
from collections import OrderedDict
from functools import partial

import torch
from torch import nn

from .engine import BACKBONES
from .point_engine import VARIANTS, PointEngine, level_plan
from .voxel_model import VisionTransformer


class _Container(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f'{type(self).__name__} is a parameter container; its arithmetic runs in the HIP PointEngine')


class Local_op(_Container):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = nn.Conv1d(in_channels, out_channels, kernel_size=1, bias=False)
        self.conv2 = nn.Conv1d(out_channels, out_channels, kernel_size=1, bias=False)
        self.bn1, self.bn2 = nn.BatchNorm1d(out_channels), nn.BatchNorm1d(out_channels)
        self.relu = nn.ReLU()


class PointEmbed(_Container):
    """Created by the reference (model.py:226) but never called by forward_features; kept for checkpoint compatibility."""

    def __init__(self, cfg):
        super().__init__()
        self.conv1 = nn.Conv1d(cfg.input_dim, 64, kernel_size=1, bias=False)
        self.conv2 = nn.Conv1d(64, 64, kernel_size=1, bias=False)
        self.bn1, self.bn2 = nn.BatchNorm1d(64), nn.BatchNorm1d(64)
        self.gather_local_0 = Local_op(128, cfg.embed_dim // 4)
        self.gather_local_1 = Local_op(256, cfg.embed_dim // 4)
        self.relu = nn.ReLU()


class PointNetSetAbstraction(_Container):
    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all, knn=False):
        super().__init__()
        self.npoint, self.radius, self.nsample, self.knn, self.group_all = npoint, radius, nsample, knn, group_all
        self.mlp_convs, self.mlp_bns, self.pos_embeds = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        last = in_channel
        for out in mlp:
            self.mlp_convs.append(nn.Conv2d(last, out, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out))
            last = out
        self.last_pos_embed = nn.Sequential(nn.Linear(3, last), nn.ReLU(), nn.Linear(last, last))   # unused by forward


class TransitionDown(_Container):
    def __init__(self, k, nneighbor, channels):
        super().__init__()
        self.sa = PointNetSetAbstraction(k, 0, nneighbor, channels[0], channels[1:], group_all=False, knn=True)


class _SwapAxes(nn.Module):
    def forward(self, x):
        return x.transpose(1, 2)


class TransitionUp(_Container):
    def __init__(self, dim1, dim2, dim_out):
        super().__init__()
        self.fc1 = nn.Sequential(nn.Linear(dim1, dim_out), _SwapAxes(), nn.BatchNorm1d(dim_out), _SwapAxes(), nn.ReLU())
        self.fc2 = nn.Sequential(nn.Linear(dim2, dim_out), _SwapAxes(), nn.BatchNorm1d(dim_out), _SwapAxes(), nn.ReLU())


class _PointForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, starts, *params):
        eng = model._engine
        eng.refresh_weight_planes()
        out = eng.forward(x.contiguous().float(), starts, training=model.training).clone()
        ctx.model, ctx.batch = model, x.shape[0]
        return out

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        eng = model._engine
        ws = eng.workspace(ctx.batch)
        eng.zero_grad()
        if eng.task == 'cls':
            ws.dlogits.copy_(dlogits)
        else:
            ws.dlogits.zero_()
            ws.dlogits.view(ctx.batch, eng.N, -1)[..., :eng.ncls].copy_(dlogits)
        eng.backward(ctx.batch)
        # always COPIES of the arena slices: autograd keeps ("steals") the returned tensors as param.grad, and a view of the arena
        # would be wiped by the next backward's eng.zero_grad() under gradient accumulation / zero_grad(set_to_none=False), or
        # added to itself by AccumulateGrad; a second node (forward_images) also reuses the arena before autograd has accumulated
        two = eng.images is not None
        grads = [None if not need or (two and k.startswith(_IMAGE_ONLY)) else eng.arena.grad(k).clone()
                 for k, need in zip(eng.shapes, ctx.needs_input_grad[3:])]
        return (None, None, None) + tuple(grads)


_IMAGE_ONLY = ('patch_embed.', 'pos_embed', 'head.')              # not in the graph of model(points) (variants)
_POINT_ONLY = ('fc1.', 'fc_pos_embed.', 'transition_', 'new_head.')   # not in the graph of forward_images


class _PointImageForward(torch.autograd.Function):
    """PointTransformerSeg.forward_images(images) of the variants as one autograd node on the shared blocks."""

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
        grads = [eng.arena.grad(k).clone() if need and not k.startswith(_POINT_ONLY) else None
                 for k, need in zip(eng.shapes, ctx.needs_input_grad[2:])]
        return (None, None) + tuple(grads)


class _PointTransformer(VisionTransformer):
    _task = 'cls'
    _variant = '3DViT'

    def __init__(self, cfg):
        npoints, nneighbor, n_c, d_points = cfg.num_point, cfg.model.nneighbor, cfg.num_class, cfg.input_dim
        self.transformer_backbone = cfg.model.transformer_backbone
        self.pretrained = cfg.model.pretrained
        if self.transformer_backbone not in BACKBONES:
            raise ValueError("Unknown transformer backbone name!")
        if nneighbor != 16:
            raise ValueError('the HIP kNN kernel is built for nneighbor = 16 (config/model/3DViT.yaml)')
        bb = BACKBONES[self.transformer_backbone]
        super().__init__(patch_size=16, embed_dim=bb['embed_dim'], depth=bb['depth'], num_heads=bb['num_heads'], mlp_ratio=4,
                         qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6))
        self.dist_token = None
        self.n_classes = n_c
        cfg.embed_dim = bb['embed_dim']                                     # the reference writes this back (model.py:221)
        print(self.transformer_backbone)
        if self.pretrained:
            path = getattr(cfg.model, 'pretrained_path', None)
            if path is None:
                raise RuntimeError('pretrained=True needs the DeiT checkpoint; there is no network here -- set '
                                   'cfg.model.pretrained_path=<local .pth> or cfg.model.pretrained=False')
            ckpt = torch.load(path, map_location='cpu')
            sd = ckpt['model'] if 'model' in ckpt else ckpt
            own = self.state_dict()
            self.load_state_dict({k: v for k, v in sd.items() if k in own and v.shape == own[k].shape}, strict=False)
        vv = VARIANTS[self._variant]
        c0, keep, chans = level_plan(self._variant, self.embed_dim, npoints)
        if vv['image']:                       # the variants keep timm's PatchEmbed / pos_embed / head for forward_images and,
            if self.pretrained:               # with the DeiT checkpoint loaded, freeze head + patch_embed (3DViT_1_layer/model.py:285-289)
                self.head.weight.requires_grad = False
                self.head.bias.requires_grad = False
                for param in self.patch_embed.parameters():
                    param.requires_grad = False
            self.use_pos_embed = True
        else:
            self.patch_embed = PointEmbed(cfg)
        self._head_kind = 'AMSoftmax' if getattr(cfg.model, 'head', 'default') == 'AMSoftmax' else 'default'
        if self._head_kind == 'AMSoftmax':                 # models/3DViT/model.py:230-231, 427-428 (new_head in the variants, e.g. 3DViT_1_layer:218-219)
            if self._task != 'seg':
                raise ValueError("cfg.model.head == 'AMSoftmax' with PointTransformerCls fails in the reference itself: AMSoftmaxLayer.forward "
                                 "unpacks `B, N, C = x.shape` (models/3DViT/model.py:135) and the classifier feeds it the 2-D x.mean(1)")
            from .voxel_model import AMSoftmaxLayer
            setattr(self, vv['head'], AMSoftmaxLayer(c0, n_c))
        else:
            setattr(self, vv['head'], nn.Linear(c0, n_c))
        self.pos_embed_type = 'default'
        self.transition_downs = nn.ModuleList()
        for npoint, ch in zip(keep, chans):
            self.transition_downs.append(TransitionDown(npoint, nneighbor, [ch // 2 + 3, ch, ch]))
        self.transition_ups = nn.ModuleList()
        for ch in reversed(chans):
            self.transition_ups.append(TransitionUp(ch, ch // 2, ch // 2))
        self.fc1 = nn.Sequential(nn.Linear(d_points, c0), nn.ReLU(), nn.Linear(c0, c0))
        self.fc_pos_embed = nn.Sequential(nn.Linear(3, c0), nn.ReLU(), nn.Linear(c0, c0))
        self._cfg_io = (npoints, d_points)
        self._engine = None
        self.s3d_fps_starts = None           # optional (start0, start1) override of the random FPS start indices

    def s3d_engine(self, device=None):
        device = torch.device(device) if device is not None else next(self.parameters()).device
        if self._engine is not None and self._engine.device == device:
            return self._engine
        eng = PointEngine(backbone=self.transformer_backbone, n_points=self._cfg_io[0], d_points=self._cfg_io[1],
                          n_classes=self.n_classes, task=self._task, device=device, variant=self._variant, head=self._head_kind)
        own = dict(self.named_parameters())
        sd = {k: own[k].detach() for k in eng.shapes}
        bufs = dict(self.named_buffers())
        for k in eng.bns:
            sd[k + '.running_mean'], sd[k + '.running_var'] = bufs[k + '.running_mean'], bufs[k + '.running_var']
        eng.load_state_dict(sd)
        for k in eng.shapes:
            own[k].data = eng.arena.param(k)
        for k, bn in eng.bns.items():                      # module buffers alias the engine's running statistics
            bufs[k + '.running_mean'].data = bn.run_mean
            bufs[k + '.running_var'].data = bn.run_var
        if eng.images is not None:
            eng.images.frozen_head = not self.head.weight.requires_grad
            eng.images.frozen_stem = not self.patch_embed.proj.weight.requires_grad
            eng.images.frozen_pos = not self.pos_embed.requires_grad
        self._engine = eng
        return eng

    def forward(self, x, type='points'):
        if type != 'points':
            # the reference's other branch (3DViT_1_layer/model.py:350-352) applies self.head to forward_images' OUTPUT, i.e.
            # Linear(D, 1000) to a [B, 1000] tensor -- a shape error for every backbone; train_partseg_lwf.py:224 calls
            # forward_images directly
            raise RuntimeError("forward(x, type='images') fails in the reference as well (head applied twice); call forward_images(x)")
        if not x.is_cuda:
            raise RuntimeError(f'{self.__class__.__name__} runs on the HIP engine: move the model and the batch to the MI355X '
                               f'(no CPU fallback; the CPU reference lives in oracle/)')
        eng = self.s3d_engine(x.device)
        B, N = x.shape[0], x.shape[1]
        if self.s3d_fps_starts is not None:
            starts = tuple(s.to(x.device) for s in self.s3d_fps_starts)
        else:   # same draws as the reference: torch.randint on the default (CPU) generator, then moved (pointnet_util.py:65)
            starts = tuple(torch.randint(0, n, (B,), dtype=torch.long).to(x.device) for n in eng.Nin)
        own = dict(self.named_parameters())
        for m in self.transition_downs.modules():      # classifier.apply(bn_momentum_adjust) (train_partseg.py:97-99,130)
            if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
                eng.bn_momentum = m.momentum
                break
        out = _PointForward.apply(self, x, starts, *[own[k] for k in eng.shapes])
        if self.training:
            for k in eng.bns:
                dict(self.named_buffers())[k + '.num_batches_tracked'].add_(1)
        return out


    def forward_images(self, x):
        """3DViT_1_layer/model.py:323-337 (same in 3DViT_0_layer / 3DViT_LWF): PatchEmbed -> cls + pos_embed -> blocks -> norm -> head."""
        if not VARIANTS[self._variant]['image']:
            raise AttributeError(f"'{type(self).__name__}' object has no attribute 'forward_images'")   # models/3DViT/model.py has none
        if not x.is_cuda:
            raise RuntimeError(f'{type(self).__name__} runs on the HIP engine: move the model and the images to the MI355X '
                               f'(no CPU fallback)')
        eng = self.s3d_engine(x.device)
        own = dict(self.named_parameters())
        return _PointImageForward.apply(self, x, *[own[k] for k in eng.shapes])


class PointTransformerCls(_PointTransformer):
    _task = 'cls'


class PointTransformerSeg(_PointTransformer):
    _task = 'seg'


class PointTransformerSeg1Layer(_PointTransformer):
    """models/3DViT_1_layer/model.py: one TransitionDown (N/4 points, D channels) / TransitionUp pair, base width D/2."""
    _task, _variant = 'seg', '3DViT_1_layer'


class PointTransformerSeg0Layer(_PointTransformer):
    """models/3DViT_0_layer/model.py: every point is a token (fc1 maps straight to D); per-point head on the block output."""
    _task, _variant = 'seg', '3DViT_0_layer'


class PointTransformerSegLWF(_PointTransformer):
    """models/3DViT_LWF/model.py: two levels keeping N/4 and N/16 points, base width D/4, + forward_images."""
    _task, _variant = 'seg', '3DViT_LWF'


def model_module(name):
    """Stand-in for importlib.import_module('models.{}.model'.format(name)) (train_partseg.py:74): an object whose
    PointTransformerSeg (and, for '3DViT', PointTransformerCls) attribute is the drop-in class of that model directory."""
    import types
    table = {'3DViT': dict(PointTransformerCls=PointTransformerCls, PointTransformerSeg=PointTransformerSeg),
             '3DViT_1_layer': dict(PointTransformerSeg=PointTransformerSeg1Layer),
             '3DViT_0_layer': dict(PointTransformerSeg=PointTransformerSeg0Layer),
             '3DViT_LWF': dict(PointTransformerSeg=PointTransformerSegLWF)}
    if name not in table:
        raise ModuleNotFoundError(f"No module named 'models.{name}'")
    return types.SimpleNamespace(**table[name])
