"""The 2-D branch of the model: Feature3D_ViT2D_V2.forward_images (models/vit_3d_2d_pretrain.py:435-451), used by the
learning-without-forgetting loop (train_cls_voxel.py:250-267): timm PatchEmbed (Conv2d(3, D, 16, stride 16)) -> cls concat ->
+ pos_embed -> the SAME 12 blocks as the voxel path -> norm -> 2-D head (1000 classes) on the cls row.

Everything runs on the library's kernels: s3d_image_patchify + the TOKEN-epilogue GEMM, s3d_blocks_fwd/bwd on a second activation
workspace (N = 197 tokens), the strided final norm, the linear head.  Gradients ACCUMULATE into the engine's arena, so a voxel
backward followed by an image backward yields d(CE_voxel + lambda * CE_image)/d(parameters) exactly as one autograd backward
does in the reference.  `frozen` mirrors __load_backbone_weight (:427-432): patch_embed / pos_embed / head get no gradient;
the point variants (models/3DViT_1_layer/model.py:285-289) freeze patch_embed and head but keep pos_embed trainable, hence the
three separate flags.  The same class serves PointEngine (PointTransformerSeg.forward_images of the 3DViT_* variants)."""
import ctypes

import torch

from . import _lib as L
from .engine import LN_EPS, _BlockScratch, _BlockWorkspace

IMG_SIZE, IMG_PATCH, IMG_CHANS, IMG_CLASSES = 224, 16, 3, 1000


def image_param_shapes(D):
    """Extra arena entries of the 2-D branch: (stem, head) in forward order."""
    stem = {'patch_embed.proj.weight': (D, IMG_CHANS, IMG_PATCH, IMG_PATCH), 'patch_embed.proj.bias': (D,),
            'pos_embed': (1, (IMG_SIZE // IMG_PATCH) ** 2 + 1, D)}
    head = {'head.weight': (IMG_CLASSES, D), 'head.bias': (IMG_CLASSES,)}
    return stem, head


class ImageBranch:
    def __init__(self, eng):
        self.eng, self.lib = eng, eng.lib
        self.np = (IMG_SIZE // IMG_PATCH) ** 2
        self.ntok = self.np + 1
        self.K = IMG_CHANS * IMG_PATCH * IMG_PATCH
        self.C = IMG_CLASSES
        self.frozen_stem = self.frozen_pos = self.frozen_head = False
        self._ws = {}

    @property
    def frozen(self):
        return self.frozen_stem and self.frozen_pos and self.frozen_head

    @frozen.setter
    def frozen(self, on):
        self.frozen_stem = self.frozen_pos = self.frozen_head = bool(on)

    def workspace(self, B):
        ws = self._ws.get(B)
        if ws is not None:
            return ws
        e, dev = self.eng, self.eng.device
        D, M = e.D, B * self.ntok
        f32 = dict(dtype=torch.float32, device=dev)
        ws = type('IWS', (), {})()
        ws.B, ws.M = B, M
        ws.a = torch.zeros(2, M, self.K, dtype=torch.bfloat16, device=dev)
        ws.blocks = _BlockWorkspace(e.depth, B, self.ntok, D, e.H, e.hidden, dev, e.split)
        ws.scratch = _BlockScratch(M, D, e.H, e.hidden, B * e.H * self.ntok, dev, depth=e.depth)
        ws.fstats = torch.empty(2, B, **f32)
        ws.feat = torch.empty(B, D, **f32)
        ws.logits = torch.empty(B, self.C, **f32)
        ws.dlogits = torch.empty(B, self.C, **f32)
        ws.dfeat = torch.empty(B, D, **f32)
        ws.loss = torch.zeros(2, **f32)
        ws.head_scratch = torch.empty(self.C + B, **f32)
        self._ws[B] = ws
        return ws

    def _head_args(self, ws):
        a = self.eng.arena
        return L.fill(L.S3dHeadArgs(), feat=ws.feat, B=ws.B, D=self.eng.D, C=self.C, W=a.param('head.weight'),
                      bias=a.param('head.bias'), logits=ws.logits, am_softmax=0, am_scale=1.0, dlogits=ws.dlogits,
                      dfeat=ws.dfeat, dW=None if self.frozen_head else a.grad('head.weight'),
                      dbias=None if self.frozen_head else a.grad('head.bias'), scratch=ws.head_scratch)

    # ------------------------------------------------------------------ forward
    def forward(self, img):
        """img: [B,3,224,224] float32 on the device -> logits [B,1000] (a workspace tensor, overwritten by the next call)."""
        assert img.is_cuda and img.dtype == torch.float32 and img.is_contiguous(), 'images must be a contiguous fp32 device tensor'
        B, Cc, H, W = img.shape
        assert Cc == IMG_CHANS and H == IMG_SIZE and W == IMG_SIZE, \
            f"Input image size ({H}*{W}) doesn't match model ({IMG_SIZE}*{IMG_SIZE})."          # timm PatchEmbed.forward
        e, lib, s = self.eng, self.lib, L.current_stream()
        a, D = e.arena, e.D
        ws = self.workspace(B)
        L.check(lib.s3d_image_patchify(L.ptr(img), L.ptr(ws.a[0]), L.ptr(ws.a[1]), ctypes.c_long(self.K), B, IMG_CHANS, H, W,
                                       IMG_PATCH, s), 'image_patchify')
        k = 'patch_embed.proj.weight'
        g = L.fill(L.S3dGemmArgs(), A_hi=ws.a[0], A_lo=ws.a[1], lda=self.K, B_hi=a.hi_of(k), B_lo=a.lo_of(k), ldb=self.K,
                   M=ws.M, N=D, K=self.K, bias=a.param('patch_embed.proj.bias'), C=ws.blocks.x[0], ldc=D, alpha=1.0,
                   cls=a.param('cls_token'), pos=a.param('pos_embed'), ntok=self.ntok)
        L.check(lib.s3d_gemm(0, 0, 1 if e.split else 0, 3, ctypes.byref(g), 1, s), 'patch-embed gemm')
        L.check(lib.s3d_blocks_fwd(ctypes.byref(ws.blocks.shape), e.bparams, ws.blocks.acts, e.depth, s), 'blocks_fwd (images)')
        ln = L.fill(L.S3dLnArgs(), x=ws.blocks.x[e.depth], ldx=self.ntok * D, rows=B, D=D, eps=LN_EPS,
                    gamma=a.param('norm.weight'), beta=a.param('norm.bias'), out_f32=ws.feat, ldo=D,
                    mean=ws.fstats[0], rstd=ws.fstats[1])
        L.check(lib.s3d_layernorm_fwd(ctypes.byref(ln), s), 'final norm (images)')
        L.check(lib.s3d_head_fwd(ctypes.byref(self._head_args(ws)), s), 'head_fwd (images)')
        return ws.logits

    def cross_entropy(self, B, target, weight=None, grad_scale=1.0):
        """F.cross_entropy(img_pred, label_teacher) (train_cls_voxel.py:266); grad_scale = lambda_weight folds the loss weight
        into d(loss)/d(logits).  Returns the UNSCALED mean NLL as a device scalar."""
        ws = self.workspace(B)
        assert target.dtype == torch.int64 and target.is_cuda
        ce = L.fill(L.S3dCeArgs(), logits=ws.logits, target=target, weight=weight, rows=B, C=self.C, loss=ws.loss,
                    dlogits=ws.dlogits, grad_scale=grad_scale)
        L.check(self.lib.s3d_cross_entropy(ctypes.byref(ce), L.current_stream()), 'cross_entropy (images)')
        return ws.loss[0]

    # ------------------------------------------------------------------ backward
    def backward(self, B, dlogits=None):
        """Accumulates d(image loss)/d(param) into the engine's gradient arena."""
        e, lib, s = self.eng, self.lib, L.current_stream()
        a, D = e.arena, e.D
        ws = self.workspace(B)
        if dlogits is not None and dlogits.data_ptr() != ws.dlogits.data_ptr():
            ws.dlogits.copy_(dlogits)
        L.check(lib.s3d_head_bwd(ctypes.byref(self._head_args(ws)), s), 'head_bwd (images)')
        sc = ws.scratch
        sc.zero_dx_a()
        lb = L.fill(L.S3dLnBwdArgs(), dy=ws.dfeat, lddy=D, x=ws.blocks.x[e.depth], ldx=self.ntok * D, mean=ws.fstats[0],
                    rstd=ws.fstats[1], gamma=a.param('norm.weight'), dx=sc.dx_a, lddx=self.ntok * D, dx_bf=sc.dx_a_bf,
                    lddxbf=self.ntok * D, dgamma=a.grad('norm.weight'), dbeta=a.grad('norm.bias'), rows=B, D=D)
        L.check(lib.s3d_layernorm_bwd(ctypes.byref(lb), s), 'final norm bwd (images)')
        L.check(lib.s3d_blocks_bwd(ctypes.byref(ws.blocks.shape), e.bparams, e.bgrads, ws.blocks.acts, ctypes.byref(sc.c),
                                   e.depth - 1, 0, s), 'blocks_bwd (images)')
        if not self.frozen_stem:                             # d(patch_embed.proj.weight) += dx^T patches
            g = L.fill(L.S3dGemmArgs(), A_hi=sc.dx_a_bf, lda=D, B_hi=ws.a[0], ldb=self.K, M=D, N=self.K, K=ws.M,
                       C=a.grad('patch_embed.proj.weight'), ldc=self.K, alpha=1.0)
            L.check(lib.s3d_gemm(1, 1, 0, 6, ctypes.byref(g), 0, s), 'patch-embed wgrad')
        pg = L.fill(L.S3dPosGradArgs(), dx=sc.dx_a, groups=B, ntok=self.ntok, D=D,
                    dpos=None if self.frozen_pos else a.grad('pos_embed'), dcls=a.grad('cls_token'),
                    dbias=None if self.frozen_stem else a.grad('patch_embed.proj.bias'))
        L.check(lib.s3d_token_grads(ctypes.byref(pg), s), 'image token grads')
