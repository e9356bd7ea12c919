"""Drop-in 3-D tokenizer modules: same class names, constructor signature, attributes and state_dict keys as the
reference's models/embed_layer_3d_modality.py (VoxelEmbed :150-177, VoxelEmbed_no_average :43-70,
VoxelNaiveProjection :182-209).  Parameters live in a plain nn.Conv3d / nn.Conv2d container (checkpoint
compatibility); the arithmetic runs in the HIP tokenizer (s3d_voxel_fold + the split-bf16 patch GEMM)."""
import ctypes
from collections import OrderedDict

import torch
from torch import nn

from . import _lib as L


class _VoxelTokenizer(nn.Module):
    _conv_name = 'conv3d_1'
    _fold_mode = 0
    _grid_dims = 2           # output patch-grid rank (P x P or P x P x P)

    def __init__(self, voxel_size=128, cell_size=16, patch_size=8, in_chans=1, embed_dim=768):
        super().__init__()
        if in_chans != 1:
            raise ValueError('the HIP tokenizer supports single-channel occupancy grids (in_chans=1), as every '
                             'reference call site uses (train_cls_voxel.py:115-127)')
        self.voxel_size = (voxel_size,) * 3
        self.cell_size = (cell_size,) * 3
        self.patch_size = patch_size
        self.num_patches = patch_size ** self._grid_dims
        self.embed_dim = embed_dim
        conv = nn.Conv2d if self._conv_name == 'conv2d_1' else nn.Conv3d
        self.proj = nn.Sequential(OrderedDict([(self._conv_name, conv(in_channels=in_chans, out_channels=embed_dim,
                                                                     kernel_size=cell_size, stride=cell_size))]))

    @property
    def s3d_kind(self):
        return type(self).__name__

    def _check_input(self, x):
        B, C, H, W, V = x.shape
        assert H == self.voxel_size[0] and W == self.voxel_size[1] and V == self.voxel_size[2], \
            f"Input voxel size ({H}*{W}*{V}) doesn't match model ({self.voxel_size[0]}*{self.voxel_size[1]}*{self.voxel_size[2]})."

    def forward(self, x):
        """Stand-alone tokenizer forward (inference utility; inside Feature3D_ViT2D_V2 the tokenizer is fused with
        token assembly).  [B,1,V,V,V] fp32 on the device -> [B,D,P,P] (or [B,D,P,P,P])."""
        self._check_input(x)
        if not x.is_cuda:
            raise RuntimeError(f'{self.s3d_kind}: the HIP tokenizer needs a device tensor (no CPU fallback; the CPU '
                               f'reference lives in oracle/)')
        lib, s = L.lib(), L.current_stream()
        conv = self.proj[0]
        B, V, c, P, D = x.shape[0], self.voxel_size[0], self.cell_size[0], self.patch_size, self.embed_dim
        n = self.num_patches
        Kc = conv.weight[0].numel()
        Kp = (Kc + 7) // 8 * 8
        x = x.contiguous().float()
        a = torch.zeros(2, B * (n + 1), Kp, dtype=torch.bfloat16, device=x.device)
        fa = L.fill(L.S3dFoldArgs(), x=x, a_hi=a[0], a_lo=a[1], lda=Kp, B=B, V=V, c=c, P=P, mode=self._fold_mode)
        L.check(lib.s3d_voxel_fold(ctypes.byref(fa), s), 'voxel_fold')
        w = torch.zeros(2, D, Kp, dtype=torch.bfloat16, device=x.device)
        wf = conv.weight.detach().reshape(D, Kc).float().contiguous()
        L.check(lib.s3d_split_bf16(L.ptr(wf), L.ptr(w[0]), L.ptr(w[1]), ctypes.c_long(D), ctypes.c_long(Kc),
                                   ctypes.c_long(Kp), s), 'split')
        out = torch.empty(B * (n + 1), D, dtype=torch.float32, device=x.device)
        bias = conv.bias.detach().float().contiguous()
        g = L.fill(L.S3dGemmArgs(), A_hi=a[0], A_lo=a[1], lda=Kp, B_hi=w[0], B_lo=w[1], ldb=Kp, M=B * (n + 1), N=D, K=Kp,
                   bias=bias, C=out, ldc=D, alpha=(1.0 / P if self._fold_mode == 0 else 1.0))
        L.check(lib.s3d_gemm(0, 0, 1, 4, ctypes.byref(g), 1, s), 'tokenizer gemm')
        tok = out.view(B, n + 1, D)[:, 1:]                       # drop the (unused) cls slot of each sample
        return tok.transpose(1, 2).reshape(B, D, *([P] * self._grid_dims))


class VoxelEmbed(_VoxelTokenizer):
    """Conv3d(1->D, k=s=cell) then mean over the z-patch axis -> [B,D,P,P]; num_patches = P^2."""
    _fold_mode = 0


class VoxelEmbed_no_average(_VoxelTokenizer):
    """Conv3d(1->D, k=s=cell) -> [B,D,P,P,P]; num_patches = P^3."""
    _fold_mode = 2
    _grid_dims = 3


class VoxelNaiveProjection(_VoxelTokenizer):
    """clamp(sum over z, 0, 1) then Conv2d(1->D, k=s=cell) -> [B,D,P,P]; num_patches = P^2."""
    _conv_name = 'conv2d_1'
    _fold_mode = 1
