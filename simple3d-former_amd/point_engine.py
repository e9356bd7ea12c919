"""PointEngine: PointTransformerCls / PointTransformerSeg (reference models/3DViT/model.py:144-337, 341-535) training step
on one MI355X through the C ABI of libs3d_hip.so.

Pipeline (deit_tiny: D = 192, C0 = D/4 = 48):
  fc1(x) + fc_pos_embed(xyz)                       [B,N,C0]     two 2-layer MLPs              -> MFMA GEMMs
  TransitionDown 0: FPS(N) -> kNN16 -> gather -> conv1x1+BN+ReLU x2 -> max_k      [B,N,2C0]   s3d_fps/knn/group_project + GEMMs + batchnorm
                    (the first convolution is linear in a COPY of per-point features: Pf = feats . Wf^T once per point, then
                     Pf[idx] + xyz_rel . Wx^T + b per grouped row -- s3d_group_project_*, 16x fewer GEMM rows, no grouped operand)
  TransitionDown 1: FPS(N/4) -> ...                                               [B,N/4,D]
  cat cls -> 12 timm blocks -> LayerNorm -> drop cls                              s3d_blocks_fwd (shared with the voxel path)
  TransitionUp 0/1: Linear+BN+ReLU on both inputs, 3-NN inverse-distance interpolation, add
  cls: mean over points -> Linear head -> CE          seg: per-point Linear head -> CE over B*N rows
Train-mode BatchNorm (batch statistics, running-stat update).  The FPS start indices (torch.randint in the reference,
data/pointnet_util.py:65) are an explicit input so that runs are reproducible / comparable with the oracle.

`variant` selects one of the reference's four model directories (config/model/*.yaml `name`, imported by
train_partseg.py:74 / train_partseg_lwf.py): the same layers around the same transformer with 2, 1 or 0 TransitionDown /
TransitionUp pairs (VARIANTS below).  The three variants are part-segmentation only, keep timm's 2-D stem / `head` for
forward_images (models/3DViT_1_layer/model.py:323-337 -> ImageBranch on the shared blocks) and name the point head
`new_head`; `lwf_train_step` is the train_partseg_lwf.py:207-228 step."""
import contextlib
import ctypes
import math
import os

import numpy as np
import torch

from . import _lib as L
from .engine import BACKBONES, LN_EPS, ParamArena, _BlockScratch, _BlockWorkspace, _round_up

# blocks whose wgrads share one grouped full-K launch on the long-sequence dgrad chain (capi.hip: block_bwd_chain); 0 = paired launches
POINT_WGRAD_GROUP = int(os.environ.get('S3D_POINT_WGRAD_GROUP', '12'))

KNN = 16
BN_EPS = 1e-5
GEOM_STREAM = os.environ.get('S3D_POINT_GEOM_STREAM', '1') != '0'    # geometry chain on a side stream (see PointEngine._geometry)
# capture_train_step_pipelined: geometry and training step as TWO graphs replayed on two streams instead of one graph with a side
# branch -- on this runtime a captured graph with two live branches anywhere runs ALL of its nodes in a slower mode (+1.2 us per
# node on a chain of empty kernels, tools/probes/graph_branch_probe.py), a graph that is one dependency chain does not
TWO_CHAINS = os.environ.get('S3D_POINT_TWO_CHAINS', '1') != '0'

# levels: TransitionDown/Up pairs; first_div: td 0 keeps N / first_div points (models/3DViT/model.py:242 `npoints // 4 ** i`,
# the variants `npoints // 4 ** (i + 1)`, 3DViT_1_layer/model.py:231); head: key of the point head; image: forward_images exists.
VARIANTS = {
    '3DViT': dict(levels=2, first_div=1, head='head', image=False),
    '3DViT_LWF': dict(levels=2, first_div=4, head='new_head', image=True),
    '3DViT_1_layer': dict(levels=1, first_div=4, head='new_head', image=True),
    '3DViT_0_layer': dict(levels=0, first_div=1, head='new_head', image=True),
}


def level_plan(variant, D, n_points):
    """-> (C0 = D / 2**levels, [points kept by td i], [out channels of td i])."""
    if variant not in VARIANTS:
        raise ValueError(f'unknown point model {variant!r}; expected one of {sorted(VARIANTS)}')
    v = VARIANTS[variant]
    C0 = D >> v['levels']
    return C0, [n_points // (v['first_div'] * 4 ** i) for i in range(v['levels'])], [C0 * 2 ** (i + 1) for i in range(v['levels'])]


def point_param_shapes(backbone, n_classes, d_points, variant='3DViT', head='default'):
    cfg = BACKBONES[backbone]
    D, depth = cfg['embed_dim'], cfg['depth']
    vv = VARIANTS[variant]
    levels = vv['levels']
    C0 = D >> levels
    sh = {}
    if vv['image']:
        from .image_branch import image_param_shapes
        sh.update(image_param_shapes(D)[0])
    sh['fc1.0.weight'] = (C0, d_points); sh['fc1.0.bias'] = (C0,)
    sh['fc1.2.weight'] = (C0, C0); sh['fc1.2.bias'] = (C0,)
    sh['fc_pos_embed.0.weight'] = (C0, 3); sh['fc_pos_embed.0.bias'] = (C0,)
    sh['fc_pos_embed.2.weight'] = (C0, C0); sh['fc_pos_embed.2.bias'] = (C0,)
    for i in range(levels):
        ch = C0 * 2 ** (i + 1)
        cin = ch // 2 + 3
        p = f'transition_downs.{i}.sa.'
        sh[p + 'mlp_convs.0.weight'] = (ch, cin, 1, 1); sh[p + 'mlp_convs.0.bias'] = (ch,)
        sh[p + 'mlp_bns.0.weight'] = (ch,); sh[p + 'mlp_bns.0.bias'] = (ch,)
        sh[p + 'mlp_convs.1.weight'] = (ch, ch, 1, 1); sh[p + 'mlp_convs.1.bias'] = (ch,)
        sh[p + 'mlp_bns.1.weight'] = (ch,); sh[p + 'mlp_bns.1.bias'] = (ch,)
    sh['cls_token'] = (1, 1, D)
    for i in range(depth):
        p = f'blocks.{i}.'
        sh[p + 'norm1.weight'] = (D,); sh[p + 'norm1.bias'] = (D,)
        sh[p + 'attn.qkv.weight'] = (3 * D, D); sh[p + 'attn.qkv.bias'] = (3 * D,)
        sh[p + 'attn.proj.weight'] = (D, D); sh[p + 'attn.proj.bias'] = (D,)
        sh[p + 'norm2.weight'] = (D,); sh[p + 'norm2.bias'] = (D,)
        sh[p + 'mlp.fc1.weight'] = (4 * D, D); sh[p + 'mlp.fc1.bias'] = (4 * D,)
        sh[p + 'mlp.fc2.weight'] = (D, 4 * D); sh[p + 'mlp.fc2.bias'] = (D,)
    sh['norm.weight'] = (D,); sh['norm.bias'] = (D,)
    for j, i in enumerate(reversed(range(levels))):
        ch = C0 * 2 ** i
        p = f'transition_ups.{j}.'
        sh[p + 'fc1.0.weight'] = (ch, ch * 2); sh[p + 'fc1.0.bias'] = (ch,)
        sh[p + 'fc1.2.weight'] = (ch,); sh[p + 'fc1.2.bias'] = (ch,)
        sh[p + 'fc2.0.weight'] = (ch, ch); sh[p + 'fc2.0.bias'] = (ch,)
        sh[p + 'fc2.2.weight'] = (ch,); sh[p + 'fc2.2.bias'] = (ch,)
    hk = vv['head']
    if head == 'AMSoftmax':               # AMSoftmaxLayer.W [in_feats][n_classes] (models/3DViT/model.py:132, selected at :230-231 / :427-428)
        sh[hk + '.W'] = (C0, n_classes)
    else:
        sh[hk + '.weight'] = (n_classes, C0); sh[hk + '.bias'] = (n_classes,)
    if vv['image']:
        sh.update(image_param_shapes(D)[1])
    return sh


def bn_buffer_names(backbone, variant='3DViT'):
    names = []
    levels = VARIANTS[variant]['levels']
    for i in range(levels):
        for j in range(2):
            names.append(f'transition_downs.{i}.sa.mlp_bns.{j}')
    for j in range(levels):
        names += [f'transition_ups.{j}.fc1.2', f'transition_ups.{j}.fc2.2']
    return names


class _Linear:
    """y = x @ W^T + b with W [out][in] taken from the arena (padded bf16 planes when `in` or `out` need padding)."""

    def __init__(self, eng, key, out_pad=None):
        a = eng.arena
        shp = a.shapes[key + '.weight']
        self.key, self.out, self.inn = key, shp[0], int(np.prod(shp[1:]))
        self.kpad = _round_up(self.inn, 8)
        self.opad = out_pad or self.out
        self.padded = (self.kpad != self.inn) or (self.opad != self.out)
        dev = eng.device
        if self.padded:
            self.w = torch.zeros(2, self.opad, self.kpad, dtype=torch.bfloat16, device=dev)
            self.gpad = torch.zeros(self.opad, self.kpad, dtype=torch.float32, device=dev)
            self.bias = torch.zeros(self.opad, dtype=torch.float32, device=dev) if self.opad != self.out else a.param(key + '.bias')
            self.gbias = torch.zeros(self.opad, dtype=torch.float32, device=dev) if self.opad != self.out else a.grad(key + '.bias')
        else:
            self.w = (a.hi_of(key + '.weight'), a.lo_of(key + '.weight'))
            self.bias, self.gbias = a.param(key + '.bias'), a.grad(key + '.bias')
        self.eng = eng

    def refresh(self):
        if self.padded:
            a, lib = self.eng.arena, self.eng.lib
            w = a.param(self.key + '.weight')
            L.check(lib.s3d_split_bf16(L.ptr(w), L.ptr(self.w[0]), L.ptr(self.w[1]), ctypes.c_long(self.out),
                                       ctypes.c_long(self.inn), ctypes.c_long(self.kpad), L.current_stream()), 'split')
            if self.opad != self.out:
                self.bias[:self.out].copy_(a.param(self.key + '.bias'))

    def fwd(self, a_hi, a_lo, rows, epi, **kw):
        """a planes [rows][kpad]; epi in F32 / RELU / RESID ...; kw: C, ldc, O_hi, O_lo, ldo, aux, ldaux, R, ldr"""
        g = L.fill(L.S3dGemmArgs(), A_hi=a_hi, A_lo=a_lo, lda=self.kpad, B_hi=self.w[0], B_lo=self.w[1], ldb=self.kpad,
                   M=rows, N=self.opad, K=self.kpad, bias=self.bias, alpha=1.0, **kw)
        L.check(self.eng.lib.s3d_gemm(0, 0, 1 if self.eng.split else 0, epi, ctypes.byref(g), 1, L.current_stream()), self.key)

    def bwd(self, dy_bf, x_hi, rows, dx=None, dx_epi=4, **kw):
        """dy_bf [rows][opad] bf16, x_hi [rows][kpad]; accumulates dW/db; optional dx = dy @ W (epilogue dx_epi)."""
        lib, s = self.eng.lib, L.current_stream()
        a = self.eng.arena
        gw = self.gpad if self.padded else a.grad(self.key + '.weight')
        if self.padded:
            self.gpad.zero_()
            if self.opad != self.out:
                self.gbias.zero_()
        g = L.fill(L.S3dGemmArgs(), A_hi=dy_bf, lda=self.opad, B_hi=x_hi, ldb=self.kpad, M=self.opad, N=self.kpad, K=rows,
                   C=gw, ldc=self.kpad, alpha=1.0, bias_grad=self.gbias)
        L.check(lib.s3d_gemm(1, 1, 0, 6, ctypes.byref(g), 0, s), self.key + ' wgrad')
        if self.padded:
            a.grad(self.key + '.weight').view(self.out, self.inn).add_(self.gpad[:self.out, :self.inn])
            if self.opad != self.out:
                a.grad(self.key + '.bias').add_(self.gbias[:self.out])
        if dx is not None or kw:
            g = L.fill(L.S3dGemmArgs(), A_hi=dy_bf, lda=self.opad, B_hi=self.w[0], ldb=self.kpad, M=rows, N=self.kpad, K=self.opad,
                       alpha=1.0, **({'C': dx, 'ldc': self.kpad} if dx is not None else {}), **kw)
            L.check(lib.s3d_gemm(0, 1, 0, dx_epi, ctypes.byref(g), 1, s), self.key + ' dgrad')


class _AmHead:
    """AMSoftmaxLayer (models/3DViT/model.py:123-142; cfg.model.head == 'AMSoftmax', :427-428) as the per-point head of the
    part-segmentation models: logits = 30 * (x / |x|) @ (W / |W[:, c]|), W [C0][n_classes].  Runs as a row normalisation
    (s3d_l2norm_rows_fwd / _bwd) around the same GEMMs as the Linear head, on the derived weight Wl[c][d] = s * W[d][c] / |W[:, c]|
    (s3d_am_weight_fwd, refreshed with the other weight planes; its gradient is mapped back onto W by s3d_am_weight_bwd)."""
    SCALE = 30.0

    def __init__(self, eng, key, out_pad):
        a = eng.arena
        self.key, self.eng = key, eng
        self.inn, self.out = a.shapes[key + '.W']
        self.kpad, self.opad = _round_up(self.inn, 8), out_pad
        dev = eng.device
        self.wl = torch.zeros(self.opad, self.kpad, dtype=torch.float32, device=dev)          # Wl; pad rows / columns stay zero
        self.w = torch.zeros(2, self.opad, self.kpad, dtype=torch.bfloat16, device=dev)
        self.inv_w = torch.empty(self.out, dtype=torch.float32, device=dev)
        self.gpad = torch.zeros(self.opad, self.kpad, dtype=torch.float32, device=dev)
        self.bias = torch.zeros(self.opad, dtype=torch.float32, device=dev)                   # the layer has no bias

    def refresh(self):
        a, lib, s = self.eng.arena, self.eng.lib, L.current_stream()
        L.check(lib.s3d_am_weight_fwd(L.ptr(a.param(self.key + '.W')), self.inn, self.out, ctypes.c_float(self.SCALE), L.ptr(self.wl),
                                      self.kpad, L.ptr(self.inv_w), s), 'am_weight_fwd')
        L.check(lib.s3d_split_bf16(L.ptr(self.wl), L.ptr(self.w[0]), L.ptr(self.w[1]), ctypes.c_long(self.opad), ctypes.c_long(self.kpad),
                                   ctypes.c_long(self.kpad), s), 'split Wl')

    def normalise(self, x, rows, inv_norm, planes):
        """x fp32 [rows][C0] -> x / |x| as the GEMM operand planes [rows][kpad]; keeps 1 / |x| for the backward."""
        L.check(self.eng.lib.s3d_l2norm_rows_fwd(L.ptr(x), ctypes.c_long(self.inn), ctypes.c_long(rows), self.inn, L.ptr(inv_norm), L.ptr(planes[0]),
                                                 L.ptr(planes[1]), ctypes.c_long(self.kpad), L.current_stream()), 'l2norm_rows_fwd')

    def fwd(self, a_hi, a_lo, rows, epi, **kw):
        g = L.fill(L.S3dGemmArgs(), A_hi=a_hi, A_lo=a_lo, lda=self.kpad, B_hi=self.w[0], B_lo=self.w[1], ldb=self.kpad,
                   M=rows, N=self.opad, K=self.kpad, bias=self.bias, alpha=1.0, **kw)
        L.check(self.eng.lib.s3d_gemm(0, 0, 1 if self.eng.split else 0, epi, ctypes.byref(g), 1, L.current_stream()), self.key)

    def bwd(self, dy_bf, xn_hi, rows, x, inv_norm, dx):
        """dy_bf [rows][opad] bf16, xn_hi the normalised rows; accumulates dW; dx (fp32 [rows][C0]) = gradient wrt the UN-normalised x."""
        lib, s, a = self.eng.lib, L.current_stream(), self.eng.arena
        self.gpad.zero_()
        g = L.fill(L.S3dGemmArgs(), A_hi=dy_bf, lda=self.opad, B_hi=xn_hi, ldb=self.kpad, M=self.opad, N=self.kpad, K=rows,
                   C=self.gpad, ldc=self.kpad, alpha=1.0)
        L.check(lib.s3d_gemm(1, 1, 0, 6, ctypes.byref(g), 0, s), self.key + ' wgrad')
        L.check(lib.s3d_am_weight_bwd(L.ptr(self.gpad), self.kpad, L.ptr(a.param(self.key + '.W')), L.ptr(self.inv_w), self.inn, self.out,
                                      ctypes.c_float(self.SCALE), L.ptr(a.grad(self.key + '.W')), s), 'am_weight_bwd')
        g = L.fill(L.S3dGemmArgs(), A_hi=dy_bf, lda=self.opad, B_hi=self.w[0], ldb=self.kpad, M=rows, N=self.kpad, K=self.opad,
                   alpha=1.0, C=dx, ldc=self.kpad)
        L.check(lib.s3d_gemm(0, 1, 0, 4, ctypes.byref(g), 1, s), self.key + ' dgrad')
        L.check(lib.s3d_l2norm_rows_bwd(L.ptr(dx), ctypes.c_long(self.kpad), L.ptr(x), ctypes.c_long(self.inn), L.ptr(inv_norm), ctypes.c_long(rows),
                                        self.inn, L.ptr(dx), ctypes.c_long(self.kpad), s), 'l2norm_rows_bwd')


class _GroupProj:
    """First 1x1 convolution of a TransitionDown (mlp_convs.0, weight [ch][3 + C]) in its factored form: the feature columns
    Wf = W[:, 3:] become a per-point projection (MFMA GEMM over B*N rows), the xyz columns are applied per grouped row by
    s3d_group_project_fwd.  Holds the split-bf16 planes of Wf."""

    def __init__(self, eng, key, C, ch):
        self.eng, self.key, self.C, self.ch, self.cin = eng, key, C, ch, C + 3
        dev = eng.device
        self.wtmp = torch.empty(ch, C, dtype=torch.float32, device=dev)
        self.w = torch.zeros(2, ch, C, dtype=torch.bfloat16, device=dev)          # Wf   [ch][C]: forward B operand
        self.wt_tmp = torch.empty(C, ch, dtype=torch.float32, device=dev)
        self.wt = torch.zeros(2, C, ch, dtype=torch.bfloat16, device=dev)         # Wf^T [C][ch]: B operand of the split-precision dgrad

    def weight(self):
        return self.eng.arena.param(self.key + '.weight').view(self.ch, self.cin)

    def refresh(self):
        self.wtmp.copy_(self.weight()[:, 3:])
        L.check(self.eng.lib.s3d_split_bf16(L.ptr(self.wtmp), L.ptr(self.w[0]), L.ptr(self.w[1]), ctypes.c_long(self.ch),
                                            ctypes.c_long(self.C), ctypes.c_long(self.C), L.current_stream()), 'split Wf')
        self.wt_tmp.copy_(self.wtmp.t())
        L.check(self.eng.lib.s3d_split_bf16(L.ptr(self.wt_tmp), L.ptr(self.wt[0]), L.ptr(self.wt[1]), ctypes.c_long(self.C),
                                            ctypes.c_long(self.ch), ctypes.c_long(self.ch), L.current_stream()), 'split Wf^T')

    def args(self, t, xyz_in, B, **kw):
        a = self.eng.arena
        return L.fill(L.S3dGroupProjArgs(), xyz=xyz_in, new_xyz=t.new_xyz, idx=t.idx, B=B, N=t.Nin, S=t.S, K=KNN, C=self.ch,
                      W=a.param(self.key + '.weight'), ldw=self.cin, bias=a.param(self.key + '.bias'), ldp=self.ch, **kw)


class _BatchNorm:
    def __init__(self, eng, key, C):
        dev = eng.device
        self.eng, self.key, self.C = eng, key, C
        self.mean = torch.zeros(C, device=dev); self.rstd = torch.zeros(C, device=dev)
        self.run_mean = torch.zeros(C, device=dev); self.run_var = torch.ones(C, device=dev)
        self.sums = torch.zeros(2 * C, dtype=torch.float64, device=dev)      # forward statistics (re-pointed into the engine's flat block)
        self.sums_b = torch.zeros(2 * C, dtype=torch.float64, device=dev)    # backward statistics
        self.prezeroed = False              # the engine zeroes every layer's statistics with one memset per direction

    def _args(self, x, rows, K=0, **kw):
        a = self.eng.arena
        return L.fill(L.S3dBnArgs(), x=x, ldx=self.C, rows=rows, C=self.C, K=K, eps=BN_EPS, momentum=self.eng.bn_momentum,
                      momentum_dev=self.eng.hyper[3:4],
                      gamma=a.param(self.key + '.weight'), beta=a.param(self.key + '.bias'), mean=self.mean, rstd=self.rstd,
                      run_mean=self.run_mean, run_var=self.run_var, eval_mode=0 if self.eng.training else 1,
                      sums_zeroed=1 if self.prezeroed else 0, **kw)

    def fwd(self, x, rows, K=0, **kw):
        L.check(self.eng.lib.s3d_batchnorm_fwd(ctypes.byref(self._args(x, rows, K, sums=self.sums, **kw)), L.current_stream()), self.key)

    def bwd(self, x, rows, dy, dx, K=0, arg=None):
        """dy: fp32 or bf16 [rows (or groups)][C] gradient wrt the output"""
        a = self.eng.arena
        gkw = dict(dy_bf=dy) if dy.dtype == torch.bfloat16 else dict(dy=dy)
        args = self._args(x, rows, K, lddy=self.C, dx=dx, lddx=self.C, arg=arg, dgamma=a.grad(self.key + '.weight'),
                          dbeta=a.grad(self.key + '.bias'), sums=self.sums_b, **gkw)
        L.check(self.eng.lib.s3d_batchnorm_bwd(ctypes.byref(args), L.current_stream()), self.key + ' bwd')


class _TwoChainStep:
    """graphs[p] of PointEngine.capture_train_step_pipelined: replay() launches the geometry graph (next batch -> the other
    geometry set) on the geometry stream and the training-step graph (this batch, this set) on the current stream.  Each is one
    dependency chain; they overlap as two graph launches, ordered against each other at their boundaries only."""

    def __init__(self, geometry, step, stream):
        self.geometry, self.step, self.stream = geometry, step, stream

    def replay(self):
        main = torch.cuda.current_stream()
        main.wait_stream(self.stream)          # this batch's geometry (the previous replay's geometry graph) is complete
        self.stream.wait_stream(main)          # the previous step no longer reads the set the geometry graph overwrites
        with torch.cuda.stream(self.stream):
            self.geometry.replay()
        self.step.replay()


class PointEngine:
    def __init__(self, *, backbone='deit_tiny_patch16_224', n_points, d_points, n_classes, task='cls', device='cuda', split=True,
                 lr=0.01, momentum=0.9, bn_momentum=0.1, variant='3DViT', head='default'):
        if backbone not in BACKBONES:
            raise ValueError("Unknown transformer backbone name!")
        if task not in ('cls', 'seg'):
            raise ValueError(f'unknown task {task!r}')
        if variant not in VARIANTS:
            raise ValueError(f'unknown point model {variant!r}; expected one of {sorted(VARIANTS)}')
        if variant != '3DViT' and task != 'seg':
            raise ValueError(f'models/{variant}/model.py defines PointTransformerSeg only')
        if head not in ('default', 'AMSoftmax'):
            raise ValueError(f'unknown head {head!r}')
        if head == 'AMSoftmax' and task != 'seg':
            # AMSoftmaxLayer.forward unpacks `B, N, C = x.shape` (models/3DViT/model.py:135); PointTransformerCls hands it the 2-D
            # x.mean(1) (:325, :336), so the reference itself raises on the first forward -- there is nothing to be parity-compatible with
            raise ValueError("head='AMSoftmax' with the classification model fails in the reference itself (AMSoftmaxLayer.forward "
                             "needs a [B, N, C] input, models/3DViT/model.py:135); it is the per-point head of PointTransformerSeg")
        self.am = head == 'AMSoftmax'
        self.lib = L.lib()
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('PointEngine runs on an MI355X (cuda/HIP device) only; the CPU reference lives in oracle/')
        cfg = BACKBONES[backbone]
        self.D, self.depth, self.H = cfg['embed_dim'], cfg['depth'], cfg['num_heads']
        self.hidden = 4 * self.D
        self.variant, self.vv = variant, VARIANTS[variant]
        self.levels = self.vv['levels']
        self.N, self.dp, self.ncls, self.task = n_points, d_points, n_classes, task
        self.C0, self.S, self.ch = level_plan(variant, self.D, n_points)
        div = self.vv['first_div'] * 4 ** max(self.levels - 1, 0)
        assert n_points % div == 0 and n_points <= 2048, f'num_point must be a multiple of {div} and <= 2048'
        self.Nin = ([n_points] + self.S)[:self.levels]          # points entering td i
        assert all(n >= KNN for n in self.Nin), f'every TransitionDown needs >= {KNN} input points'
        self.cin = [c // 2 + 3 for c in self.ch]
        self.split = bool(split)
        # {lr, momentum, grad_scale, BatchNorm momentum} live on the DEVICE and the kernels read them at run time, so a captured HIP
        # graph follows the reference's per-epoch schedules (lr decay train_partseg.py:121-125, bn_momentum_adjust :126-130) and a
        # data-parallel trainer's 1/world without re-capture.  The attributes below are views of that block.
        self._hyper_host = [float(lr), float(momentum), 1.0, float(bn_momentum)]
        self.hyper = torch.tensor(self._hyper_host, dtype=torch.float32, device=self.device)
        self.training = True                # BatchNorm mode: batch statistics (model.train()) vs running statistics
        self.shapes = point_param_shapes(backbone, n_classes, d_points, variant, head)
        self.arena = ParamArena(self.shapes, self.device)
        self.buf = torch.zeros_like(self.arena.p)                   # SGD momentum buffer
        self.sgd_steps = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.world_size = 1
        a = self.arena
        # layers
        self.fc1 = [_Linear(self, 'fc1.0'), _Linear(self, 'fc1.2')]
        self.fcp = [_Linear(self, 'fc_pos_embed.0'), _Linear(self, 'fc_pos_embed.2')]
        self.td = []
        for i in range(self.levels):
            p = f'transition_downs.{i}.sa.'
            self.td.append(dict(gp=_GroupProj(self, p + 'mlp_convs.0', self.cin[i] - 3, self.ch[i]), b0=_BatchNorm(self, p + 'mlp_bns.0', self.ch[i]),
                                c1=_Linear(self, p + 'mlp_convs.1'), b1=_BatchNorm(self, p + 'mlp_bns.1', self.ch[i])))
        self.tu = []
        for j, i in enumerate(reversed(range(self.levels))):
            ch = self.C0 * 2 ** i
            p = f'transition_ups.{j}.'
            self.tu.append(dict(l1=_Linear(self, p + 'fc1.0'), b1=_BatchNorm(self, p + 'fc1.2', ch),
                                l2=_Linear(self, p + 'fc2.0'), b2=_BatchNorm(self, p + 'fc2.2', ch), ch=ch))
        self.head_key = self.vv['head']
        if self.am:
            self.head = _AmHead(self, self.head_key, _round_up(n_classes, 8))
        else:
            self.head = _Linear(self, self.head_key, out_pad=_round_up(n_classes, 8)) if task == 'seg' else None
        self._linears = self.fc1 + self.fcp + [t['gp'] for t in self.td] + [t['c1'] for t in self.td] + \
            [t[k] for t in self.tu for k in ('l1', 'l2')] + ([self.head] if self.head else [])
        self.bns = {t[k].key: t[k] for t in self.td for k in ('b0', 'b1')}
        self.bns.update({t[k].key: t[k] for t in self.tu for k in ('b1', 'b2')})
        # one flat block holds the fp64 statistics of every BatchNorm (forward halves first, then the backward halves): ONE memset at
        # the start of a training forward / of a backward instead of one per layer and pass (~14 five-microsecond launches per step)
        n2c = sum(2 * bn.C for bn in self.bns.values())
        self.bn_sums = torch.zeros(2 * n2c, dtype=torch.float64, device=self.device)
        off = 0
        for bn in (self.bns.values() if os.environ.get('S3D_BN_ONE_MEMSET', '1') != '0' else ()):      # =0: a memset per layer and pass
            bn.sums = self.bn_sums[off:off + 2 * bn.C]
            bn.sums_b = self.bn_sums[n2c + off:n2c + off + 2 * bn.C]
            bn.prezeroed = True
            off += 2 * bn.C
        self.bn_sums_f, self.bn_sums_b = self.bn_sums[:n2c], self.bn_sums[n2c:]
        # transformer block tables
        self.bparams = (L.S3dBlockParams * self.depth)()
        self.bgrads = (L.S3dBlockGrads * self.depth)()
        for i in range(self.depth):
            p = f'blocks.{i}.'
            L.fill(self.bparams[i], ln1_w=a.param(p + 'norm1.weight'), ln1_b=a.param(p + 'norm1.bias'),
                   ln2_w=a.param(p + 'norm2.weight'), ln2_b=a.param(p + 'norm2.bias'), qkv_b=a.param(p + 'attn.qkv.bias'),
                   proj_b=a.param(p + 'attn.proj.bias'), fc1_b=a.param(p + 'mlp.fc1.bias'), fc2_b=a.param(p + 'mlp.fc2.bias'),
                   qkv_w_hi=a.hi_of(p + 'attn.qkv.weight'), qkv_w_lo=a.lo_of(p + 'attn.qkv.weight'),
                   proj_w_hi=a.hi_of(p + 'attn.proj.weight'), proj_w_lo=a.lo_of(p + 'attn.proj.weight'),
                   fc1_w_hi=a.hi_of(p + 'mlp.fc1.weight'), fc1_w_lo=a.lo_of(p + 'mlp.fc1.weight'),
                   fc2_w_hi=a.hi_of(p + 'mlp.fc2.weight'), fc2_w_lo=a.lo_of(p + 'mlp.fc2.weight'))
            L.fill(self.bgrads[i], ln1_w=a.grad(p + 'norm1.weight'), ln1_b=a.grad(p + 'norm1.bias'), ln2_w=a.grad(p + 'norm2.weight'),
                   ln2_b=a.grad(p + 'norm2.bias'), qkv_w=a.grad(p + 'attn.qkv.weight'), qkv_b=a.grad(p + 'attn.qkv.bias'),
                   proj_w=a.grad(p + 'attn.proj.weight'), proj_b=a.grad(p + 'attn.proj.bias'), fc1_w=a.grad(p + 'mlp.fc1.weight'),
                   fc1_b=a.grad(p + 'mlp.fc1.bias'), fc2_w=a.grad(p + 'mlp.fc2.weight'), fc2_b=a.grad(p + 'mlp.fc2.bias'))
        self._ws = {}
        self.images = None
        if self.vv['image']:
            from .image_branch import ImageBranch
            self.images = ImageBranch(self)

    # ------------------------------------------------------------------ hyper-parameters (device-resident)
    def _set_hyper(self, i, v):
        v = float(v)
        if self._hyper_host[i] != v:
            self._hyper_host[i] = v
            self.hyper[i:i + 1].fill_(v)                 # stream-ordered: later launches AND graph replays see it

    lr = property(lambda self: self._hyper_host[0], lambda self, v: self._set_hyper(0, v))
    momentum = property(lambda self: self._hyper_host[1], lambda self, v: self._set_hyper(1, v))
    grad_scale = property(lambda self: self._hyper_host[2], lambda self, v: self._set_hyper(2, v))
    bn_momentum = property(lambda self: self._hyper_host[3], lambda self, v: self._set_hyper(3, v))

    def set_lr(self, lr):
        """Per-epoch learning-rate decay of the point trainers (train_partseg.py:121-125); takes effect on captured graphs too."""
        self.lr = lr

    def set_bn_momentum(self, m):
        """classifier.apply(bn_momentum_adjust) (train_partseg.py:126-130); takes effect on captured graphs too."""
        self.bn_momentum = m

    # ------------------------------------------------------------------ parameters / buffers
    def train_state(self):
        """Every tensor a training step mutates (parameters, gradients, momentum buffer, step flag, BatchNorm running statistics):
        what a capture warm-up has to snapshot and restore."""
        return [self.arena.p, self.arena.g, self.buf, self.sgd_steps] + self.bn_buffers()

    @contextlib.contextmanager
    def _preserved_state(self):
        """Warm-up steps of a graph capture run on whatever the static buffers hold: restore the training state afterwards so that
        capturing mid-training applies no spurious update (as VoxelEngine.capture_train_step does)."""
        state = self.train_state()
        snap = [t.clone() for t in state]
        try:
            yield
        finally:
            torch.cuda.synchronize()
            for t, sv in zip(state, snap):
                t.copy_(sv)
            self.refresh_weight_planes()
            torch.cuda.synchronize()

    def load_state_dict(self, sd):
        self.arena.load(sd)
        for k, bn in self.bns.items():
            if k + '.running_mean' in sd:
                bn.run_mean.copy_(sd[k + '.running_mean']); bn.run_var.copy_(sd[k + '.running_var'])
        self.refresh_weight_planes()

    def state_dict(self):
        sd = self.arena.state_dict()
        for k, bn in self.bns.items():
            sd[k + '.running_mean'] = bn.run_mean.clone(); sd[k + '.running_var'] = bn.run_var.clone()
        return sd

    def bn_buffers(self):
        """Running statistics of every BatchNorm (what DDP's constructor broadcast copies besides the parameters)."""
        return [t for bn in self.bns.values() for t in (bn.run_mean, bn.run_var)]

    def refresh_weight_planes(self):
        self.arena.refresh_planes()
        for l in self._linears:
            l.refresh()

    def zero_grad(self):
        self.arena.g.zero_()

    # ------------------------------------------------------------------ workspace
    def workspace(self, B):
        ws = self._ws.get(B)
        if ws is not None:
            return ws
        dev, D, C0, N = self.device, self.D, self.C0, self.N
        f32 = dict(dtype=torch.float32, device=dev); b16 = dict(dtype=torch.bfloat16, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        ws = type('WS', (), {})()
        ws.B = B
        BN = B * N
        dpp = _round_up(self.dp, 8)
        ws.xp = torch.zeros(2, BN, dpp, **b16); ws.xyzp = torch.zeros(2, BN, 8, **b16)
        ws.h1 = torch.empty(2, BN, C0, **b16); ws.h1pre = torch.empty(BN, C0, **b16)
        ws.h2 = torch.empty(2, BN, C0, **b16); ws.h2pre = torch.empty(BN, C0, **b16)
        ws.f = torch.empty(BN, C0, **f32); ws.fp = torch.empty(2, BN, C0, **b16)
        ws.td = []
        xyz_n = N
        cprev = C0
        for i in range(self.levels):
            S, ch = self.S[i], self.ch[i]
            R = B * S * KNN
            t = type('TD', (), {})()
            t.S, t.R, t.Nin, t.Cin = S, R, xyz_n, cprev
            t.fp = torch.empty(2, B * xyz_n, cprev, **b16)       # planes of the level's input features
            t.Pf = torch.empty(B * xyz_n, ch, **f32)              # feats . Wf^T per point
            t.dPf = torch.empty(B * xyz_n, ch, **f32); t.dPfp = torch.empty(2, B * xyz_n, ch, **b16)
            t.x1 = torch.empty(R, ch, **f32); t.y1 = torch.empty(2, R, ch, **b16)
            t.x2 = torch.empty(R, ch, **f32); t.out = torch.empty(B * S, ch, **f32)
            t.arg = torch.empty(B * S, ch, dtype=torch.uint8, device=dev)
            t.dx = torch.empty(R, ch, **b16)                  # bf16 gradient scratch (dx2 then dx1)
            t.dy1 = torch.empty(R, ch, **b16)                 # d(conv1 input) straight from the dgrad epilogue as bf16 (read twice by the BatchNorm backward)
            t.dout = torch.empty(B * S, ch, **f32)            # gradient wrt this level's output features
            ws.td.append(t)
            xyz_n, cprev = S, ch
        S1 = self.S[-1] if self.levels else N                   # points that become tokens
        ws.S1 = S1
        ws.ntok = S1 + 1
        M = B * ws.ntok
        ws.zero_pos = torch.zeros(ws.ntok, D, **f32)
        ws.blocks = _BlockWorkspace(self.depth, B, ws.ntok, D, self.H, 4 * D, dev, self.split)
        ring = None
        if POINT_WGRAD_GROUP > 0 and M > 8192:
            # the blocks' wgrads leave the backward chain: dy tensors kept in a ring, up to six blocks' weight gradients as one full-K launch
            # (capi.hip: block_bwd_chain, the long-sequence variant)
            slot = int(self.lib.s3d_block_wgrad_slot_bytes(ctypes.byref(ws.blocks.shape)))
            ring = (min(POINT_WGRAD_GROUP, 12, self.depth), 1, slot)
        ws.scratch = _BlockScratch(M, D, self.H, 4 * D, B * self.H * ws.ntok, dev, depth=self.depth, wgrad_ring=ring, ln_bwd_fuse=False)
        ws.nstats = torch.empty(2, M, **f32)
        ws.xn = torch.empty(M, D, **f32)                       # norm(x) incl. cls rows
        ws.t = torch.empty(B * S1, D, **f32); ws.tp = torch.empty(2, B * S1, D, **b16)
        ws.dxn = torch.empty(M, D, **f32); ws.dt = torch.empty(B * S1, D, **f32)
        ws.zero_cls = torch.zeros(D, **f32)
        ws.tu = []
        pts = [N] + self.S                                      # points per resolution, fine -> coarse
        lvl = [(pts[self.levels - j], pts[self.levels - j - 1], C0 * 2 ** (self.levels - 1 - j)) for j in range(self.levels)]
        for j, (Sc, Sf, ch) in enumerate(lvl):                  # (coarse pts, fine pts, channels)
            u = type('TU', (), {})()
            u.Sc, u.Sf, u.ch = Sc, Sf, ch
            u.u1 = torch.empty(B * Sc, ch, **f32); u.f1 = torch.empty(B * Sc, ch, **f32)
            u.u2 = torch.empty(B * Sf, ch, **f32); u.f2 = torch.empty(B * Sf, ch, **f32)
            u.out = torch.empty(B * Sf, ch, **f32)
            u.inp2 = torch.empty(2, B * Sf, ch, **b16)         # planes of the fine-level input (p0 / f)
            u.inp1 = torch.empty(2, B * Sc, 2 * ch, **b16) if j >= 1 else None     # planes of the coarse input (tu j-1's output)
            u.df1 = torch.empty(B * Sc, ch, **f32)
            u.dxb1 = torch.empty(B * Sc, ch, **b16); u.dxb2 = torch.empty(B * Sf, ch, **b16)
            u.din1 = torch.empty(B * Sc, 2 * ch, **f32)        # gradient wrt the coarse input
            ws.tu.append(u)
        ws.geo = [self._new_geometry(ws, B), None]              # geometry sets: [in use, prefetched for the next batch (lazily)]
        self._activate(ws, ws.geo[0])
        ws.dv1 = torch.empty(BN, C0, **f32)
        ws.df = torch.empty(BN, C0, **f32); ws.dfb = torch.empty(BN, C0, **b16)
        ws.dh = torch.empty(BN, C0, **b16)
        ws.loss = torch.zeros(2, **f32)
        if self.task == 'cls':
            ws.feat = torch.empty(B, C0, **f32); ws.dfeat = torch.empty(B, C0, **f32)
            ws.logits = torch.empty(B, self.ncls, **f32); ws.dlogits = torch.empty(B, self.ncls, **f32)
        else:
            cp = self.head.opad
            ws.v1p = torch.empty(2, BN, C0, **b16) if (self.levels or self.am) else None
            ws.inv_norm = torch.empty(BN, **f32) if self.am else None
            ws.logits = torch.empty(BN, cp, **f32); ws.dlogits = torch.empty(BN, cp, **f32)
            ws.dlb = torch.empty(BN, cp, **b16)
        self._ws[B] = ws
        return ws

    # ------------------------------------------------------------------ small helpers
    def _conv_bn(self, conv, bn, planes, rows, x, ch, **bn_kw):
        """1x1 convolution (GEMM, fp32 output x) + train-mode BatchNorm.  Where the GEMM runs on the 128x128 staged-epilogue kernels
        its epilogue also accumulates the column sums / sums of squares of x (S3dGemmArgs::col_sums), so the BatchNorm skips its
        own statistics pass over x (0.5 - 0.8 GB per level-0 tensor)."""
        fused = self.training and conv.opad == ch and bool(self.lib.s3d_gemm_col_sums_ok(1 if self.split else 0, int(rows), int(conv.opad)))
        if fused:
            if not bn.prezeroed:
                bn.sums.zero_()
            conv.fwd(planes[0], planes[1], rows, 4, C=x, ldc=ch, col_sums=bn.sums)
        else:
            conv.fwd(planes[0], planes[1], rows, 4, C=x, ldc=ch)
        bn.fwd(x, rows, have_sums=1 if fused else 0, **bn_kw)

    def _pack(self, x, C, ldx, rows, planes, lo=True):
        L.check(self.lib.s3d_pack_rows(L.ptr(x), C, ldx, ctypes.c_long(rows), L.ptr(planes[0]), L.ptr(planes[1]) if lo else None,
                                       planes.shape[-1], L.current_stream()), 'pack_rows')

    def _pack_bf(self, x, C, rows, out):
        L.check(self.lib.s3d_pack_rows(L.ptr(x), C, C, ctypes.c_long(rows), L.ptr(out), None, C, L.current_stream()), 'pack_rows')

    # ------------------------------------------------------------------ geometry
    def _new_geometry(self, ws, B):
        """Buffers for everything derived from the coordinates of ONE batch (the outputs of _geometry)."""
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev); i32 = dict(dtype=torch.int32, device=dev)
        g = type('Geometry', (), {})()
        g.xyz = torch.empty(B, self.N, 3, **f32)
        g.td, g.tu, g.events = [], [], [None] * (self.levels + 1)
        for t in ws.td:
            g.td.append(dict(fps_idx=torch.empty(B, t.S, **i32), new_xyz=torch.empty(B, t.S, 3, **f32), idx=torch.empty(B, t.S, KNN, **i32),
                             inv_off=torch.empty(B, t.Nin + 1, **i32), inv_rows=torch.empty(B, t.S * KNN, **i32)))   # transposed neighbour lists
        for u in ws.tu:
            g.tu.append(dict(idx=torch.empty(B, u.Sf, 3, **i32), w=torch.empty(B, u.Sf, 3, **f32)))
        return g

    @staticmethod
    def _activate(ws, g):
        """Points the workspace's geometry fields at set g (plain attribute rebinding: kernels read the pointers at launch)."""
        ws.xyz = g.xyz
        xyz_in = g.xyz
        for t, d in zip(ws.td, g.td):
            t.fps_idx, t.new_xyz, t.idx, t.inv_off, t.inv_rows = d['fps_idx'], d['new_xyz'], d['idx'], d['inv_off'], d['inv_rows']
            t.xyz_in = xyz_in
            xyz_in = t.new_xyz
        for u, d in zip(ws.tu, g.tu):
            u.idx, u.w = d['idx'], d['w']

    def _geometry(self, ws, B, x, starts, g, serial=False):
        """Everything that depends on the coordinates only: per level FPS -> kNN(16) -> transposed neighbour lists, then the 3-NN
        tables of the TransitionUps.  FPS is `npoint` strictly sequential iterations on ONE workgroup per cloud (32 - 128
        workgroups on 256 CUs) and the kNN kernels are short, so the chain runs on a side stream next to the feature path (the
        input MLPs, then level i's GEMMs / BatchNorms while level i+1's geometry is computed); the feature path waits on one
        event per level.  Works the same under HIP-graph capture (fork = wait_stream, join = the last wait_event).
        Fills geometry set g; g.events = [event per level..., event after the 3-NN tables] (None when the side stream is off,
        S3D_POINT_GEOM_STREAM=0)."""
        lib, nl = self.lib, self.levels
        if nl == 0:
            g.xyz.copy_(x[..., :3])
            return
        side = None
        if GEOM_STREAM and not serial:
            if getattr(self, '_side', None) is None:
                self._side = torch.cuda.Stream(device=self.device)
            side = self._side
            side.wait_stream(torch.cuda.current_stream())
        events = []
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            s = L.current_stream()
            g.xyz.copy_(x[..., :3])
            xyz_in = g.xyz
            for i in range(nl):
                t, d = ws.td[i], g.td[i]
                L.check(lib.s3d_fps(L.ptr(xyz_in), ctypes.c_long(3), L.ptr(starts[i]), B, t.Nin, t.S, L.ptr(d['fps_idx']), L.ptr(d['new_xyz']), s), 'fps')
                L.check(lib.s3d_knn(L.ptr(d['new_xyz']), L.ptr(xyz_in), B, t.S, t.Nin, KNN, L.ptr(d['idx']), None, s), 'knn')
                L.check(lib.s3d_neighbor_csr(L.ptr(d['idx']), B, t.Nin, t.S, KNN, L.ptr(d['inv_off']), L.ptr(d['inv_rows']), s), 'neighbor_csr')
                xyz_in = d['new_xyz']
                events.append(self._mark(side))
            res_xyz = [g.xyz] + [d['new_xyz'] for d in g.td]     # per resolution, fine -> coarse
            for j in range(nl):                                  # tu j interpolates resolution nl-j onto resolution nl-j-1
                u, d = ws.tu[j], g.tu[j]
                L.check(lib.s3d_knn(L.ptr(res_xyz[nl - j - 1]), L.ptr(res_xyz[nl - j]), B, u.Sf, u.Sc, 3, L.ptr(d['idx']), L.ptr(d['w']), s), 'knn3')
            events.append(self._mark(side))
        g.events = events

    @staticmethod
    def _mark(side):
        if side is None:
            return None
        ev = torch.cuda.Event()
        ev.record(side)
        return ev

    def forward(self, x, starts, training=True, geometry=None):
        """x [B,N,d_points] fp32 device tensor (xyz in the first 3 columns); starts = one int64 [B] tensor per TransitionDown.
        training=False normalises with the BatchNorm running statistics (model.eval()); backward needs training=True.
        geometry: a set prepared earlier for THIS batch (prepare_geometry / train_step_pipelined), already joined into the
        current stream -- the forward then skips FPS / kNN and waits for nothing."""
        self.training = bool(training)
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        B, N, dp = x.shape
        assert N == self.N and dp == self.dp, f'input {tuple(x.shape)} does not match the model (N={self.N}, d={self.dp})'
        ws = self.workspace(B)
        lib, s, a, C0, D = self.lib, L.current_stream(), self.arena, self.C0, self.D
        BN = B * N
        if self.training:
            self.bn_sums_f.zero_()              # every layer's forward statistics (gather kernel / GEMM epilogue / statistics pass add into them)
        if geometry is None:
            assert len(starts) >= self.levels, f'{self.variant} needs {self.levels} FPS start tensors'
            geometry = ws.geo[0]
            self._geometry(ws, B, x, starts, geometry)
            geom = geometry.events
        else:
            geom = [None] * (self.levels + 1)
        self._activate(ws, geometry)
        # fc1(x) + fc_pos_embed(xyz)
        self._pack(x, dp, dp, BN, ws.xp)
        self._pack(x, 3, dp, BN, ws.xyzp)
        self.fc1[0].fwd(ws.xp[0], ws.xp[1], BN, 7, O_hi=ws.h1[0], O_lo=ws.h1[1], ldo=C0, aux=ws.h1pre, ldaux=C0)
        self.fc1[1].fwd(ws.h1[0], ws.h1[1], BN, 4, C=ws.f, ldc=C0)
        self.fcp[0].fwd(ws.xyzp[0], ws.xyzp[1], BN, 7, O_hi=ws.h2[0], O_lo=ws.h2[1], ldo=C0, aux=ws.h2pre, ldaux=C0)
        self.fcp[1].fwd(ws.h2[0], ws.h2[1], BN, 2, C=ws.f, ldc=C0, R=ws.f, ldr=C0)
        # transition downs
        feats, cin_feats = ws.f, C0
        for i in range(self.levels):
            t, lay = ws.td[i], self.td[i]
            if geom[i] is not None:
                torch.cuda.current_stream().wait_event(geom[i])          # this level's FPS / kNN / transposed lists are ready
            ch, gp, xyz_in = self.ch[i], lay['gp'], t.xyz_in
            self._pack(feats, cin_feats, cin_feats, B * t.Nin, t.fp)
            g = L.fill(L.S3dGemmArgs(), A_hi=t.fp[0], A_lo=t.fp[1], lda=cin_feats, B_hi=gp.w[0], B_lo=gp.w[1], ldb=cin_feats,
                       M=B * t.Nin, N=ch, K=cin_feats, C=t.Pf, ldc=ch, alpha=1.0)
            L.check(lib.s3d_gemm(0, 0, 1 if self.split else 0, 4, ctypes.byref(g), 1, s), 'per-point projection')
            fused = self.training                    # the gather kernel also accumulates the BatchNorm statistics of x1
            if fused and not lay['b0'].prezeroed:
                lay['b0'].sums.zero_()
            L.check(lib.s3d_group_project_fwd(ctypes.byref(gp.args(t, xyz_in, B, Pf=t.Pf, x=t.x1, ldx=ch,
                                                                   sums=lay['b0'].sums if fused else None)), s), 'group_project_fwd')
            lay['b0'].fwd(t.x1, t.R, y_hi=t.y1[0], y_lo=t.y1[1], ldo=ch, have_sums=1 if fused else 0)
            self._conv_bn(lay['c1'], lay['b1'], t.y1, t.R, t.x2, ch, K=KNN, y=t.out, arg=t.arg)
            feats, cin_feats = t.out, ch
        # tokens -> blocks -> norm -> drop cls
        S1 = ws.S1
        L.check(lib.s3d_assemble_tokens(L.ptr(feats), L.ptr(a.param('cls_token')), L.ptr(ws.zero_pos), L.ptr(ws.blocks.x[0]),
                                        ctypes.c_long(B), S1, D, s), 'assemble')
        L.check(lib.s3d_blocks_fwd(ctypes.byref(ws.blocks.shape), self.bparams, ws.blocks.acts, self.depth, s), 'blocks_fwd')
        M = B * ws.ntok
        ln = L.fill(L.S3dLnArgs(), x=ws.blocks.x[self.depth], ldx=D, rows=M, D=D, eps=LN_EPS, gamma=a.param('norm.weight'),
                    beta=a.param('norm.bias'), out_f32=ws.xn, ldo=D, mean=ws.nstats[0], rstd=ws.nstats[1])
        L.check(lib.s3d_layernorm_fwd(ctypes.byref(ln), s), 'final norm')
        L.check(lib.s3d_assemble_tokens_bwd(L.ptr(ws.xn), L.ptr(ws.t), ctypes.c_long(B), S1, D, s), 'drop cls')
        self._pack(ws.t, D, D, B * S1, ws.tp)
        # transition ups
        coarse_planes = ws.tp
        nl = self.levels
        res_feats = [ws.f] + [t.out for t in ws.td]
        fine_feats = [res_feats[nl - j - 1] for j in range(nl)]
        if nl and geom[nl] is not None:
            torch.cuda.current_stream().wait_event(geom[nl])             # 3-NN tables (and the join of the side stream)
        for j in range(nl):
            u, lay = ws.tu[j], self.tu[j]
            ch = u.ch
            self._conv_bn(lay['l1'], lay['b1'], coarse_planes, B * u.Sc, u.u1, ch, y=u.f1, ldo=ch)
            self._pack(fine_feats[j], ch, ch, B * u.Sf, u.inp2)
            self._conv_bn(lay['l2'], lay['b2'], u.inp2, B * u.Sf, u.u2, ch, y=u.f2, ldo=ch)
            L.check(lib.s3d_interp3(L.ptr(u.f1), u.Sc, L.ptr(u.f2), L.ptr(u.idx), L.ptr(u.w), B, u.Sf, ch, L.ptr(u.out), s), 'interp3')
            if j + 1 < nl:
                self._pack(u.out, ch, ch, B * u.Sf, ws.tu[j + 1].inp1)
                coarse_planes = ws.tu[j + 1].inp1
        if nl == 0:                                              # 3DViT_0_layer: the head reads the tokens (C0 = D)
            if self.am:
                self.head.normalise(ws.t, BN, ws.inv_norm, ws.v1p)
            else:
                ws.v1p = ws.tp
            self.head.fwd(ws.v1p[0], ws.v1p[1], BN, 4, C=ws.logits, ldc=self.head.opad)
            return ws.logits.view(B, N, self.head.opad)[..., :self.ncls]
        v1 = ws.tu[-1].out
        if self.task == 'cls':
            L.check(lib.s3d_mean_points(L.ptr(v1), B, N, C0, L.ptr(ws.feat), s), 'mean')
            L.check(lib.s3d_head_fwd(ctypes.byref(self._head_args(ws)), s), 'head_fwd')
            return ws.logits
        if self.am:
            self.head.normalise(v1, BN, ws.inv_norm, ws.v1p)
        else:
            self._pack(v1, C0, C0, BN, ws.v1p)
        self.head.fwd(ws.v1p[0], ws.v1p[1], BN, 4, C=ws.logits, ldc=self.head.opad)
        return ws.logits.view(B, N, self.head.opad)[..., :self.ncls]

    def _head_args(self, ws):
        a = self.arena
        hk = self.head_key
        return L.fill(L.S3dHeadArgs(), feat=ws.feat, B=ws.B, D=self.C0, C=self.ncls, W=a.param(hk + '.weight'), bias=a.param(hk + '.bias'),
                      logits=ws.logits, am_softmax=0, am_scale=1.0, dlogits=ws.dlogits, dfeat=ws.dfeat, dW=a.grad(hk + '.weight'),
                      dbias=a.grad(hk + '.bias'))

    # ------------------------------------------------------------------ loss
    def cross_entropy(self, B, target):
        ws = self.workspace(B)
        rows = B if self.task == 'cls' else B * self.N
        ld = 0 if self.task == 'cls' else self.head.opad
        tgt = target.reshape(-1)
        assert tgt.dtype == torch.int64 and tgt.is_cuda and tgt.is_contiguous()
        ce = L.fill(L.S3dCeArgs(), logits=ws.logits, target=tgt, rows=rows, C=self.ncls, loss=ws.loss, dlogits=ws.dlogits,
                    grad_scale=1.0, ld=ld)
        L.check(self.lib.s3d_cross_entropy(ctypes.byref(ce), L.current_stream()), 'cross_entropy')
        return ws.loss[0]

    # ------------------------------------------------------------------ backward
    def backward(self, B):
        self.backward_top(B)
        self.backward_bottom(B)

    def grad_split(self):
        """Arena offset where the gradients written by backward_top start (cls_token, blocks, norm, TransitionUps, heads):
        everything at or above it is final when backward_top returns, so a data-parallel trainer can all-reduce that slice
        while backward_bottom (TransitionDowns + the input MLPs) still runs."""
        return self.arena.offsets['cls_token']

    def backward_top(self, B):
        """head -> TransitionUps -> final norm -> 12 blocks -> cls-token gradient."""
        ws = self.workspace(B)
        lib, s, a, C0, D, N = self.lib, L.current_stream(), self.arena, self.C0, self.D, self.N
        BN = B * N
        self.bn_sums_b.zero_()                  # every layer's backward statistics (backward_bottom runs after this call)
        # head
        if self.task == 'cls':
            L.check(lib.s3d_head_bwd(ctypes.byref(self._head_args(ws)), s), 'head_bwd')
            L.check(lib.s3d_bcast_rows(L.ptr(ws.dfeat), N, C0, ctypes.c_long(BN), ctypes.c_float(1.0 / N), L.ptr(ws.dv1), s), 'bcast')
        else:
            cp = self.head.opad
            self._pack_bf(ws.dlogits, cp, BN, ws.dlb)
            if self.am:
                self.head.bwd(ws.dlb, ws.v1p[0], BN, ws.tu[-1].out if self.levels else ws.t, ws.inv_norm, ws.dv1 if self.levels else ws.dt)
            else:
                self.head.bwd(ws.dlb, ws.v1p[0], BN, dx=ws.dv1 if self.levels else ws.dt, dx_epi=4)
        # transition ups (reverse)
        nl = self.levels
        dfine = ws.dv1                                           # gradient wrt the last tu's output [BN, C0]
        for j in reversed(range(nl)):
            u, lay = ws.tu[j], self.tu[j]
            ch = u.ch
            u.df1.zero_()
            L.check(lib.s3d_interp3_bwd(L.ptr(dfine), L.ptr(u.idx), L.ptr(u.w), B, u.Sc, u.Sf, ch, L.ptr(u.df1), s), 'interp3_bwd')
            # fine branch: f2 = relu(bn(l2(inp2)));  d(f2) = dfine
            lay['b2'].bwd(u.u2, B * u.Sf, dfine, u.dxb2)
            dfine_in = ws.df if j == nl - 1 else ws.td[nl - 2 - j].dout      # gradient wrt the fine-level input features (f / p_i)
            lay['l2'].bwd(u.dxb2, u.inp2[0], B * u.Sf, dx=dfine_in, dx_epi=4)
            # coarse branch
            lay['b1'].bwd(u.u1, B * u.Sc, u.df1, u.dxb1)
            coarse_x = u.inp1[0] if j >= 1 else ws.tp[0]
            dcoarse = u.din1 if j >= 1 else ws.dt
            lay['l1'].bwd(u.dxb1, coarse_x, B * u.Sc, dx=dcoarse, dx_epi=4)
            dfine = dcoarse                                      # tu j's coarse input is tu j-1's output
        # drop-cls backward -> final norm -> blocks
        S1 = ws.S1
        M = B * ws.ntok
        L.check(lib.s3d_assemble_tokens(L.ptr(ws.dt), L.ptr(ws.zero_cls), L.ptr(ws.zero_pos), L.ptr(ws.dxn), ctypes.c_long(B), S1, D, s), 'undrop')
        sc = ws.scratch
        lb = L.fill(L.S3dLnBwdArgs(), dy=ws.dxn, lddy=D, x=ws.blocks.x[self.depth], ldx=D, mean=ws.nstats[0], rstd=ws.nstats[1],
                    gamma=a.param('norm.weight'), dx=sc.dx_a, lddx=D, dx_bf=sc.dx_a_bf, lddxbf=D, dgamma=a.grad('norm.weight'),
                    dbeta=a.grad('norm.bias'), rows=M, D=D)
        L.check(lib.s3d_layernorm_bwd(ctypes.byref(lb), s), 'final norm bwd')
        L.check(lib.s3d_blocks_bwd(ctypes.byref(ws.blocks.shape), self.bparams, self.bgrads, ws.blocks.acts, ctypes.byref(sc.c),
                                   self.depth - 1, 0, s), 'blocks_bwd')
        pg = L.fill(L.S3dPosGradArgs(), dx=sc.dx_a, groups=B, ntok=ws.ntok, D=D, dcls=a.grad('cls_token'))
        L.check(lib.s3d_token_grads(ctypes.byref(pg), s), 'cls grad')
        dtok = ws.td[-1].dout if nl else ws.df                   # 3DViT_0_layer: the tokens are f itself
        L.check(lib.s3d_assemble_tokens_bwd(L.ptr(sc.dx_a), L.ptr(dtok), ctypes.c_long(B), S1, D, s), 'tokens bwd')

    def backward_bottom(self, B):
        """TransitionDowns (reverse) -> fc1 / fc_pos_embed."""
        ws = self.workspace(B)
        lib, s, C0, N, nl = self.lib, L.current_stream(), self.C0, self.N, self.levels
        BN = B * N
        # transition downs (reverse): dout of the top level is complete; the lower levels' dout already hold their tu's contribution
        for i in reversed(range(nl)):
            t, lay = ws.td[i], self.td[i]
            ch = self.ch[i]
            lay['b1'].bwd(t.x2, t.R, t.dout, t.dx, K=KNN, arg=t.arg)
            lay['c1'].bwd(t.dx, t.y1[0], t.R, dx_epi=0, O_hi=t.dy1, ldo=ch)
            lay['b0'].bwd(t.x1, t.R, t.dy1, t.dx)
            gp, a = lay['gp'], self.arena
            L.check(lib.s3d_group_project_bwd(ctypes.byref(gp.args(t, t.xyz_in, B, dx=t.dx, lddx=ch, dPf=t.dPf, inv_off=t.inv_off,
                                                                   inv_rows=t.inv_rows,
                                                                   dW=a.grad(gp.key + '.weight'), dbias=a.grad(gp.key + '.bias'))), s),
                    'group_project_bwd')
            rows = B * t.Nin
            self._pack(t.dPf, ch, ch, rows, t.dPfp)
            gwf = a.grad(gp.key + '.weight').view(ch, gp.cin)[:, 3:]          # d(Wf) lives inside the conv weight's gradient
            g = L.fill(L.S3dGemmArgs(), A_hi=t.dPfp[0], lda=ch, B_hi=t.fp[0], ldb=t.Cin, M=ch, N=t.Cin, K=rows, C=gwf, ldc=gp.cin, alpha=1.0)
            L.check(lib.s3d_gemm(1, 1, 0, 6, ctypes.byref(g), 0, s), 'Wf wgrad')
            dprev = ws.td[i - 1].dout if i >= 1 else ws.df                     # already holds the TransitionUp's contribution
            # dfeats += dPf . Wf in split precision (dPf is a SUM of ~16 bf16 rows; rounding it to one bf16 plane put the sampled
            # gradients of the input MLPs at the edge of the parity bar), as an NT product against Wf^T
            g = L.fill(L.S3dGemmArgs(), A_hi=t.dPfp[0], A_lo=t.dPfp[1], lda=ch, B_hi=gp.wt[0], B_lo=gp.wt[1], ldb=ch, M=rows, N=t.Cin,
                       K=ch, C=dprev, ldc=t.Cin, R=dprev, ldr=t.Cin, alpha=1.0)
            L.check(lib.s3d_gemm(0, 0, 1, 2, ctypes.byref(g), 1, s), 'per-point dgrad')
        # the two input MLPs: f = fc1(x) + fc_pos_embed(xyz)
        self._pack_bf(ws.df, C0, BN, ws.dfb)
        self.fc1[1].bwd(ws.dfb, ws.h1[0], BN, dx_epi=8, O_hi=ws.dh, ldo=C0, aux=ws.h1pre, ldaux=C0)
        self.fc1[0].bwd(ws.dh, ws.xp[0], BN)
        self.fcp[1].bwd(ws.dfb, ws.h2[0], BN, dx_epi=8, O_hi=ws.dh, ldo=C0, aux=ws.h2pre, ldaux=C0)
        self.fcp[0].bwd(ws.dh, ws.xyzp[0], BN)

    # ------------------------------------------------------------------ optimizer
    def sgd_step(self):
        """torch.optim.SGD(lr=0.01, momentum=0.9) (train_cls.py:91) + weight-plane refresh + gradient zeroing."""
        a = self.arena
        L.check(self.lib.s3d_sgd_step_dev(L.ptr(a.p), L.ptr(a.g), L.ptr(self.buf), L.ptr(a.hi), L.ptr(a.lo), ctypes.c_long(a.numel),
                                          L.ptr(self.hyper), L.ptr(self.sgd_steps), L.current_stream()), 'sgd')
        for l in self._linears:
            l.refresh()

    def train_step(self, x, target, starts):
        B = x.shape[0]
        self.forward(x, starts)
        loss = self.cross_entropy(B, target)
        self.backward(B)
        self.sgd_step()
        return loss

    def prepare_geometry(self, x, starts, slot=0):
        """FPS / kNN / transposed lists / 3-NN tables of batch x into geometry set `slot`, joined into the current stream."""
        ws = self.workspace(x.shape[0])
        if ws.geo[slot] is None:
            ws.geo[slot] = self._new_geometry(ws, x.shape[0])
        g = ws.geo[slot]
        self._geometry(ws, x.shape[0], x, starts, g)
        if g.events[-1] is not None:
            torch.cuda.current_stream().wait_event(g.events[-1])
        return g

    def train_step_pipelined(self, x, target, starts, next_x, next_starts, slot):
        """One training step on batch x, whose geometry was prepared into set `slot` by the previous call (or prepare_geometry),
        while the side stream prepares the geometry of the NEXT batch into the other set.  The geometry of a batch depends on
        its coordinates (and the FPS start draws) only, and FPS in particular is a long chain of sequential iterations on one
        workgroup per cloud: computed inside the step it is 1.5 ms of mostly idle GPU before the first TransitionDown can
        start (cfg-5); computed one step ahead it hides behind the GEMMs of the current step.  Every step still does one
        geometry pass and one forward / backward / update; the two only belong to consecutive batches."""
        B = x.shape[0]
        ws = self.workspace(B)
        cur = ws.geo[slot]
        assert cur is not None, 'prepare_geometry(x, starts, slot) must run before the first pipelined step'
        if ws.geo[1 - slot] is None:
            ws.geo[1 - slot] = self._new_geometry(ws, B)
        nxt = ws.geo[1 - slot]
        self._geometry(ws, B, next_x, next_starts, nxt)          # side stream (forks from here)
        self.forward(x, starts, geometry=cur)
        loss = self.cross_entropy(B, target)
        self.backward(B)
        self.sgd_step()
        if nxt.events[-1] is not None:
            torch.cuda.current_stream().wait_event(nxt.events[-1])    # join: the next step finds its geometry complete
        return loss

    def capture_train_step_pipelined(self, xs, ys, starts):
        """Two HIP graphs over static buffers xs[p], ys[p], starts[p] (p = 0, 1): graph p trains on batch p with geometry set p and
        prepares set 1-p from batch 1-p.  Protocol: fill buffers 0, prepare_geometry(xs[0], starts[0], 0); then for every step
        write the NEXT batch into buffers 1-p, graphs[p].replay(), p ^= 1.  Returns (graphs, loss scalar tensor)."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with self._preserved_state():
            with torch.cuda.stream(side):            # warm-up: kernel attributes, workspaces, both geometry sets
                self.prepare_geometry(xs[0], starts[0], 0)
                for p in (0, 1):
                    self.train_step_pipelined(xs[p], ys[p], starts[p], xs[1 - p], starts[1 - p], p)
            torch.cuda.current_stream().wait_stream(side)
        graphs = []
        if TWO_CHAINS and self.levels > 0:
            B = xs[0].shape[0]
            ws = self.workspace(B)
            if getattr(self, '_geo_stream', None) is None:
                self._geo_stream = torch.cuda.Stream(device=self.device)
            for p in (0, 1):
                gg = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gg):           # one chain: FPS -> kNN -> lists per level, 3-NN tables (set 1-p from buffers 1-p)
                    self._geometry(ws, B, xs[1 - p], starts[1 - p], ws.geo[1 - p], serial=True)
                ws.geo[p].events = [None] * (self.levels + 1)      # the step graph orders itself behind the geometry graph as a whole
                sg = torch.cuda.CUDAGraph()
                with torch.cuda.graph(sg):           # one chain: forward / loss / backward / update on batch p with geometry set p
                    self.forward(xs[p], starts[p], geometry=ws.geo[p])
                    loss = self.cross_entropy(B, ys[p])
                    self.backward(B)
                    self.sgd_step()
                graphs.append(_TwoChainStep(gg, sg, self._geo_stream))
            self.prepare_geometry(xs[0], starts[0], 0)
            torch.cuda.current_stream().wait_stream(self._geo_stream)
            return graphs, loss
        for p in (0, 1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                loss = self.train_step_pipelined(xs[p], ys[p], starts[p], xs[1 - p], starts[1 - p], p)
            graphs.append(g)
        self.prepare_geometry(xs[0], starts[0], 0)
        return graphs, loss

    def lwf_train_step(self, x, target, starts, img, img_target, lambda_weight=0.1):
        """train_partseg_lwf.py:207-228: loss = CE(seg_pred, target) + lambda * CE(forward_images(images), label_teacher); one
        backward (here: the image backward accumulates into the same gradient arena), one SGD step.
        Returns (point loss, image loss) device scalars."""
        if self.images is None:
            raise RuntimeError(f'{self.variant} has no forward_images (models/3DViT/model.py replaces patch_embed by PointEmbed)')
        B = x.shape[0]
        self.forward(x, starts)
        loss = self.cross_entropy(B, target)
        self.backward(B)
        Bi = img.shape[0]
        self.images.forward(img)
        loss_i = self.images.cross_entropy(Bi, img_target, grad_scale=lambda_weight)
        self.images.backward(Bi)
        self.sgd_step()
        return loss, loss_i

    def capture_train_step(self, x, target, starts):
        """Captures train_step over the given (static) input buffers into a HIP graph -- the step enqueues ~400 kernels and no
        host synchronisation, so replaying it removes the Python / launch pacing.  Copy new batches (and FPS start indices)
        into x / target / starts, then graph.replay().  Returns (graph, loss scalar tensor)."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with self._preserved_state():
            with torch.cuda.stream(side):            # warm-up on a side stream: kernel attributes, workspaces
                self.train_step(x, target, starts)
            torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss = self.train_step(x, target, starts)
        return graph, loss
