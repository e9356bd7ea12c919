"""VoxelEngine: host-side orchestration of the Simple3D-Former voxel training step on one MI355X.

Owns a flat fp32 parameter arena (+ gradient / Adam-moment arenas and split-bf16 weight planes of the same layout),
pre-allocated activation workspaces, and enqueues the whole forward / backward / optimizer step through the C ABI
of libs3d_hip.so on torch's current HIP stream.  PyTorch is used for device memory, streams and (in parallel.py)
torch.distributed only -- all arithmetic of the path runs in the hand-written gfx950 kernels.

Reference call sites this replaces: Feature3D_ViT2D_V2.forward_features/forward
(models/vit_3d_2d_pretrain.py:453-526), F.cross_entropy + loss.backward() + optimizer.step()
(train_cls_voxel.py:277-288)."""
import contextlib
import ctypes
import os
import math

import numpy as np
import torch

from . import _lib as L

BACKBONES = {  # models/vit_3d_2d_pretrain.py:279-325 -- deit_base is built with num_heads=3 (reference quirk)
    'deit_tiny_patch16_224': dict(embed_dim=192, depth=12, num_heads=3),
    'deit_small_patch16_224': dict(embed_dim=384, depth=12, num_heads=6),
    'deit_base_patch16_224': dict(embed_dim=768, depth=12, num_heads=3),
    'deit_base_distilled_patch16_224': dict(embed_dim=768, depth=12, num_heads=3),
    'vit_base_patch16_224_21k': dict(embed_dim=768, depth=12, num_heads=3),
}
LN_EPS = 1e-6
FOLD_MODE = {'VoxelEmbed': 0, 'VoxelNaiveProjection': 1, 'VoxelEmbed_no_average': 2}
# Arena order inside a block: both LayerNorms first, then the GEMM parameters as ONE contiguous range -- the range whose optimizer update
# rides on the next block's backward launches (S3dAdamFill); the LayerNorm gradients are only final after the batched partial-sum
# reduction at the end of s3d_blocks_bwd.
BLOCK_PARAM_ORDER = ['norm1.weight', 'norm1.bias', 'norm2.weight', 'norm2.bias', 'attn.qkv.weight', 'attn.qkv.bias', 'attn.proj.weight',
                     'attn.proj.bias', 'mlp.fc1.weight', 'mlp.fc1.bias', 'mlp.fc2.weight', 'mlp.fc2.bias']
BLOCK_PARAM_SHAPE = {'norm1.weight': lambda D: (D,), 'norm1.bias': lambda D: (D,), 'norm2.weight': lambda D: (D,), 'norm2.bias': lambda D: (D,),
                     'attn.qkv.weight': lambda D: (3 * D, D), 'attn.qkv.bias': lambda D: (3 * D,), 'attn.proj.weight': lambda D: (D, D),
                     'attn.proj.bias': lambda D: (D,), 'mlp.fc1.weight': lambda D: (4 * D, D), 'mlp.fc1.bias': lambda D: (4 * D,),
                     'mlp.fc2.weight': lambda D: (D, 4 * D), 'mlp.fc2.bias': lambda D: (D,)}


BUCKET_ALIGN = 512


def _round_up(x, m):
    return (x + m - 1) // m * m


def voxel_param_shapes(*, backbone, embed_layer, cell, patch, n_classes, pos_embedding='default', head='default',
                       image_branch=False):
    """Ordered {state_dict key: shape} of the parameters the voxel forward touches (forward order, so that
    gradient buckets complete back-to-front during backward)."""
    cfg = BACKBONES[backbone]
    D, depth = cfg['embed_dim'], cfg['depth']
    c = cell
    shapes = {}
    if image_branch:                         # 2-D stem of forward_images (vit_3d_2d_pretrain.py:435-451)
        from .image_branch import image_param_shapes
        shapes.update(image_param_shapes(D)[0])
    if embed_layer == 'VoxelNaiveProjection':
        shapes['voxel_embed.proj.conv2d_1.weight'] = (D, 1, c, c)
        shapes['voxel_embed.proj.conv2d_1.bias'] = (D,)
        ntok = patch ** 2 + 1
    else:
        shapes['voxel_embed.proj.conv3d_1.weight'] = (D, 1, c, c, c)
        shapes['voxel_embed.proj.conv3d_1.bias'] = (D,)
        ntok = (patch ** 2 if embed_layer == 'VoxelEmbed' else patch ** 3) + 1
    if pos_embedding == 'group_embed':      # vit_3d_2d_pretrain.py:377-383 (forward order: tokens -> encoder layer -> blocks)
        if embed_layer != 'VoxelEmbed_no_average':
            raise ValueError('group_embed needs the 5-D VoxelEmbed_no_average tokenizer (vit_3d_2d_pretrain.py:473-474)')
        shapes['group_cls_token'] = (1, 1, D)
        shapes['group_pos_embed'] = (1, patch + 1, D)
        g = 'group_embed.'
        shapes[g + 'self_attn.in_proj_weight'] = (3 * D, D); shapes[g + 'self_attn.in_proj_bias'] = (3 * D,)
        shapes[g + 'self_attn.out_proj.weight'] = (D, D); shapes[g + 'self_attn.out_proj.bias'] = (D,)
        shapes[g + 'linear1.weight'] = (D, D); shapes[g + 'linear1.bias'] = (D,)
        shapes[g + 'linear2.weight'] = (D, D); shapes[g + 'linear2.bias'] = (D,)
        shapes[g + 'norm1.weight'] = (D,); shapes[g + 'norm1.bias'] = (D,)
        shapes[g + 'norm2.weight'] = (D,); shapes[g + 'norm2.bias'] = (D,)
        ntok = patch ** 2 + 1
    shapes['cls_token'] = (1, 1, D)
    shapes['voxel_pos_embed'] = (1, ntok, D)
    for i in range(depth):
        for k in BLOCK_PARAM_ORDER:
            shapes[f'blocks.{i}.{k}'] = BLOCK_PARAM_SHAPE[k](D)
    shapes['norm.weight'] = (D,); shapes['norm.bias'] = (D,)
    if head == 'AMSoftmax':
        shapes['voxel_head.W'] = (D, n_classes)
    else:
        shapes['voxel_head.weight'] = (n_classes, D)
        shapes['voxel_head.bias'] = (n_classes,)
    if image_branch:
        from .image_branch import image_param_shapes
        shapes.update(image_param_shapes(D)[1])
    return shapes


class ParamArena:
    """Flat fp32 arena + same-layout gradient / moment arenas + split-bf16 (hi, lo) planes."""

    def __init__(self, shapes, device):
        self.shapes = dict(shapes)
        self.offsets = {}
        off = 0
        for k, shp in self.shapes.items():
            if k.startswith('blocks.') and k.endswith('.norm1.weight'):
                # a block starts on a BUCKET_ALIGN boundary: data-parallel gradient buckets are arena slices that begin at block starts,
                # and the sharded optimizer (parallel.ShardedDataParallelTrainer) cuts every bucket into world-size equal, 8-element
                # aligned shards (reduce_scatter_tensor / all_gather_into_tensor need equal chunks) -- 512 = 8 x 64 ranks.  The pad
                # elements are ordinary zeros with zero gradients: Adam leaves them at zero.
                off = _round_up(off, BUCKET_ALIGN)
            self.offsets[k] = off
            off = _round_up(off + int(np.prod(shp)), 8)       # 16-byte aligned bf16 planes, 32-byte aligned fp32
        self.numel = _round_up(off, BUCKET_ALIGN)
        self.device = device
        self.p = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.g = torch.zeros_like(self.p)
        self.m = torch.zeros_like(self.p)
        self.v = torch.zeros_like(self.p)
        self.hi = torch.zeros(self.numel, dtype=torch.bfloat16, device=device)
        self.lo = torch.zeros_like(self.hi)

    def _view(self, flat, k):
        o, shp = self.offsets[k], self.shapes[k]
        return flat[o:o + int(np.prod(shp))].view(shp)

    def param(self, k): return self._view(self.p, k)
    def grad(self, k): return self._view(self.g, k)
    def hi_of(self, k): return self._view(self.hi, k)
    def lo_of(self, k): return self._view(self.lo, k)

    def load(self, sd):
        with torch.no_grad():
            for k in self.shapes:
                self.param(k).copy_(sd[k].to(self.device, torch.float32).reshape(self.shapes[k]))

    def state_dict(self):
        return {k: self.param(k).detach().clone() for k in self.shapes}

    def refresh_planes_range(self, start, end):
        """hi / lo planes of the arena slice [start, end) from the fp32 parameters (the sharded optimizer's all-gathered buckets)."""
        off = lambda t, b: ctypes.c_void_p(t.data_ptr() + b * start)
        L.check(L.lib().s3d_split_bf16(off(self.p, 4), off(self.hi, 2), off(self.lo, 2), ctypes.c_long(1), ctypes.c_long(end - start),
                                       ctypes.c_long(end - start), L.current_stream()), 'split_bf16 (range)')

    def refresh_planes(self):
        """fp32 arena -> split-bf16 planes (needed whenever the parameters were changed outside s3d_adam_step)."""
        lib = L.lib()
        L.check(lib.s3d_split_bf16(L.ptr(self.p), L.ptr(self.hi), L.ptr(self.lo), ctypes.c_long(1),
                                   ctypes.c_long(self.numel), ctypes.c_long(self.numel), L.current_stream()), 'split_bf16')


class _BlockWorkspace:
    """Saved activations of `depth` consecutive blocks on M = Bb*N rows, plus the ctypes act table."""

    def __init__(self, depth, Bb, N, D, H, hidden, device, split, ln_fuse=None, cls_only=False, fuse=None, precise=False,
                 shared_lo=None):
        """shared_lo (default: unless precise): the low planes of xn1 / qkv / xn2 / hact feed the NEXT forward launch's split product
        and nothing else (the backward reads the high planes; att_lo stays per block, the attention backward's delta reads it), so
        every block writes them to the same buffers instead of to `depth` of them -- 11.5 MB per cfg-2 block that would otherwise travel through L2 and the Infinity Cache to HBM once,
        pushing out the saved activations the backward is about to read (the batch-64 step is cache-resident: DESIGN section 6)."""
        M = Bb * N
        shared_lo = (not precise and os.environ.get('S3D_SHARED_LO', '1') != '0') if shared_lo is None else bool(shared_lo)
        if precise and shared_lo:
            # the split-precision backward reads every block's OWN xn1_lo / qkv_lo / xn2_lo / hact_lo (capi.hip block_bwd_split checks the
            # pointers, it cannot check whose data they hold)
            raise ValueError('_BlockWorkspace(precise=True) needs per-block low planes: shared_lo must be False')
        self.shared_lo = shared_lo
        nlo = 1 if shared_lo else depth
        ln_fuse = LN_FUSE if ln_fuse is None else bool(ln_fuse)
        fuse = FUSED_BLOCKS if fuse is None else bool(fuse)
        self.cls_only = bool(cls_only)
        f32 = dict(dtype=torch.float32, device=device)
        b16 = dict(dtype=torch.bfloat16, device=device)
        self.depth, self.Bb, self.N, self.M = depth, Bb, N, M
        self.x = [torch.empty(M, D, **f32) for _ in range(depth + 1)]
        self.x_mid = [torch.empty(M, D, **f32) for _ in range(depth)]
        self.stats = torch.empty(depth, 4, M, **f32)
        self.lse = torch.empty(depth, Bb * H * N, **f32)
        self.xn1, self.xn1_lo = torch.empty(depth, M, D, **b16), torch.empty(nlo, M, D, **b16)
        self.qkv, self.qkv_lo = torch.empty(depth, M, 3 * D, **b16), torch.empty(nlo, M, 3 * D, **b16)
        self.att, self.att_lo = torch.empty(depth, M, D, **b16), torch.empty(depth, M, D, **b16)       # delta = sum(dO * (O_hi + O_lo)) reads att_lo
        self.xn2, self.xn2_lo = torch.empty(depth, M, D, **b16), torch.empty(nlo, M, D, **b16)
        self.hpre = torch.empty(depth, M, hidden, **b16)
        self.hact, self.hact_lo = torch.empty(depth, M, hidden, **b16), torch.empty(nlo, M, hidden, **b16)
        self.hpre_lo = torch.empty(depth, M, hidden, **b16) if precise else None      # split-precision backward: hpre to 16 bits
        self.acts = (L.S3dBlockActs * depth)()
        for i in range(depth):
            j = 0 if shared_lo else i
            L.fill(self.acts[i], x_in=self.x[i], x_mid=self.x_mid[i], x_out=self.x[i + 1],
                   mean1=self.stats[i, 0], rstd1=self.stats[i, 1], mean2=self.stats[i, 2], rstd2=self.stats[i, 3],
                   lse=self.lse[i], xn1_hi=self.xn1[i], xn1_lo=self.xn1_lo[j], qkv_hi=self.qkv[i],
                   qkv_lo=self.qkv_lo[j], att_hi=self.att[i], att_lo=self.att_lo[i], xn2_hi=self.xn2[i],
                   xn2_lo=self.xn2_lo[j], hpre=self.hpre[i], hact_hi=self.hact[i], hact_lo=self.hact_lo[j],
                   hpre_lo=self.hpre_lo[i] if precise else None)
        # one ticket per 32-row band: the LayerNorms that follow attn.proj / mlp.fc2 run inside those GEMM launches (left zero)
        self.ln_tickets = torch.zeros((M + 31) // 32 + 8, dtype=torch.int32, device=device)
        self.shape = L.S3dBlockShape(Bb=Bb, N=N, D=D, H=H, hidden=hidden, eps=LN_EPS, split=1 if split else 0,
                                     ln_tickets=self.ln_tickets.data_ptr() if ln_fuse else None,
                                     cls_only_block=depth if self.cls_only else 0, fuse=(0 if FUSED_BWD else 1) if fuse else -1)


# LayerNorm forward inside the producing GEMM launch (gemm.hip ln_band_tail: the last-arriving tile of a row band normalises it).
# Built, parity-green and measured on cfg-2: 2.18 - 2.21 ms/step vs 1.96 - 1.99 with the stand-alone LayerNorm kernels -- the
# in-launch hand-off (write-through stores, store acknowledgement, agent-scope ticket, L1-bypassing re-read: four dependent trips to
# the memory side of L2) costs more than the ~6.5 us kernel boundary + LayerNorm launch it removes.  Opt-in: S3D_LN_FUSE=1.
LN_FUSE = os.environ.get('S3D_LN_FUSE', '0') == '1'
# norm1 + qkv + attention and norm2 + fc1 + GELU as one launch each (csrc/fused_block.hip; S3dBlockShape::fuse) wherever the shape
# qualifies (small token counts: cfg-1 / cfg-2).  S3D_FUSED_BLOCKS=0: the seven-launch block forward everywhere (A/B measurements).
FUSED_BLOCKS = os.environ.get('S3D_FUSED_BLOCKS', '1') != '0'
# ... and in the backward attn.proj's dgrad inside the attention-backward launch, its wgrad on the qkv pair launch (blk_attn_bwd_kernel;
# S3dBlockShape::fuse = 0).  S3D_FUSED_BWD=0: fused forward only (fuse = 1), for A/B measurements.
FUSED_BWD = os.environ.get('S3D_FUSED_BWD', '1') != '0'
# The last block's output is consumed at the class-token rows only (norm(x)[:, 0]): its row-local tail (proj, norm2, mlp) and their
# backward run on those rows alone (S3dBlockShape::cls_only_block).  S3D_CLS_ONLY=0: dense, as the reference computes it.
CLS_ONLY = os.environ.get('S3D_CLS_ONLY', '1') != '0'
# Adam beside the backward (VoxelEngine.update_slices): the arena is updated in this many slices, each on a second stream as soon
# as the backward segment that finishes its gradients has been enqueued (train_step only; 0 = one update after the whole backward).
# Measured and NOT the default: every node of a captured graph that lies in a region with two live
# branches costs ~2.8 us more on this runtime (cfg-2: 1.85 -> 1.96 - 2.25 ms; the same with only ~25 us of small tail launches on the
# second branch: 1.75 -> 2.17 ms), far more than the update's 0.13 ms.  A captured step has to stay ONE chain.
UPDATE_OVERLAP = int(os.environ.get('S3D_UPDATE_OVERLAP', '0'))
UPDATE_WORKGROUPS = int(os.environ.get('S3D_UPDATE_WORKGROUPS', '0'))
# optimizer.step() as filler workgroups inside the backward launches (S3dAdamFill, csrc/adam_fill.h): built, bitwise equal to the single
# update, measured 0.6 - 1.4 % SLOWER at cfg-2 (each launch slows down by what its share costs as a stream of its own:
# profiles/r04_adam_fill.txt) -> opt-in (S3D_ADAM_FILL=1 / VoxelEngine.adam_fill = True).
ADAM_FILL = os.environ.get('S3D_ADAM_FILL', '0') == '1'
# Round 5: the small-batch backward as a dgrad chain + grouped wgrads (capi.hip: block_bwd_chain; S3dBlockScratch::wg_ring).  WGRAD_GROUP =
# blocks per grouped wgrad launch (= ring slots; 0: the paired dgrad + wgrad launches of rounds 1 - 4), DGRAD_SPLITK = k-slices of the
# fc1 / qkv dgrads (their planes are added by the LayerNorm backward).
WGRAD_GROUP = int(os.environ.get('S3D_WGRAD_GROUP', '4'))
DGRAD_SPLITK = int(os.environ.get('S3D_DGRAD_SPLITK', '3'))
LN_BWD_FUSE = os.environ.get('S3D_LN_BWD_FUSE', '1') != '0'             # LayerNorm backward as the epilogue of the fc1 / qkv dgrads (row statistics)
WGRAD_OVERWRITE = os.environ.get('S3D_WGRAD_OVERWRITE', '1') != '0'     # train_step: grouped wgrads store instead of read-modify-write
# stored dropout mask of the group encoder layer's attention (S3dEncActs::attn_mask): one bit per weight, quadratic in the batch
ATTN_MASK_BUDGET_BYTES = int(float(os.environ.get('S3D_ATTN_MASK_BUDGET_GB', '8')) * 2 ** 30)
FUSE_LOSS_END = os.environ.get('S3D_FUSE_LOSS_END', '1') != '0'     # final norm + head + CE + their backward in two launches
LN_PARTIAL_BLOCKS = int(os.environ.get('S3D_LN_PARTIAL_BLOCKS', '-1'))    # 0: LayerNorm backward uses atomics; -1: by row count


def ln_partial_blocks(rows, D=None):
    """Workgroups (= rows of column-sum partials) of a LayerNorm backward over `rows` rows: 208 at cfg-2's 1664 rows (two rows per
    wave); 416 for the 16 k - 190 k-row passes of the point path / cfg-3, where 208 four-wave workgroups leave every SIMD with one
    wave and the narrow (D = 192) rows with too few bytes in flight (cfg-4: 16.54 -> 16.32 ms; cfg-3 / cfg-5 unchanged)."""
    if LN_PARTIAL_BLOCKS >= 0:
        return LN_PARTIAL_BLOCKS
    if D == 192 and rows > 8192:
        # the 192-wide dgrads of the point path carry their LayerNorm backward as an epilogue on 64-row tiles (bwd_gemm.hip:
        # dgrad_lnrows_kernel): one row of partials per tile
        return max(416, (rows + 63) // 64)
    return 208 if rows <= 8192 else 416


class _BlockScratch:
    def __init__(self, M, D, H, hidden, BHN, device, depth=0, precise=False, wgrad_ring=None, ln_bwd_fuse=True):
        """wgrad_ring = (slots, dgrad_splitk, slot_bytes): the dy ring of the dgrad chain (S3dBlockScratch::wg_ring, round 5) -- every
        block keeps d(x_out) / d(x_mid) / dh / dqkv in a slot of its own and the wgrads of `slots` blocks run as one grouped launch."""
        f32 = dict(dtype=torch.float32, device=device)
        b16 = dict(dtype=torch.bfloat16, device=device)
        planes = wgrad_ring[1] if wgrad_ring else 1
        self._dxn_planes = torch.empty(planes, M, D, **f32)          # k-slices of the split-K dgrads (added by the LayerNorm backward)
        self.dxn = self._dxn_planes[0]
        # dx_a and its bf16 copy share one allocation: the backward starts from "zero except the cls rows", one fill instead of two
        self._dxa_raw = torch.empty(M * D * (8 if precise else 6), dtype=torch.uint8, device=device)
        self.dx_a = self._dxa_raw[:M * D * 4].view(torch.float32).view(M, D)
        self.dx_a_bf = self._dxa_raw[M * D * 4:M * D * 6].view(torch.bfloat16).view(M, D)
        self.dx_b = torch.empty(M, D, **f32)
        self.dx_b_bf = torch.empty(M, D, **b16)
        self.dh = torch.empty(M, hidden, **b16)
        self.dqkv = torch.empty(M, 3 * D, **b16)
        self.datt = torch.empty(M, D, **b16)
        self.delta = torch.empty(BHN, **f32)
        self.zero_dx_a = self._dxa_raw.zero_
        self.c = L.S3dBlockScratch()
        L.fill(self.c, dxn=self.dxn, dx_a=self.dx_a, dx_b=self.dx_b, dx_a_bf=self.dx_a_bf, dx_b_bf=self.dx_b_bf,
               dh=self.dh, dqkv=self.dqkv, datt=self.datt, delta=self.delta)
        if precise:            # split-precision backward (S3dBlockScratch::dx_a_lo ...): a lo plane behind every bf16 gradient buffer
            self.dx_a_lo = self._dxa_raw[M * D * 6:].view(torch.bfloat16).view(M, D)
            self.dx_b_lo = torch.empty(M, D, **b16); self.dh_lo = torch.empty(M, hidden, **b16)
            self.dqkv_lo = torch.empty(M, 3 * D, **b16); self.datt_lo = torch.empty(M, D, **b16)
            L.fill(self.c, dx_a_lo=self.dx_a_lo, dx_b_lo=self.dx_b_lo, dh_lo=self.dh_lo, dqkv_lo=self.dqkv_lo, datt_lo=self.datt_lo)
        self.M, self.D = M, D
        self.wg_ring = None
        self.ln_aux = None
        if wgrad_ring and not precise:
            slots, splitk, slot_bytes = wgrad_ring
            self.wg_ring = torch.empty(slots * slot_bytes, dtype=torch.uint8, device=device)
            L.fill(self.c, wg_ring=self.wg_ring, wg_slots=slots, dgrad_splitk=splitk)
            if LN_BWD_FUSE and ln_bwd_fuse and depth > 0:
                # the LayerNorm backward launches folded into the fc1 / qkv dgrads (S3dBlockScratch::ln_aux / ln_rowstat)
                self.ln_aux = torch.zeros(depth * 2 * (hidden + 3 * D), **f32)
                self.ln_rowstat = torch.zeros(4 * M, **f32)
                L.fill(self.c, ln_aux=self.ln_aux, ln_rowstat=self.ln_rowstat)
        nblk = ln_partial_blocks(M, D)
        if depth > 0 and nblk > 0:
            # column-sum partials of the 2*depth LayerNorms of one s3d_blocks_bwd call (S3dBlockScratch::ln_partial)
            self.ln_partial = torch.empty(2 * depth, nblk, 2, D, **f32)
            L.fill(self.c, ln_partial=self.ln_partial, ln_partial_blocks=nblk)


def _cls_scratch(base, rows, D, device, precise=False):
    """A copy of the S3dBlockScratch table `base` with zero-initialised class-row gradient buffers for a block pass of `rows` rows
    (S3dBlockShape::cls_only_block): only the class rows are ever written, the rest stays zero."""
    c = L.S3dBlockScratch()
    ctypes.memmove(ctypes.byref(c), ctypes.byref(base), ctypes.sizeof(c))
    bufs = (torch.zeros(rows, D, dtype=torch.float32, device=device), torch.zeros(rows, D, dtype=torch.bfloat16, device=device),
            torch.zeros(rows, D, dtype=torch.bfloat16, device=device))
    L.fill(c, dx_b_cls=bufs[0], dx_b_bf_cls=bufs[1], datt_cls=bufs[2])
    if precise:
        bufs += (torch.zeros(rows, D, dtype=torch.bfloat16, device=device), torch.zeros(rows, D, dtype=torch.bfloat16, device=device))
        L.fill(c, dx_b_lo_cls=bufs[3], datt_lo_cls=bufs[4])
    return c, bufs


class VoxelEngine:
    """deit_* backbone + VoxelEmbed / VoxelNaiveProjection / VoxelEmbed_no_average tokenizer with the `default`
    positional embedding (vit_3d_2d_pretrain.py:455-470) and Linear / AM-softmax head."""

    def __init__(self, *, backbone, embed_layer, voxel_size, cell, patch, n_classes, pos_embedding='default',
                 head='default', device='cuda', split=True, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, image_branch=False, ln_fuse=None,
                 precise_backward=False, backward=None):
        if backward not in (None, 'bf16', 'split'):
            raise ValueError(f"backward must be 'bf16' (plain bf16 MFMA operands: the default, what bench.py times) or 'split' "
                             f"(every gradient product on hi + lo operands: precise_backward), not {backward!r}")
        precise_backward = bool(precise_backward) or backward == 'split'
        if backbone not in BACKBONES:
            raise ValueError("Unknown transformer backbone name!")           # vit_3d_2d_pretrain.py:393-394
        if pos_embedding not in (None, 'default', 'group_embed'):
            raise ValueError("Unknown positional embedding scheme!")         # vit_3d_2d_pretrain.py:389
        if embed_layer not in FOLD_MODE:
            raise ValueError(f'unknown embed layer {embed_layer!r}')         # train_cls_voxel.py:138-142
        self.lib = L.lib()                                                    # raises if the HIP library is missing
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('VoxelEngine runs on an MI355X (cuda/HIP device) only; the CPU reference lives in oracle/')
        cfg = BACKBONES[backbone]
        self.group = pos_embedding == 'group_embed'
        self.cfg = dict(backbone=backbone, embed_layer=embed_layer, voxel_size=voxel_size, cell=cell, patch=patch,
                        n_classes=n_classes, pos_embedding='group_embed' if self.group else 'default', head=head)
        self.D, self.depth, self.H = cfg['embed_dim'], cfg['depth'], cfg['num_heads']
        self.hidden = 4 * self.D
        self.V, self.c, self.P, self.C = voxel_size, cell, patch, n_classes
        assert (voxel_size - cell) // cell + 1 == patch, 'patch_size must equal floor((V-c)/c)+1'
        self.embed_layer = embed_layer
        self.fold_mode = FOLD_MODE[embed_layer]
        self.ntok = (patch ** 3 if embed_layer == 'VoxelEmbed_no_average' else patch ** 2) + 1
        if self.group:                      # pass 1: B*P*P groups of P+1 tokens; pass 2: B sequences of P*P+1 tokens
            self.fold_mode = 3
            self.ntok = patch + 1
            self.ntok2 = patch ** 2 + 1
            self.enc_heads = 4              # nn.TransformerEncoderLayer(nhead=4), vit_3d_2d_pretrain.py:381
        self.Kc = cell ** 2 if embed_layer == 'VoxelNaiveProjection' else cell ** 3
        self.Kpad = _round_up(self.Kc, 8)
        self.am = head == 'AMSoftmax'
        self.split = bool(split)
        # Parity mode: the backward (train_cls_voxel.py:287) in split precision -- every dgrad / wgrad a three-MFMA product on hi + lo
        # operands without split-K, fp32 attention backward, gradients carried as hi + lo pairs -- so that gradients can be held to
        # ~1e-4 of the reference instead of the plain-bf16 noise floor.  Several times slower.  Tests only.
        self.precise = bool(precise_backward)
        self.backward_precision = 'split' if self.precise else 'bf16'
        if self.precise and (image_branch or not split):
            raise NotImplementedError('precise_backward covers the split-bf16 voxel path (no image branch)')
        self.ln_fuse = ln_fuse              # None: S3D_LN_FUSE decides (default off, see LN_FUSE above)
        self.conv_key = 'voxel_embed.proj.conv2d_1' if embed_layer == 'VoxelNaiveProjection' else 'voxel_embed.proj.conv3d_1'
        self.shapes = voxel_param_shapes(backbone=backbone, embed_layer=embed_layer, cell=cell, patch=patch,
                                         n_classes=n_classes, head=head, pos_embedding=self.cfg['pos_embedding'],
                                         image_branch=image_branch)
        self.arena = ParamArena(self.shapes, self.device)
        self._build_param_tables()
        self.images = None
        if image_branch:                    # forward_images / LwF (vit_3d_2d_pretrain.py:435-451, train_cls_voxel.py:250-267)
            from .image_branch import ImageBranch
            self.images = ImageBranch(self)
        if self.Kpad != self.Kc:   # padded conv-weight planes (row pitch Kpad) refreshed from the arena each step
            self.conv_hi = torch.zeros(self.D, self.Kpad, dtype=torch.bfloat16, device=self.device)
            self.conv_lo = torch.zeros_like(self.conv_hi)
            self.conv_gpad = torch.zeros(self.D, self.Kpad, dtype=torch.float32, device=self.device)
        self.adam_state = torch.zeros(9, dtype=torch.int32, device=self.device)
        self.set_optimizer(lr=lr, betas=betas, eps=eps)
        self._ws = {}
        self._graphs = {}
        self.grads_owned = False            # True while a fused step that owns the gradient arena runs its backward (blocks_backward_range)
        self.capture_epoch = 0              # bumped whenever something baked into captured graphs changes (set_dropout)
        self.world_size = 1
        # group_embed encoder-layer dropout: 0 = eval mode; set_dropout(0.1) = the reference's training mode (hash-based masks)
        self.dropout_p = 0.0
        self.dropout_seed = torch.zeros(1, dtype=torch.int64, device=self.device)

    # ------------------------------------------------------------------ parameters
    def _build_param_tables(self):
        a = self.arena
        self.bparams = (L.S3dBlockParams * self.depth)()
        self.bgrads = (L.S3dBlockGrads * self.depth)()
        for i in range(self.depth):
            p = f'blocks.{i}.'
            L.fill(self.bparams[i], ln1_w=a.param(p + 'norm1.weight'), ln1_b=a.param(p + 'norm1.bias'),
                   ln2_w=a.param(p + 'norm2.weight'), ln2_b=a.param(p + 'norm2.bias'),
                   qkv_b=a.param(p + 'attn.qkv.bias'), proj_b=a.param(p + 'attn.proj.bias'),
                   fc1_b=a.param(p + 'mlp.fc1.bias'), fc2_b=a.param(p + 'mlp.fc2.bias'),
                   qkv_w_hi=a.hi_of(p + 'attn.qkv.weight'), qkv_w_lo=a.lo_of(p + 'attn.qkv.weight'),
                   proj_w_hi=a.hi_of(p + 'attn.proj.weight'), proj_w_lo=a.lo_of(p + 'attn.proj.weight'),
                   fc1_w_hi=a.hi_of(p + 'mlp.fc1.weight'), fc1_w_lo=a.lo_of(p + 'mlp.fc1.weight'),
                   fc2_w_hi=a.hi_of(p + 'mlp.fc2.weight'), fc2_w_lo=a.lo_of(p + 'mlp.fc2.weight'))
            L.fill(self.bgrads[i], ln1_w=a.grad(p + 'norm1.weight'), ln1_b=a.grad(p + 'norm1.bias'),
                   ln2_w=a.grad(p + 'norm2.weight'), ln2_b=a.grad(p + 'norm2.bias'),
                   qkv_w=a.grad(p + 'attn.qkv.weight'), qkv_b=a.grad(p + 'attn.qkv.bias'),
                   proj_w=a.grad(p + 'attn.proj.weight'), proj_b=a.grad(p + 'attn.proj.bias'),
                   fc1_w=a.grad(p + 'mlp.fc1.weight'), fc1_b=a.grad(p + 'mlp.fc1.bias'),
                   fc2_w=a.grad(p + 'mlp.fc2.weight'), fc2_b=a.grad(p + 'mlp.fc2.bias'))
        if self.group:
            g = 'group_embed.'
            self.eparams = L.fill(L.S3dEncParams(), in_b=a.param(g + 'self_attn.in_proj_bias'), out_b=a.param(g + 'self_attn.out_proj.bias'),
                                  l1_b=a.param(g + 'linear1.bias'), l2_b=a.param(g + 'linear2.bias'),
                                  n1_w=a.param(g + 'norm1.weight'), n1_b=a.param(g + 'norm1.bias'),
                                  n2_w=a.param(g + 'norm2.weight'), n2_b=a.param(g + 'norm2.bias'),
                                  in_w_hi=a.hi_of(g + 'self_attn.in_proj_weight'), in_w_lo=a.lo_of(g + 'self_attn.in_proj_weight'),
                                  out_w_hi=a.hi_of(g + 'self_attn.out_proj.weight'), out_w_lo=a.lo_of(g + 'self_attn.out_proj.weight'),
                                  l1_w_hi=a.hi_of(g + 'linear1.weight'), l1_w_lo=a.lo_of(g + 'linear1.weight'),
                                  l2_w_hi=a.hi_of(g + 'linear2.weight'), l2_w_lo=a.lo_of(g + 'linear2.weight'))
            self.egrads = L.fill(L.S3dEncGrads(), in_w=a.grad(g + 'self_attn.in_proj_weight'), in_b=a.grad(g + 'self_attn.in_proj_bias'),
                                 out_w=a.grad(g + 'self_attn.out_proj.weight'), out_b=a.grad(g + 'self_attn.out_proj.bias'),
                                 l1_w=a.grad(g + 'linear1.weight'), l1_b=a.grad(g + 'linear1.bias'),
                                 l2_w=a.grad(g + 'linear2.weight'), l2_b=a.grad(g + 'linear2.bias'),
                                 n1_w=a.grad(g + 'norm1.weight'), n1_b=a.grad(g + 'norm1.bias'),
                                 n2_w=a.grad(g + 'norm2.weight'), n2_b=a.grad(g + 'norm2.bias'))

    def load_state_dict(self, sd):
        self.arena.load(sd)
        self.refresh_weight_planes()

    def state_dict(self):
        return self.arena.state_dict()

    def refresh_weight_planes(self):
        self.arena.refresh_planes()
        self._refresh_conv_planes()

    def refresh_planes_range(self, start, end):
        """hi / lo planes of the arena slice [start, end) (an all-gathered bucket of the sharded data-parallel step)."""
        self.arena.refresh_planes_range(start, end)

    def _refresh_conv_planes(self):
        if self.Kpad != self.Kc:
            w = self.arena.param(self.conv_key + '.weight')
            L.check(self.lib.s3d_split_bf16(L.ptr(w), L.ptr(self.conv_hi), L.ptr(self.conv_lo), ctypes.c_long(self.D),
                                            ctypes.c_long(self.Kc), ctypes.c_long(self.Kpad), L.current_stream()), 'split conv')

    def optimizer_state(self):
        """-> dict(lr, betas, eps, grad_scale, step) currently on the device."""
        raw = self.adam_state.cpu().numpy()
        f = raw[:5].view(np.float32)
        return dict(lr=float(f[0]), betas=(float(f[1]), float(f[2])), eps=float(f[3]), grad_scale=float(f[4]), step=int(raw[7]))

    def set_optimizer(self, lr=None, betas=None, eps=None, grad_scale=None, step=None):
        """torch.optim.Adam hyper-parameters (train_cls_voxel.py:195); lives on the device so graph replays see it.  Arguments left
        at None keep their current value -- in particular a data-parallel trainer's grad_scale = 1/world survives a later
        set_optimizer(lr=...) (gradients stay the MEAN over ranks)."""
        if getattr(self, '_adam_init', False):
            cur = self.optimizer_state()
        else:
            cur = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, step=0)
            self._adam_init = True
        lr = cur['lr'] if lr is None else lr
        betas = cur['betas'] if betas is None else betas
        eps = cur['eps'] if eps is None else eps
        grad_scale = cur['grad_scale'] if grad_scale is None else grad_scale
        cur_step = cur['step'] if step is None else int(step)
        st = np.zeros(9, dtype=np.int32)
        st[:7] = np.array([lr, betas[0], betas[1], eps, grad_scale, 0.0, 0.0], dtype=np.float32).view(np.int32)
        st[7] = cur_step
        self.adam_state.copy_(torch.from_numpy(st))

    def set_dropout(self, p, seed=None):
        """Dropout of nn.TransformerEncoderLayer inside group_embed (vit_3d_2d_pretrain.py:381; p = 0.1 when model.train()).
        The seed lives on the device and is advanced once per training step (advance_dropout_seed).  The probability is a
        launch argument, i.e. baked into captured graphs: this engine's graph cache is keyed by it (toggling train / eval re-uses
        both captures) and capture_epoch is bumped so that trainers holding their own captures re-capture."""
        p = float(p)
        if p != self.dropout_p:
            self.capture_epoch += 1           # trainers holding their own captures re-capture; this engine's cache is keyed by p

        self.dropout_p = p
        if seed is not None:
            self.dropout_seed.fill_(int(seed))
        for ws in self._ws.values():
            if self.group:
                ws.enc.shape.dropout_p = self.dropout_p
                self._ensure_attn_mask(ws)

    def _ensure_attn_mask(self, ws):
        """S3dEncActs::attn_mask: one bit per attention weight of the seq-first encoder layer, written by the forward while it evaluates
        the dropout hash and read by the two backward kernels (long sequences only: the cooperative kernels, G >= 192)."""
        e = ws.enc
        t = (ws.G + 31) // 32
        words = self.ntok * self.enc_heads * t * t * 32 + 256      # grows with G^2: 3.5 GB at cfg-3 batch 64, 14 GB at batch 128
        want = (self.dropout_p > 0 and ws.G >= 192 and os.environ.get('S3D_NO_ATTN_MASK') != '1'       # (A/B knob)
                and words * 4 <= ATTN_MASK_BUDGET_BYTES)             # above the budget the kernels evaluate the hash (bit-identical, ~2 % slower)
        if not want:
            # eval mode / short sequences: nothing reads it.  The buffer itself is KEPT (train <-> eval toggles must neither free memory
            # that a captured training graph has baked into its kernel arguments nor re-zero gigabytes per toggle)
            e.acts.attn_mask = None
            return
        buf = getattr(self, '_attn_mask', None)                      # ONE buffer for every batch-size workspace, sized for the largest G seen
        if buf is None or buf.numel() < words:
            if buf is not None and self._graphs:
                # graphs captured so far hold the OLD pointer: it stays alive for as long as the engine does (a caller may still replay
                # its handle -- self-consistent on the old buffer), and the cache forgets them so that the next capture sees the new one
                self._attn_mask_retired = getattr(self, '_attn_mask_retired', []) + [buf]
                self._graphs.clear()
            buf = torch.zeros(words, dtype=torch.int32, device=self.device)
            self._attn_mask = buf
            self.capture_epoch += 1                                  # trainers holding their own captures re-capture
            for other in self._ws.values():
                if getattr(other, 'enc', None) is not None and other.enc.acts.attn_mask:
                    other.enc.acts.attn_mask = buf.data_ptr()
        e.acts.attn_mask = buf.data_ptr()

    def advance_dropout_seed(self):
        """Fresh dropout masks for the next forward: one device-side increment (graph-replay safe).  Every path that trains calls
        it once per step -- train_step, the data-parallel trainer's eager step and its captured phase 0."""
        if self.group and self.dropout_p > 0:
            self.dropout_seed.add_(1)

    def set_lr(self, lr):
        self.adam_state[0:1].copy_(torch.tensor([lr], dtype=torch.float32).view(torch.int32))

    # ------------------------------------------------------------------ workspaces
    def workspace(self, B):
        ws = self._ws.get(B)
        if ws is not None:
            return ws
        dev, D = self.device, self.D
        G = B * self.P * self.P if self.group else B          # sequences in the (first) block pass
        M = G * self.ntok
        ws = type('WS', (), {})()
        ws.B, ws.M, ws.G = B, M, G
        ws.a = torch.zeros(2, M, self.Kpad, dtype=torch.bfloat16, device=dev)     # cls rows / pad columns stay 0
        # Where the backward runs as the dgrad chain (capi.hip: wgrad_chain_ok) a DENSE last block is four launches + its share of a grouped
        # wgrad; its class-rows-only variant keeps the seven paired launches and costs more than the rows it skips save (cfg-2 same-box:
        # 1.527 -> 1.508 ms dense).  At cfg-3's 188 k rows the class-rows-only block stays (279 against 290 ms).
        chain = (WGRAD_GROUP > 0 and M <= 8192 and not self.group and not self.precise and FUSED_BLOCKS and FUSED_BWD
                 and D in (192, 384) and self.H * 64 == D and self.ntok <= 32)
        ws.cls_only = CLS_ONLY and not chain
        ws.blocks = _BlockWorkspace(self.depth, G, self.ntok, D, self.H, self.hidden, dev, self.split, self.ln_fuse, cls_only=ws.cls_only,
                                    precise=self.precise)
        bhn = max(G * self.H * self.ntok, (self.ntok * self.enc_heads * G) if self.group else 0,
                  (B * self.H * self.ntok2) if self.group else 0)
        ring = None
        if WGRAD_GROUP > 0 and M <= 8192 and not self.group and not self.precise:
            slot = int(self.lib.s3d_block_wgrad_slot_bytes(ctypes.byref(ws.blocks.shape)))
            ring = (min(WGRAD_GROUP, 6, self.depth), max(1, min(DGRAD_SPLITK, 4)), slot)
        ws.scratch = _BlockScratch(M, D, self.H, self.hidden, bhn, dev, depth=self.depth, precise=self.precise, wgrad_ring=ring)
        ws.sc1, ws._cls1 = _cls_scratch(ws.scratch.c, M, D, dev, self.precise) if ws.cls_only else (ws.scratch.c, None)       # scratch table of the (first) pass
        if self.group:
            f32 = dict(dtype=torch.float32, device=dev)
            b16 = dict(dtype=torch.bfloat16, device=dev)
            ws.blocks2 = _BlockWorkspace(self.depth, B, self.ntok2, D, self.H, self.hidden, dev, self.split, self.ln_fuse, cls_only=CLS_ONLY,
                                         precise=self.precise)
            ws.M2 = B * self.ntok2
            ws.sc2, ws._cls2 = _cls_scratch(ws.scratch.c, ws.M2, D, dev, self.precise) if CLS_ONLY else (ws.scratch.c, None)    # ... of pass 2
            e = type('ENC', (), {})()
            e.x_in = torch.empty(M, D, **f32); e.s1 = torch.empty(M, D, **f32); e.x1 = torch.empty(M, D, **f32)
            e.s2 = torch.empty(M, D, **f32); e.stats = torch.empty(4, M, **f32)
            e.lse = torch.empty(self.ntok * self.enc_heads * G, **f32)
            e.xin = torch.empty(2, M, D, **b16); e.qkv = torch.empty(2, M, 3 * D, **b16); e.att = torch.empty(2, M, D, **b16)
            e.x1p = torch.empty(2, M, D, **b16); e.fpre = torch.empty(M, D, **b16); e.f = torch.empty(2, M, D, **b16)
            e.fpre_lo = torch.empty(M, D, **b16) if self.precise else None
            e.acts = L.fill(L.S3dEncActs(), x_in=e.x_in, s1=e.s1, x1=e.x1, s2=e.s2, x_out=ws.blocks.x[0], mean1=e.stats[0],
                            rstd1=e.stats[1], mean2=e.stats[2], rstd2=e.stats[3], lse=e.lse, xin_hi=e.xin[0], xin_lo=e.xin[1],
                            qkv_hi=e.qkv[0], qkv_lo=e.qkv[1], att_hi=e.att[0], att_lo=e.att[1], x1_hi=e.x1p[0], x1_lo=e.x1p[1],
                            fpre=e.fpre, f_hi=e.f[0], f_lo=e.f[1], fpre_lo=e.fpre_lo)
            e.shape = L.S3dEncShape(G=G, Nb=self.ntok, D=D, H=self.enc_heads, Dff=D, eps=1e-5, split=1 if self.split else 0,
                                    dropout_p=self.dropout_p, seed=self.dropout_seed.data_ptr())
            ws.enc = e
            self._ensure_attn_mask(ws)
            ws.gstats = torch.empty(2, G, **f32)              # final-norm statistics of the pass-1 cls rows
            ws.gfeat = torch.empty(G, D, **f32)               # norm(x)[:, 0] of pass 1  -> tokens of pass 2
            ws.dgfeat = torch.empty(G, D, **f32)
        ws.fstats = torch.empty(2, B, dtype=torch.float32, device=dev)
        ws.last = ws.blocks2 if self.group else ws.blocks      # the block pass whose cls rows feed the head
        ws.ntok_last = self.ntok2 if self.group else self.ntok
        ws.feat = torch.empty(B, D, dtype=torch.float32, device=dev)
        ws.logits = torch.empty(B, self.C, dtype=torch.float32, device=dev)
        ws.dlogits = torch.empty(B, self.C, dtype=torch.float32, device=dev)
        ws.dfeat = torch.empty(B, D, dtype=torch.float32, device=dev)
        ws.loss = torch.zeros(2, dtype=torch.float32, device=dev)
        ws.head_scratch = torch.empty(self.C + B, dtype=torch.float32, device=dev)
        self._ws[B] = ws
        return ws

    # ------------------------------------------------------------------ forward
    def forward(self, x, block_ranges=None, before_range=None):
        """x: [B,1,V,V,V] float32 on the device (the trainer's voxel.float(), train_cls_voxel.py:276) -> logits."""
        ws = self.forward_features(x, block_ranges, before_range)
        a, lib, s, B = self.arena, self.lib, L.current_stream(), x.shape[0]
        ln = L.fill(L.S3dLnArgs(), x=ws.last.x[self.depth], ldx=ws.ntok_last * self.D, rows=B, D=self.D, eps=LN_EPS,
                    gamma=a.param('norm.weight'), beta=a.param('norm.bias'), out_f32=ws.feat, ldo=self.D,
                    mean=ws.fstats[0], rstd=ws.fstats[1])
        L.check(lib.s3d_layernorm_fwd(ctypes.byref(ln), s), 'final norm')
        L.check(lib.s3d_head_fwd(ctypes.byref(self._head_args(ws)), s), 'head_fwd')
        self._loss_end_done = False
        return ws.logits

    def forward_features(self, x, block_ranges=None, before_range=None):
        """Everything up to the last block's output (the class rows of ws.last.x[depth] feed the final norm + head).
        block_ranges = [(first, last), ...] ascending: the first block pass as s3d_blocks_fwd_range calls with before_range(i) in front of
        range i > 0 (parallel.ShardedDataParallelTrainer: range i's parameters arrive by all-gather while ranges < i compute)."""
        ws = self.forward_tokens(x)
        if block_ranges is None:
            L.check(self.lib.s3d_blocks_fwd(ctypes.byref(ws.blocks.shape), self.bparams, ws.blocks.acts, self.depth, L.current_stream()), 'blocks_fwd')
        else:
            for i, (first, last) in enumerate(block_ranges):
                if before_range is not None and i > 0:
                    before_range(i)                 # (range 0's parameters were awaited by the caller: the tokenizer needs them too)
                self.forward_blocks(ws, first, last)
        return self.forward_tail(ws)

    def forward_tokens(self, x):
        """Tokenizer (+ the seq-first encoder layer of group_embed): fills the input of block 0 of the (first) block pass."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(), 'voxel grid must be a contiguous fp32 device tensor'
        B, Cc, H, W, V = x.shape
        assert Cc == 1 and H == self.V and W == self.V and V == self.V, \
            f"Input voxel size ({H}*{W}*{V}) doesn't match model ({self.V}*{self.V}*{self.V})."   # embed_layer_3d_modality.py:36
        ws = self.workspace(B)
        ws._ln_aux_cont = None                  # a new forward: its backward starts with fresh row-statistics vectors
        lib, s, a = self.lib, L.current_stream(), self.arena
        fa = L.fill(L.S3dFoldArgs(), x=x, a_hi=ws.a[0], a_lo=ws.a[1], lda=self.Kpad, B=B, V=self.V, c=self.c, P=self.P,
                    mode=self.fold_mode)
        L.check(lib.s3d_voxel_fold(ctypes.byref(fa), s), 'voxel_fold')
        ck = self.conv_key
        if self.Kpad != self.Kc:
            w_hi, w_lo = self.conv_hi, self.conv_lo
        else:
            w_hi, w_lo = a.hi_of(ck + '.weight'), a.lo_of(ck + '.weight')
        tok_out = ws.enc.x_in if self.group else ws.blocks.x[0]
        g = L.fill(L.S3dGemmArgs(), A_hi=ws.a[0], A_lo=ws.a[1], lda=self.Kpad, B_hi=w_hi, B_lo=w_lo, ldb=self.Kpad,
                   M=ws.M, N=self.D, K=self.Kpad, bias=a.param(ck + '.bias'), C=tok_out, ldc=self.D,
                   alpha=(1.0 / self.P if self.fold_mode == 0 else 1.0),
                   cls=a.param('group_cls_token' if self.group else 'cls_token'),
                   pos=a.param('group_pos_embed' if self.group else 'voxel_pos_embed'), ntok=self.ntok)
        L.check(lib.s3d_gemm(0, 0, 1 if self.split else 0, 3, ctypes.byref(g), 1, s), 'tokenizer gemm')
        if self.group:
            # seq-first encoder layer over the B*P*P axis, then pass 1 of the blocks on (B*P*P, P+1, D)
            L.check(lib.s3d_encoder_layer_fwd(ctypes.byref(ws.enc.shape), ctypes.byref(self.eparams),
                                              ctypes.byref(ws.enc.acts), s), 'encoder_layer_fwd')
        return ws

    def forward_blocks(self, ws, first, last):
        """Blocks first .. last of the (first) block pass (s3d_blocks_fwd_range)."""
        L.check(self.lib.s3d_blocks_fwd_range(ctypes.byref(ws.blocks.shape), self.bparams, ws.blocks.acts, self.depth, first, last,
                                              L.current_stream()), 'blocks_fwd_range')

    def forward_tail(self, ws):
        """What follows the (first) block pass: nothing for the default embedding; group_embed's pass-1 final norm, token assembly and
        the second pass over the same blocks (vit_3d_2d_pretrain.py:481-496)."""
        if self.group:
            lib, s, a, B = self.lib, L.current_stream(), self.arena, ws.B
            # norm(x)[:, 0] of every group -> '(b px py) c -> b (px py) c' + cls + voxel_pos_embed -> pass 2 (same blocks)
            ln1 = L.fill(L.S3dLnArgs(), x=ws.blocks.x[self.depth], ldx=self.ntok * self.D, rows=ws.G, D=self.D, eps=LN_EPS,
                         gamma=a.param('norm.weight'), beta=a.param('norm.bias'), out_f32=ws.gfeat, ldo=self.D,
                         mean=ws.gstats[0], rstd=ws.gstats[1])
            L.check(lib.s3d_layernorm_fwd(ctypes.byref(ln1), s), 'pass-1 final norm')
            L.check(lib.s3d_assemble_tokens(L.ptr(ws.gfeat), L.ptr(a.param('cls_token')), L.ptr(a.param('voxel_pos_embed')),
                                            L.ptr(ws.blocks2.x[0]), ctypes.c_long(B), self.P * self.P, self.D, s), 'assemble')
            L.check(lib.s3d_blocks_fwd(ctypes.byref(ws.blocks2.shape), self.bparams, ws.blocks2.acts, self.depth, s), 'blocks_fwd 2')
        return ws

    def loss_of_features(self, B, target, weight=None):
        """The loss end of a training forward whose features are in place (forward_features / forward_tokens + forward_blocks +
        forward_tail): final norm + head + F.cross_entropy, fused where forward_loss fuses it."""
        if self.am or not FUSE_LOSS_END or self.C > 256 or self.D > 1024 or self.precise:
            ws = self.workspace(B)
            a, lib, s = self.arena, self.lib, L.current_stream()
            ln = L.fill(L.S3dLnArgs(), x=ws.last.x[self.depth], ldx=ws.ntok_last * self.D, rows=B, D=self.D, eps=LN_EPS,
                        gamma=a.param('norm.weight'), beta=a.param('norm.bias'), out_f32=ws.feat, ldo=self.D,
                        mean=ws.fstats[0], rstd=ws.fstats[1])
            L.check(lib.s3d_layernorm_fwd(ctypes.byref(ln), s), 'final norm')
            L.check(lib.s3d_head_fwd(ctypes.byref(self._head_args(ws)), s), 'head_fwd')
            self._loss_end_done = False
            return self.cross_entropy(B, target, weight)
        return self.head_loss(B, target, weight)

    def head_loss(self, B, target, weight=None, grad_scale=1.0):
        """Final norm -> Linear head -> F.cross_entropy -> d(logits) -> d(feat) -> final-norm backward in two launches
        (s3d_head_loss_fused) instead of six; leaves d(loss)/d(x_final) where backward() expects it.  Linear head only."""
        assert not self.am, 'head_loss: the AM-softmax head takes the unfused kernels'
        ws = self.workspace(B)
        a, D, nt, sc = self.arena, self.D, ws.ntok_last, ws.scratch
        assert target.dtype == torch.int64 and target.is_cuda
        if not hasattr(ws, 'hl_scratch'):
            ws.hl_scratch = torch.empty(B * (2 * D + 1), dtype=torch.float32, device=self.device)
        # a dense last block reads all of d(x_out): zero except at the class rows -- cleared by the launch itself (zero_tokens), not by a fill
        hl = L.fill(L.S3dHeadLossArgs(), zero_tokens=0 if ws.cls_only else nt - 1,
                    x=ws.last.x[self.depth], ldx=nt * D, B=B, D=D, C=self.C, eps=LN_EPS,
                    gamma=a.param('norm.weight'), beta=a.param('norm.bias'), W=a.param('voxel_head.weight'),
                    bias=a.param('voxel_head.bias'), target=target, weight=weight, grad_scale=grad_scale, feat=ws.feat,
                    mean=ws.fstats[0], rstd=ws.fstats[1], logits=ws.logits, dlogits=ws.dlogits, loss=ws.loss, dx=sc.dx_a,
                    dx_bf=sc.dx_a_bf, lddx=nt * D, dW=a.grad('voxel_head.weight'), dbias=a.grad('voxel_head.bias'),
                    dgamma=a.grad('norm.weight'), dbeta=a.grad('norm.bias'), scratch=ws.hl_scratch)
        L.check(self.lib.s3d_head_loss_fused(ctypes.byref(hl), L.current_stream()), 'head_loss_fused')
        self._loss_end_done = True
        return ws.loss[0]

    def _head_args(self, ws):
        a = self.arena
        h = L.S3dHeadArgs()
        if self.am:
            L.fill(h, feat=ws.feat, B=ws.B, D=self.D, C=self.C, W=a.param('voxel_head.W'), logits=ws.logits,
                   am_softmax=1, am_scale=30.0, dlogits=ws.dlogits, dfeat=ws.dfeat, dW=a.grad('voxel_head.W'),
                   scratch=ws.head_scratch)
        else:
            L.fill(h, feat=ws.feat, B=ws.B, D=self.D, C=self.C, W=a.param('voxel_head.weight'),
                   bias=a.param('voxel_head.bias'), logits=ws.logits, am_softmax=0, am_scale=1.0, dlogits=ws.dlogits,
                   dfeat=ws.dfeat, dW=a.grad('voxel_head.weight'), dbias=a.grad('voxel_head.bias'),
                   scratch=ws.head_scratch)
        return h

    # ------------------------------------------------------------------ loss
    def cross_entropy(self, B, target, weight=None, grad_scale=1.0):
        """F.cross_entropy(logits, target[, weight]) (train_cls_voxel.py:282-285) on the logits of the last forward;
        fills ws.loss[0] and ws.dlogits."""
        ws = self.workspace(B)
        assert target.dtype == torch.int64 and target.is_cuda
        ce = L.fill(L.S3dCeArgs(), logits=ws.logits, target=target, weight=weight, rows=B, C=self.C, loss=ws.loss,
                    dlogits=ws.dlogits, grad_scale=grad_scale)
        L.check(self.lib.s3d_cross_entropy(ctypes.byref(ce), L.current_stream()), 'cross_entropy')
        return ws.loss[0]

    # ------------------------------------------------------------------ backward
    def backward_begin(self, B, dlogits=None):
        """head backward + final-norm backward: leaves d(loss)/d(x_final) in the scratch ping-pong buffer."""
        ws = self.workspace(B)
        lib, s, a, D = self.lib, L.current_stream(), self.arena, self.D
        if getattr(self, '_loss_end_done', False):
            # head_loss() already produced d(loss)/d(x_final) AND accumulated the head / final-norm gradients: a second head backward
            # from caller-supplied d(logits) would add them twice
            self._loss_end_done = False
            if dlogits is not None:
                raise RuntimeError('backward(dlogits=...) after forward_loss(): the fused loss end has already accumulated the head and '
                                   'final-norm gradients; call forward() + cross_entropy() when you want to supply d(logits) yourself')
            return ws
        if dlogits is not None and dlogits.data_ptr() != ws.dlogits.data_ptr():
            ws.dlogits.copy_(dlogits)
        L.check(lib.s3d_head_bwd(ctypes.byref(self._head_args(ws)), s), 'head_bwd')
        sc = ws.scratch
        if not ws.cls_only:
            sc.zero_dx_a()                  # with cls_only_block the last block reads the class rows of d(x_out) only
        nt = ws.ntok_last
        lb = L.fill(L.S3dLnBwdArgs(), dy=ws.dfeat, lddy=D, x=ws.last.x[self.depth], ldx=nt * D,
                    mean=ws.fstats[0], rstd=ws.fstats[1], gamma=a.param('norm.weight'), dx=sc.dx_a, lddx=nt * D,
                    dx_bf=sc.dx_a_bf, lddxbf=nt * D, dgamma=a.grad('norm.weight'), dbeta=a.grad('norm.bias'),
                    rows=B, D=D, dx_bf_lo=sc.dx_a_lo if self.precise else None)
        L.check(lib.s3d_layernorm_bwd(ctypes.byref(lb), s), 'final norm bwd')
        return ws

    def backward_segment(self, ws, first, last, with_tokenizer):
        if self.group and first == self.depth - 1:
            self._group_pass2_backward(ws)              # pass-2 blocks, token assembly, pass-1 final norm
        self.blocks_backward_range(ws, first, last)
        if with_tokenizer:
            if self.group:
                sc = ws.scratch
                L.check(self.lib.s3d_encoder_layer_bwd(ctypes.byref(ws.enc.shape), ctypes.byref(self.eparams),
                                                       ctypes.byref(self.egrads), ctypes.byref(ws.enc.acts),
                                                       ctypes.byref(sc.c), L.current_stream()), 'encoder_layer_bwd')
                self._tokenizer_backward(ws, dx=sc.dx_b, dx_bf=sc.dx_b_bf, dx_lo=sc.dx_b_lo if self.precise else None)
            else:
                self._tokenizer_backward(ws)

    def _group_pass2_backward(self, ws):
        """Backward of the second application of the shared blocks (gradients ACCUMULATE into the same arena slots,
        vit_3d_2d_pretrain.py:481-484 vs :493-496), of the token assembly and of the pass-1 final norm."""
        lib, s, a, D, sc = self.lib, L.current_stream(), self.arena, self.D, ws.scratch
        L.check(lib.s3d_blocks_bwd(ctypes.byref(ws.blocks2.shape), self.bparams, self.bgrads, ws.blocks2.acts,
                                   ctypes.byref(ws.sc2), self.depth - 1, 0, s), 'blocks_bwd pass 2')
        pg = L.fill(L.S3dPosGradArgs(), dx=sc.dx_a, groups=ws.B, ntok=self.ntok2, D=D, dpos=a.grad('voxel_pos_embed'),
                    dcls=a.grad('cls_token'))
        L.check(lib.s3d_token_grads(ctypes.byref(pg), s), 'pass-2 token grads')
        L.check(lib.s3d_assemble_tokens_bwd(L.ptr(sc.dx_a), L.ptr(ws.dgfeat), ctypes.c_long(ws.B), self.P * self.P, D, s),
                'assemble bwd')
        if not ws.cls_only:
            sc.zero_dx_a()
        lb = L.fill(L.S3dLnBwdArgs(), dy=ws.dgfeat, lddy=D, x=ws.blocks.x[self.depth], ldx=self.ntok * D,
                    mean=ws.gstats[0], rstd=ws.gstats[1], gamma=a.param('norm.weight'), dx=sc.dx_a, lddx=self.ntok * D,
                    dx_bf=sc.dx_a_bf, lddxbf=self.ntok * D, dgamma=a.grad('norm.weight'), dbeta=a.grad('norm.bias'),
                    rows=ws.G, D=D, dx_bf_lo=sc.dx_a_lo if self.precise else None)
        L.check(lib.s3d_layernorm_bwd(ctypes.byref(lb), s), 'pass-1 final norm bwd')

    def backward(self, B, dlogits=None, *, segments=None, on_segment=None):
        """Accumulates d(loss)/d(param) into the gradient arena, given d(loss)/d(logits) (ws.dlogits by default).
        `segments` = [(first_block, last_block), ...] in backward order; on_segment(i) fires when every gradient of
        segment i is complete (the last one includes the tokenizer) -- the data-parallel reducer hooks in there."""
        ws = self.backward_begin(B, dlogits)
        segs = segments or [(self.depth - 1, 0)]
        for i, (first, last) in enumerate(segs):
            self.backward_segment(ws, first, last, i == len(segs) - 1)
            if on_segment is not None:
                on_segment(i)

    def grad_buckets(self, n_buckets=3, blocks_per_bucket=None, block_counts=None):
        """Splits the backward into `n_buckets` block ranges and returns (segments, [(start, end) arena slices]) such
        that slice i holds exactly the gradients that are final once segment i has run (arena is in forward order).
        Bucket sizes shrink geometrically in backward order (depth 12, 4 buckets: blocks 11-6, 5-3, 2-1, 0 + tokenizer):
        the all-reduce of the LAST bucket cannot hide behind any compute, so it is the smallest; the first one has the
        whole rest of backward to hide behind, so it is the largest.
        blocks_per_bucket = k: UNIFORM buckets of k blocks instead (depth 12, k = 2: 11-10 | 9-8 | .. | 1-0 + tokenizer) -- the wire then
        starts after k blocks of backward instead of after half of it, which matters when the wire time of the whole gradient is about
        as long as the backward itself (cfg-2: 0.6 ms of fp32 ring all-reduce against a 0.7 ms backward, profiles/r05_dp_branch_tax.txt)."""
        if block_counts:
            # explicit bucket sizes in BACKWARD order (depth 12, [3, 3, 3, 2, 1]: 11-9 | 8-6 | 5-3 | 2-1 | 0 + tokenizer): the sharded trainer's
            # last bucket -- the only one whose wire time is exposed, and the first the next forward needs -- is the smallest
            counts = [int(c) for c in block_counts]
            assert all(c >= 1 for c in counts) and sum(counts) == self.depth, f'block_counts {counts} must be positive and sum to depth {self.depth}'
            n = len(counts)
            bounds = [self.depth]
            for c in counts:
                bounds.append(bounds[-1] - c)
            bounds = bounds[::-1]
        elif blocks_per_bucket:
            k = max(1, int(blocks_per_bucket))
            n = (self.depth + k - 1) // k
            bounds = [max(0, self.depth - (n - j) * k) for j in range(n)] + [self.depth]
            bounds[0] = 0
        else:
            n = max(1, min(n_buckets, self.depth))
            # boundaries (in blocks from the input side): depth / 2^(n-1), ..., depth / 2, depth
            bounds = [0] + [max(k, self.depth // 2 ** (n - k)) for k in range(1, n)] + [self.depth]
            for k in range(1, n + 1):                       # strictly increasing even for tiny depths
                bounds[k] = max(bounds[k], bounds[k - 1] + 1)
            bounds[n] = self.depth
        segments, slices = [], []
        end = self.arena.numel
        for k in range(n, 0, -1):
            lo_blk, hi_blk = bounds[k - 1], bounds[k] - 1
            start = self.arena.offsets[f'blocks.{lo_blk}.norm1.weight'] if lo_blk > 0 else 0
            segments.append((hi_blk, lo_blk))
            slices.append((start, end))
            end = start
        return segments, slices

    @contextlib.contextmanager
    def owning_grads(self):
        """Scope in which the caller owns the gradient arena -- zeroed by its previous optimizer step, exactly one backward, nothing else
        accumulates into it: the grouped wgrads STORE dW / db instead of read-modify-write (S3dBlockScratch::wg_overwrite).  A backward
        outside such a scope (gradient accumulation, autograd.Function callers, gradient checks) accumulates as torch does."""
        prev, self.grads_owned = self.grads_owned, True
        try:
            yield
        finally:
            self.grads_owned = prev

    def blocks_backward_range(self, ws, first, last):
        fill = getattr(self, '_fill', None)
        if fill is not None:
            ws.sc1.adam_fill = ctypes.addressof(fill['args'])
        # the caller owns the gradient arena for this step (train_step, the data-parallel trainers: zeroed by the previous Adam, nothing else
        # accumulates into it): the grouped wgrads store instead of read-modify-write
        ws.sc1.wg_overwrite = 1 if (self.grads_owned and WGRAD_OVERWRITE and not self.group and self.images is None) else 0
        if (first, last) != (self.depth - 1, 0) and ws.scratch.ln_aux is not None and not self.group:
            # a backward issued in segments: the weights-only vectors of the fused LayerNorm backward once per backward -- in front of its
            # first segment, and again whenever this segment does not CONTINUE the previous one of the same forward (a lone
            # blocks_backward_range(ws, 5, 3), a segment after a new forward ...): stale u / c vectors would give silently wrong LayerNorm
            # and dx gradients.  An optimizer slice that runs BETWEEN the segments of one backward (sliced Adam) touches only
            # blocks whose backward is done: the vectors of the blocks still to come stay valid.
            if first == self.depth - 1 or getattr(ws, '_ln_aux_cont', None) != first:
                L.check(self.lib.s3d_blocks_ln_aux(ctypes.byref(ws.blocks.shape), self.bparams, ctypes.byref(ws.sc1), self.depth - 1, 0,
                                                   L.current_stream()), 'blocks_ln_aux')
            ws._ln_aux_cont = last - 1
            ws.sc1.ln_aux_valid = 1
        try:
            L.check(self.lib.s3d_blocks_bwd(ctypes.byref(ws.blocks.shape), self.bparams, self.bgrads, ws.blocks.acts,
                                            ctypes.byref(ws.sc1), first, last, L.current_stream()), 'blocks_bwd')
        finally:
            ws.sc1.adam_fill = None
            ws.sc1.ln_aux_valid = 0
        if fill is not None:
            n = fill['n'].value
            fill['done'] += [(int(fill['ranges'][2 * i]), int(fill['ranges'][2 * i + 1])) for i in range(n)]

    def _adam_fill_begin(self):
        """Arms S3dAdamFill for the backward that follows (after adam_begin): the update of every block's GEMM parameters rides on the
        next block's backward launches."""
        a = self.arena
        ranges, n = (ctypes.c_long * 256)(), ctypes.c_int(0)
        args = L.fill(L.S3dAdamFill(), p=a.p, g=a.g, m=a.m, v=a.v, hi=a.hi, lo=a.lo, state=self.adam_state, zero_grad=1,
                      filled=ctypes.addressof(ranges), filled_cap=256, n_filled=ctypes.addressof(n))
        self._fill = dict(args=args, ranges=ranges, n=n, done=[])

    def _adam_fill_finish(self):
        """The rest of optimizer.step(): ONE launch over everything the filler shares did not cover (LayerNorm parameters, the block
        whose backward ran last, tokenizer / positional embedding / final norm / head)."""
        fill, a = self._fill, self.arena
        self._fill = None
        rest, pos = [], 0
        for off, cnt in sorted(fill['done']):
            assert off >= pos and off % 4 == 0 and cnt % 4 == 0, 'filled ranges must be disjoint and float4-aligned'
            if off > pos:
                rest.append((pos, off - pos))
            pos = off + cnt
        if pos < a.numel:
            rest.append((pos, a.numel - pos))
        while len(rest) > 64:                    # (never at depth 12: 13 gaps) merge the smallest gap rather than fail
            raise RuntimeError('adam fill: more than 64 leftover ranges')
        flat = (ctypes.c_long * (2 * len(rest)))(*[v for r in rest for v in r])
        L.check(self.lib.s3d_adam_apply_ranges(L.ptr(a.p), L.ptr(a.g), L.ptr(a.m), L.ptr(a.v), L.ptr(a.hi), L.ptr(a.lo), flat, len(rest),
                                               L.ptr(self.adam_state), 1, L.current_stream()), 'adam ranges')
        self.adam_fill_stats = dict(filled=sum(c for _, c in fill['done']), rest=sum(c for _, c in rest), ranges=len(rest))

    def _tokenizer_backward(self, ws, dx=None, dx_bf=None, dx_lo=None):
        lib, s, a, D, sc = self.lib, L.current_stream(), self.arena, self.D, ws.scratch
        dx = sc.dx_a if dx is None else dx
        dx_lo = (sc.dx_a_lo if self.precise else None) if dx_bf is None else dx_lo
        dx_bf = sc.dx_a_bf if dx_bf is None else dx_bf
        ck = self.conv_key
        padded = self.Kpad != self.Kc
        gw = self.conv_gpad if padded else a.grad(ck + '.weight')
        if padded:
            self.conv_gpad.zero_()
        g = L.fill(L.S3dGemmArgs(), A_hi=dx_bf, lda=D, B_hi=ws.a[0], ldb=self.Kpad, M=D, N=self.Kpad, K=ws.M,
                   C=gw, ldc=self.Kpad, alpha=(1.0 / self.P if self.fold_mode == 0 else 1.0))
        if self.precise:
            L.fill(g, A_lo=dx_lo, B_lo=ws.a[1])
        L.check(lib.s3d_gemm(1, 1, 1 if self.precise else 0, 6, ctypes.byref(g), 0, s), 'tokenizer wgrad')
        if padded:
            a.grad(ck + '.weight').view(D, self.Kc).add_(self.conv_gpad[:, :self.Kc])
        pg = L.fill(L.S3dPosGradArgs(), dx=dx, groups=ws.G, ntok=self.ntok, D=D,
                    dpos=a.grad('group_pos_embed' if self.group else 'voxel_pos_embed'),
                    dcls=a.grad('group_cls_token' if self.group else 'cls_token'), dbias=a.grad(ck + '.bias'))
        L.check(lib.s3d_token_grads(ctypes.byref(pg), s), 'token grads')

    # ------------------------------------------------------------------ optimizer
    def adam_step(self, zero_grad=True, wire=None):
        """wire: bf16 tensor of the arena's layout holding the (all-reduced) gradient -- the data-parallel wire format; the
        fp32 gradient arena is then only zeroed."""
        a = self.arena
        if wire is None:
            L.check(self.lib.s3d_adam_step(L.ptr(a.p), L.ptr(a.g), L.ptr(a.m), L.ptr(a.v), L.ptr(a.hi), L.ptr(a.lo),
                                           ctypes.c_long(a.numel), L.ptr(self.adam_state), 1 if zero_grad else 0,
                                           L.current_stream()), 'adam')
        else:
            assert wire.dtype == torch.bfloat16 and wire.numel() == a.numel and wire.is_cuda
            L.check(self.lib.s3d_adam_step_wire(L.ptr(a.p), L.ptr(a.g), L.ptr(wire), L.ptr(a.m), L.ptr(a.v), L.ptr(a.hi), L.ptr(a.lo),
                                                ctypes.c_long(a.numel), L.ptr(self.adam_state), 1 if zero_grad else 0,
                                                L.current_stream()), 'adam (bf16 wire)')
        self._refresh_conv_planes()

    def adam_begin(self):
        """optimizer.step() bookkeeping (step count, bias corrections) once per step; adam_apply slices follow it."""
        L.check(self.lib.s3d_adam_begin(L.ptr(self.adam_state), L.current_stream()), 'adam begin')

    def adam_apply(self, start, end, zero_grad=True, max_workgroups=0, wire=None):
        """Adam on the arena slice [start, end) on the current stream (gradients of the slice must be final there).  wire: the flat bf16
        gradient buffer of the data-parallel wire format (the slice's gradient is read from it, the fp32 arena only zeroed)."""
        a = self.arena
        off = lambda t, b: ctypes.c_void_p(t.data_ptr() + b * start)
        L.check(self.lib.s3d_adam_apply(off(a.p, 4), off(a.g, 4), None if wire is None else off(wire, 2), off(a.m, 4), off(a.v, 4),
                                        off(a.hi, 2), off(a.lo, 2), ctypes.c_long(end - start), L.ptr(self.adam_state),
                                        1 if zero_grad else 0, int(max_workgroups), L.current_stream()), 'adam slice')

    def adam_end(self):
        """Behind the last adam_apply of a step: what adam_step does after its kernel (the padded tokenizer-weight planes)."""
        self._refresh_conv_planes()

    def pack_grads(self, start, end, wire):
        """gradient arena [start, end) -> bf16 wire buffer [start, end) (round to nearest even), on the current stream."""
        a = self.arena
        L.check(self.lib.s3d_pack_bf16(ctypes.c_void_p(a.g.data_ptr() + 4 * start), ctypes.c_void_p(wire.data_ptr() + 2 * start),
                                       ctypes.c_long(end - start), L.current_stream()), 'pack_bf16')

    def zero_grad(self):
        self.arena.g.zero_()

    # ------------------------------------------------------------------ fused training step
    def train_step(self, x, target, weight=None):
        """zero_grad -> model(voxel) -> F.cross_entropy -> backward -> Adam  (train_cls_voxel.py:277-288), all on the
        HIP path.  Gradients are zeroed by the previous step's Adam kernel.  Returns the loss as a device scalar."""
        B = x.shape[0]
        self.advance_dropout_seed()                       # fresh masks every step (device-side, graph-replay safe)
        loss = self.forward_loss(x, target, weight)
        n_slices = getattr(self, 'update_slices', UPDATE_OVERLAP)
        if n_slices <= 0 or self.precise:
            if getattr(self, 'adam_fill', ADAM_FILL) and not self.precise and not self.group and self.images is None and self.world_size == 1:
                # optimizer.step() inside loss.backward(): the bookkeeping first, the GEMM parameters of block i + 1 on block i's launches,
                # the rest in one launch at the end (bitwise the results of adam_step)
                self.adam_begin()
                self._adam_fill_begin()
                self.grads_owned = True
                try:
                    self.backward(B)
                    self._adam_fill_finish()
                finally:
                    self._fill = None
                    self.grads_owned = False
                self._refresh_conv_planes()
                return loss
            # the fused step owns the gradient arena (zeroed by the previous step's Adam, no accumulation across backward calls unless the
            # image branch adds its own backward): the grouped wgrads may store instead of read-modify-write
            self.grads_owned = True
            try:
                self.backward(B)
            finally:
                self.grads_owned = False
            self.adam_step(zero_grad=True)
            return loss
        main = torch.cuda.current_stream()
        if getattr(self, '_update_stream', None) is None:
            self._update_stream = torch.cuda.Stream()
        side = self._update_stream
        self.adam_begin()
        segs, slices = self.grad_buckets(n_slices)

        def update_slice(i):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self.adam_apply(*slices[i], max_workgroups=UPDATE_WORKGROUPS)
        self.backward(B, segments=segs, on_segment=update_slice)
        main.wait_stream(side)
        self._refresh_conv_planes()
        return loss

    def forward_loss(self, x, target, weight=None, block_ranges=None, before_range=None):
        """model(voxel) + F.cross_entropy for a TRAINING step; with the Linear head (C <= 256, D <= 1024) the loss end runs fused
        (head_loss): the head / final-norm gradients are accumulated right here, so follow it with exactly one backward() (no
        dlogits) -- for inference or custom d(logits) use forward() + cross_entropy()."""
        self.forward_features(x, block_ranges, before_range)
        return self.loss_of_features(x.shape[0], target, weight)     # s3d_head_loss_fused: Linear head, C <= 256, D <= 1024

    def lwf_train_step(self, x, target, img, img_target, lambda_weight=0.1, weight=None):
        """One learning-without-forgetting step (train_cls_voxel.py:240-268): loss = CE(model(voxel), cls_idx) +
        lambda * CE(model.forward_images(images), teacher labels); both backward passes accumulate into the same gradient
        arena, then Adam.  Returns (total loss, voxel loss, image loss) as device scalars."""
        if self.images is None:
            raise RuntimeError('the engine was built without image_branch=True')
        B, Bi = x.shape[0], img.shape[0]
        self.forward(x)
        lv = self.cross_entropy(B, target, weight)
        self.images.forward(img)
        li = self.images.cross_entropy(Bi, img_target, grad_scale=lambda_weight)
        self.backward(B)
        self.images.backward(Bi)
        self.adam_step(zero_grad=True)
        return lv + lambda_weight * li, lv, li

    def capture_train_step(self, B, weight=None):
        """Captures train_step into a HIP graph over static input buffers; returns (graph, static_x, static_y, loss)."""
        ws = self.workspace(B)                      # (may grow the shared attention-mask buffer: that clears the cache, see _ensure_attn_mask)
        mask = getattr(self, '_attn_mask', None) if self.group else None
        key = (B, None if weight is None else weight.data_ptr(), self.dropout_p, getattr(self, 'update_slices', UPDATE_OVERLAP), getattr(self, 'adam_fill', ADAM_FILL),
               None if mask is None else mask.data_ptr(), self.backward_precision)     # model.train() / .eval() toggles keep both captures
        if key in self._graphs:
            return self._graphs[key]
        sx = torch.zeros(B, 1, self.V, self.V, self.V, dtype=torch.float32, device=self.device)
        sy = torch.zeros(B, dtype=torch.int64, device=self.device)
        # warm-up on a side stream (sets kernel attributes, allocates workspaces); state is restored afterwards
        snap = [t.clone() for t in (self.arena.p, self.arena.m, self.arena.v, self.arena.g, self.adam_state)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.train_step(sx, sy, weight)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for t, sv in zip((self.arena.p, self.arena.m, self.arena.v, self.arena.g, self.adam_state), snap):
            t.copy_(sv)
        self.refresh_weight_planes()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss = self.train_step(sx, sy, weight)
        self._graphs[key] = (graph, sx, sy, loss)
        return self._graphs[key]
