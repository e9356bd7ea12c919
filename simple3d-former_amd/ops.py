"""Thin tensor-level wrappers over the C ABI (one call = one libs3d_hip.so entry point).  Used by the parity tests and
by host code that needs a single operator; the training engine calls the same entry points directly."""
import ctypes

import torch

from . import _lib as L

EPI = dict(BF16_BIAS=0, GELU=1, RESID=2, TOKEN=3, F32=4, DGELU=5, ATOMIC=6, RELU=7, DRELU=8)


def _dev(t):
    assert t.is_cuda and t.is_contiguous(), 'device-resident contiguous tensor required'
    return t


def split_bf16(x, ld_out=None):
    """fp32 [rows, cols] -> (hi, lo) bf16 planes with x ~= hi + lo."""
    x = _dev(x.float())
    x2 = x.reshape(-1, x.shape[-1]) if x.dim() > 1 else x.reshape(1, -1)
    rows, cols = x2.shape
    ld = ld_out or cols
    hi = torch.zeros(rows, ld, dtype=torch.bfloat16, device=x.device)
    lo = torch.zeros_like(hi)
    L.check(L.lib().s3d_split_bf16(L.ptr(x2), L.ptr(hi), L.ptr(lo), ctypes.c_long(rows), ctypes.c_long(cols),
                                   ctypes.c_long(ld), L.current_stream()), 'split_bf16')
    return hi, lo


def gemm(ta, tb, split, epi, splitk=1, **fields):
    g = L.fill(L.S3dGemmArgs(), **fields)
    if 'alpha' not in fields:
        g.alpha = 1.0
    L.check(L.lib().s3d_gemm(int(ta), int(tb), int(split), EPI[epi] if isinstance(epi, str) else epi,
                             ctypes.byref(g), splitk, L.current_stream()), 'gemm')


def linear(x, w, bias=None, split=True):
    """y = x @ w^T + bias in fp32 out (x [M,K] fp32, w [N,K] fp32), via the (split-)bf16 MFMA GEMM."""
    M, K = x.shape
    N = w.shape[0]
    xh, xl = split_bf16(x)
    wh, wl = split_bf16(w)
    out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    gemm(0, 0, split, 'F32', A_hi=xh, A_lo=xl, lda=K, B_hi=wh, B_lo=wl, ldb=K, M=M, N=N, K=K,
         bias=None if bias is None else _dev(bias.float()), C=out, ldc=N)
    return out


def layernorm_fwd(x, gamma, beta, eps=1e-6, want_f32=True):
    rows, D = x.shape
    hi = torch.empty(rows, D, dtype=torch.bfloat16, device=x.device)
    lo = torch.empty_like(hi)
    out = torch.empty(rows, D, dtype=torch.float32, device=x.device) if want_f32 else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    a = L.fill(L.S3dLnArgs(), x=_dev(x), ldx=D, rows=rows, D=D, eps=eps, gamma=_dev(gamma), beta=_dev(beta), out_hi=hi,
               out_lo=lo, out_f32=out, ldo=D, mean=mean, rstd=rstd)
    L.check(L.lib().s3d_layernorm_fwd(ctypes.byref(a), L.current_stream()), 'layernorm_fwd')
    return out, hi, lo, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, dres=None):
    rows, D = x.shape
    dx = torch.empty_like(x)
    dx_bf = torch.empty(rows, D, dtype=torch.bfloat16, device=x.device)
    dg = torch.zeros(D, dtype=torch.float32, device=x.device)
    db = torch.zeros_like(dg)
    a = L.fill(L.S3dLnBwdArgs(), dy=_dev(dy), lddy=D, x=_dev(x), ldx=D, mean=mean, rstd=rstd, gamma=_dev(gamma),
               dres=dres, lddres=D, dx=dx, lddx=D, dx_bf=dx_bf, lddxbf=D, dgamma=dg, dbeta=db, rows=rows, D=D)
    L.check(L.lib().s3d_layernorm_bwd(ctypes.byref(a), L.current_stream()), 'layernorm_bwd')
    return dx, dx_bf, dg, db


def _drop_fields(drop):
    """drop = (p, seed tensor [1] int64 on the device, site) -> S3dAttnArgs / S3dGemmArgs dropout fields"""
    if not drop:
        return {}
    p, seed, site = drop
    return dict(drop_seed=seed, drop_site=int(site), drop_thr=int(p * 4294967296.0), drop_scale=1.0 / (1.0 - p))


def attention_fwd(qkv_hi, qkv_lo, Bb, H, N, D, sb, st, split=True, seg=0, drop=None, drop_mask=None, p_single_plane=0):
    rows = qkv_hi.shape[0]
    out_hi = torch.zeros(rows, D, dtype=torch.bfloat16, device=qkv_hi.device)
    out_lo = torch.zeros_like(out_hi)
    lse = torch.zeros(Bb * H * N, dtype=torch.float32, device=qkv_hi.device)
    a = L.fill(L.S3dAttnArgs(), qkv_hi=_dev(qkv_hi), qkv_lo=_dev(qkv_lo), ld=3 * D, out_hi=out_hi, out_lo=out_lo, ldo=D,
               lse=lse, Bb=Bb, H=H, N=N, D=D, sb=sb, st=st, scale=float((D // H) ** -0.5), seg=seg, drop_mask=drop_mask, p_single_plane=p_single_plane,
               **_drop_fields(drop))
    L.check(L.lib().s3d_attention_fwd(ctypes.byref(a), 1 if split else 0, L.current_stream()), 'attention_fwd')
    return out_hi, out_lo, lse


def attention_bwd(qkv_hi, out_hi, out_lo, lse, dout, Bb, H, N, D, sb, st, seg=0, drop=None, drop_mask=None):
    rows = qkv_hi.shape[0]
    dqkv = torch.zeros(rows, 3 * D, dtype=torch.bfloat16, device=qkv_hi.device)
    delta = torch.zeros(Bb * H * N, dtype=torch.float32, device=qkv_hi.device)
    a = L.fill(L.S3dAttnArgs(), qkv_hi=_dev(qkv_hi), ld=3 * D, out_hi=_dev(out_hi), out_lo=out_lo, ldo=D, lse=lse, Bb=Bb,
               H=H, N=N, D=D, sb=sb, st=st, scale=float((D // H) ** -0.5), dout=_dev(dout), lddo=D, dqkv=dqkv,
               lddq=3 * D, delta=delta, seg=seg, drop_mask=drop_mask, **_drop_fields(drop))
    L.check(L.lib().s3d_attention_bwd(ctypes.byref(a), L.current_stream()), 'attention_bwd')
    return dqkv


def voxel_fold(x, cell, patch, mode, ntok_rows, kpad):
    B, _, V, _, _ = x.shape
    a = torch.zeros(2, ntok_rows, kpad, dtype=torch.bfloat16, device=x.device)
    fa = L.fill(L.S3dFoldArgs(), x=_dev(x), a_hi=a[0], a_lo=a[1], lda=kpad, B=B, V=V, c=cell, P=patch, mode=mode)
    L.check(L.lib().s3d_voxel_fold(ctypes.byref(fa), L.current_stream()), 'voxel_fold')
    return a


def cross_entropy(logits, target, weight=None, grad_scale=1.0):
    rows, C = logits.shape
    loss = torch.zeros(2, dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits)
    a = L.fill(L.S3dCeArgs(), logits=_dev(logits), target=_dev(target), weight=weight, rows=rows, C=C, loss=loss,
               dlogits=dl, grad_scale=grad_scale)
    L.check(L.lib().s3d_cross_entropy(ctypes.byref(a), L.current_stream()), 'cross_entropy')
    return loss[0], dl
