"""GPU (MI355X) parity tests of the individual HIP kernels, each called THROUGH THE C ABI (libs3d_hip.so) and
compared with a plain fp32 PyTorch reference of the same operator on identical seeded inputs.

Tolerances (written next to each check): split-bf16 forward kernels carry ~16 mantissa bits -> 2e-4 relative to the
output scale; plain-bf16 backward kernels ~8 bits -> 2e-2 relative to the output rms."""
import ctypes
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import simple3d_former_amd as s3d
    from simple3d_former_amd import _lib as L
    from simple3d_former_amd import ops

DEV = 'cuda'


def rel_err(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def rms_err(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-30))


def test_library_loads_and_reports_version():
    assert L.lib().s3d_version() >= 100


def test_split_bf16_reconstructs_fp32():
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(37, 200, generator=g) * torch.logspace(-3, 3, 200)).to(DEV)
    hi, lo = ops.split_bf16(x)
    rec = hi.float() + lo.float()
    assert float(((rec - x).abs() / x.abs().clamp_min(1e-20)).max()) < 2 ** -15      # 16 mantissa bits
    assert torch.equal(hi, x.to(torch.bfloat16))                                        # hi = RNE(x), same as torch


@pytest.mark.parametrize('M,N,K', [(64, 40, 384), (100, 384, 216), (1664, 1152, 384), (1664, 384, 1536), (4096, 3072, 768),
                                   # point-path shapes: many rows, narrow n, k = 96 / 64 / 192 (k % 32 on the 128x128 LDS-DMA kernel), ragged M
                                   (70000, 96, 96), (65537, 96, 64), (33000, 192, 192), (66000, 48, 48),
                                   # long cfg-3 shapes: 128x256 tiles, sixteen waves, three stages (ragged last row tile)
                                   (33001, 768, 768), (66000, 256, 1024), (22000, 2304, 512), (12608, 3072, 768)])      # 12 608 = cfg-3 pass 2
@pytest.mark.parametrize('split', [True, False])
def test_gemm_forward_nt(M, N, K, split):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    ref = x.double() @ w.double().t() + b.double()
    got = ops.linear(x, w, b, split=split)
    e = rel_err(got, ref)
    assert e < (2e-5 if split else 2e-2), f'rel err {e:.3e}'


def test_gemm_catches_transposes():
    """A = identity-like with asymmetric B: a swapped C layout or operand transpose cannot pass."""
    M = N = K = 64
    x = torch.eye(M, K, device=DEV)
    w = (torch.arange(N * K, dtype=torch.float32, device=DEV).reshape(N, K) % 251) / 251.0
    got = ops.linear(x, w, None, split=True)
    assert rel_err(got, w.t()) < 1e-5


@pytest.mark.parametrize('M,N,K', [(78, 192, 576), (1664, 1536, 384), (1664, 384, 1152), (4096, 768, 3072),
                                   (70000, 96, 96), (66000, 192, 192), (33001, 48, 96), (131072, 56, 96),       # point path: k = 96 stays register-staged
                                   (20000, 768, 768), (16500, 3072, 768), (12608, 768, 3072), (12608, 3072, 768)])                                    # long + short reduction: 256x128 tiles, 8 waves
def test_gemm_dgrad_nn(M, N, K):
    """dx[m][i] = sum_o dy[m][o] W[o][i]: W is stored [K=o][N=i] (k-major B operand)."""
    g = torch.Generator().manual_seed(2)
    dy = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(K, N, generator=g) * 0.05).to(DEV)
    dyh, _ = ops.split_bf16(dy)
    wh, _ = ops.split_bf16(w)
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(0, 1, 0, 'F32', A_hi=dyh, lda=K, B_hi=wh, ldb=N, M=M, N=N, K=K, C=out, ldc=N)
    ref = dyh.double() @ wh.double()            # same bf16-rounded operands -> only accumulation-order error
    assert rel_err(out, ref) < 1e-5
    assert rms_err(out, dy.double() @ w.double()) < 1e-2


@pytest.mark.parametrize('rows,O,I', [(30, 64, 192), (78, 576, 192), (1664, 1536, 384), (1664, 384, 1536), (1664, 384, 216),
                                      (70016, 96, 96), (131072, 96, 48), (188160, 768, 768), (33000, 192, 96),    # long reductions
                                      (20032, 768, 3072), (16488, 512, 256), (12608, 3072, 768), (12608, 768, 768)])                                     # 256x128 tiles (output rows <= 1024), partial last k-tile
def test_gemm_wgrad_tn_with_bias_grad(rows, O, I):
    """dW[o][i] += sum_m dy[m][o] x[m][i]; db[o] += sum_m dy[m][o] (split-K fp32 atomics)."""
    g = torch.Generator().manual_seed(3)
    dy = torch.randn(rows, O, generator=g).to(DEV)
    x = torch.randn(rows, I, generator=g).to(DEV)
    dyh, _ = ops.split_bf16(dy)
    xh, _ = ops.split_bf16(x)
    dW = torch.zeros(O, I, dtype=torch.float32, device=DEV)
    db = torch.zeros(O, dtype=torch.float32, device=DEV)
    ops.gemm(1, 1, 0, 'ATOMIC', splitk=0, A_hi=dyh, lda=O, B_hi=xh, ldb=I, M=O, N=I, K=rows, C=dW, ldc=I, bias_grad=db)
    assert rel_err(dW, dyh.double().t() @ xh.double()) < 2e-5
    assert rel_err(db, dyh.double().sum(0)) < 2e-5
    # accumulation semantics: a second call adds
    ops.gemm(1, 1, 0, 'ATOMIC', splitk=0, A_hi=dyh, lda=O, B_hi=xh, ldb=I, M=O, N=I, K=rows, C=dW, ldc=I, bias_grad=db)
    assert rel_err(dW, 2 * (dyh.double().t() @ xh.double())) < 2e-5


def _gelu_grad(pre):
    a = pre.double().clone().requires_grad_(True)
    return torch.autograd.grad(F.gelu(a).sum(), a)[0]


@pytest.mark.parametrize('rows,O,I,epi,ld_scale', [
    (1664, 1152, 384, 'F32', 1),         # cfg-2 qkv:  dxn1 = dqkv @ Wqkv        || dWqkv += dqkv^T xn1      -> gemm_pair_dmat_kernel<4,3>
    (1664, 1536, 384, 'F32', 1),         # cfg-2 fc1:  dxn2 = dh @ W1            || dW1 += dh^T xn2          -> <4,3>
    (1664, 384, 1536, 'DGELU', 1),       # cfg-2 fc2:  dh = (dxo @ W2) gelu'     || dW2 += dxo^T hact        -> <5,3>
    (1664, 384, 384, 'BF16_BIAS', 1),    # cfg-2 proj: datt = dxm @ Wproj        || dWproj += dxm^T att      -> <0,3>
    (4104, 576, 192, 'F32', 1),          # cfg-5 (Bb=8 x 513 tokens, deit_tiny)
    (4104, 192, 768, 'DGELU', 1),
    (4104, 192, 192, 'BF16_BIAS', 1),
    (1672, 1152, 384, 'F32', 1),         # ragged last row tile / partial last k-tile of the wgrad (1672 = 26 * 64 + 8)
    (64, 384, 1536, 'DGELU', 26),        # the class-rows-only last block: 64 rows at pitch 26 * D (register-staged pair kernel, 32-row tiles)
    (64, 1536, 384, 'F32', 26),
    (64, 384, 384, 'BF16_BIAS', 26),
    (208, 576, 192, 'F32', 1),           # cfg-1-sized (Bb = 8): 32 / 64-row register-staged pair kernels
])
def test_gemm_pair_dgrad_and_wgrad_tight(rows, O, I, epi, ld_scale):
    """s3d_gemm_pair -- the launcher every Linear backward of s3d_block_bwd goes through -- at the shapes the benched configurations
    dispatch (cfg-2: 1664 token rows -> gemm_pair_dmat_kernel; class rows / small batches -> gemm_pair_kernel): both halves against
    fp64 products of the SAME bf16-rounded operands, so the only admissible error is fp32 accumulation order (F32 epilogue, wgrad:
    rel 2e-5) or the final bf16 rounding of the output (DGELU / BF16_BIAS epilogues: within one bf16 ulp of the fp64 value)."""
    g = torch.Generator().manual_seed(100 + rows + O)
    ld_dy, ld_x, ld_out = O * ld_scale, I * ld_scale, I * ld_scale
    dy = torch.zeros(rows, ld_dy); dy[:, :O] = torch.randn(rows, O, generator=g)
    x = torch.zeros(rows, ld_x); x[:, :I] = torch.randn(rows, I, generator=g)
    w = torch.randn(O, I, generator=g) * 0.05                       # nn.Linear weight [out][in]: the dgrad's k-major B operand
    pre = torch.zeros(rows, ld_out); pre[:, :I] = torch.randn(rows, I, generator=g)
    dyh = dy.to(DEV).to(torch.bfloat16); xh = x.to(DEV).to(torch.bfloat16); wh = w.to(DEV).to(torch.bfloat16)
    aux = pre.to(DEV).to(torch.bfloat16)
    dW = torch.zeros(O, I, dtype=torch.float32, device=DEV); db = torch.zeros(O, dtype=torch.float32, device=DEV)
    dg = L.fill(L.S3dGemmArgs(), A_hi=dyh, lda=ld_dy, B_hi=wh, ldb=I, M=rows, N=I, K=O, alpha=1.0)
    out32 = out16 = None
    if epi == 'F32':
        out32 = torch.full((rows, ld_out), float('nan'), dtype=torch.float32, device=DEV)
        L.fill(dg, C=out32, ldc=ld_out)
    else:
        out16 = torch.full((rows, ld_out), float('nan'), dtype=torch.bfloat16, device=DEV)
        L.fill(dg, O_hi=out16, ldo=ld_out)
        if epi == 'DGELU':
            L.fill(dg, aux=aux, ldaux=ld_out)
    wg = L.fill(L.S3dGemmArgs(), A_hi=dyh, lda=ld_dy, B_hi=xh, ldb=ld_x, M=O, N=I, K=rows, C=dW, ldc=I, bias_grad=db, alpha=1.0)
    for rep in (1, 2):                                               # the wgrad half accumulates, the dgrad half overwrites
        L.check(L.lib().s3d_gemm_pair(ops.EPI[epi], ctypes.byref(dg), ctypes.byref(wg), L.current_stream()), 'gemm_pair')
        ref_w = rep * (dyh[:, :O].double().t() @ xh[:, :I].double())
        assert rel_err(dW, ref_w) < 2e-5, f'wgrad half (pass {rep})'
        assert rel_err(db, rep * dyh[:, :O].double().sum(0)) < 2e-5, f'bias gradient (pass {rep})'
        ref = dyh[:, :O].double() @ wh.double()
        if epi == 'F32':
            assert rel_err(out32[:, :I], ref) < 1e-5, 'dgrad half'
        else:
            if epi == 'DGELU':
                ref = ref * _gelu_grad(aux[:, :I])
            got = out16[:, :I].double()
            assert not torch.isnan(got).any()
            ulp = ref.abs() * 2.0 ** -8 + 1e-6 * float(ref.abs().max())          # half-ulp rounding + the kernel's fast erf / exp (1e-6)
            assert bool(((got - ref).abs() <= ulp).all()), f'dgrad half: max {(got - ref).abs().max():.3e}'
            assert rms_err(got, ref) < 3e-3                                      # uniform rounding noise: 2^-9 / sqrt(3)
        if ld_scale > 1:                                                         # nothing outside the addressed columns was touched
            rest = (out32 if out32 is not None else out16)[:, I:]
            assert bool(torch.isnan(rest.float()).all())


def test_gemm_pair_with_a_second_wgrad_riding_on_the_launch():
    """s3d_gemm_pair3: the qkv pair launch of the fused backward carries attn.proj's wgrad as a third problem (cfg-2 shapes: 1664 rows) --
    all three results against fp64 products of the same bf16-rounded operands."""
    g = torch.Generator().manual_seed(77)
    rows, D = 1664, 384
    dqkv = torch.randn(rows, 3 * D, generator=g).to(DEV).to(torch.bfloat16)
    xn1 = torch.randn(rows, D, generator=g).to(DEV).to(torch.bfloat16)
    wqkv = (torch.randn(3 * D, D, generator=g) * 0.05).to(DEV).to(torch.bfloat16)
    dxm = torch.randn(rows, D, generator=g).to(DEV).to(torch.bfloat16)
    att = torch.randn(rows, D, generator=g).to(DEV).to(torch.bfloat16)
    dxn = torch.full((rows, D), float('nan'), dtype=torch.float32, device=DEV)
    dWq = torch.zeros(3 * D, D, dtype=torch.float32, device=DEV); dbq = torch.zeros(3 * D, dtype=torch.float32, device=DEV)
    dWp = torch.zeros(D, D, dtype=torch.float32, device=DEV); dbp = torch.zeros(D, dtype=torch.float32, device=DEV)
    dg = L.fill(L.S3dGemmArgs(), A_hi=dqkv, lda=3 * D, B_hi=wqkv, ldb=D, M=rows, N=D, K=3 * D, C=dxn, ldc=D, alpha=1.0)
    wg = L.fill(L.S3dGemmArgs(), A_hi=dqkv, lda=3 * D, B_hi=xn1, ldb=D, M=3 * D, N=D, K=rows, C=dWq, ldc=D, bias_grad=dbq, alpha=1.0)
    wg2 = L.fill(L.S3dGemmArgs(), A_hi=dxm, lda=D, B_hi=att, ldb=D, M=D, N=D, K=rows, C=dWp, ldc=D, bias_grad=dbp, alpha=1.0)
    L.check(L.lib().s3d_gemm_pair3(ops.EPI['F32'], ctypes.byref(dg), ctypes.byref(wg), ctypes.byref(wg2), L.current_stream()), 'gemm_pair3')
    assert rel_err(dxn, dqkv.double() @ wqkv.double()) < 1e-5
    assert rel_err(dWq, dqkv.double().t() @ xn1.double()) < 2e-5 and rel_err(dbq, dqkv.double().sum(0)) < 2e-5
    assert rel_err(dWp, dxm.double().t() @ att.double()) < 2e-5 and rel_err(dbp, dxm.double().sum(0)) < 2e-5


@pytest.mark.parametrize('rows,K,N,want', [
    (1664, 1536, 384, 3), (1664, 1152, 384, 3), (1664, 1152, 384, 4), (1664, 1536, 384, 2), (1664, 1536, 384, 1),   # cfg-2: fc1 / qkv dgrads
    (208, 576, 192, 3),                   # cfg-1-sized, deit_tiny
    (78, 392, 136, 3),                    # ragged: 78 rows, K not a multiple of 64 (partial last k-tile), N not a multiple of 64
    (1000, 72, 64, 4),                    # fewer slices than asked for (two 64-tiles)
])
def test_gemm_dgrad_splitk_planes_and_their_sum_in_layernorm_bwd(rows, K, N, want):
    """s3d_gemm_dgrad_splitk: the k-slices are STORED as partial planes (no atomics) and s3d_layernorm_bwd adds them (dy_parts); each plane
    against the fp64 product of the same bf16-rounded operands over its k range, the LayerNorm backward against the one fed the summed dy."""
    g = torch.Generator().manual_seed(500 + rows + K)
    dy = torch.randn(rows, K, generator=g).to(DEV).to(torch.bfloat16)
    w = (torch.randn(K, N, generator=g) * 0.05).to(DEV).to(torch.bfloat16)       # nn.Linear weight [out = K][in = N], read k-major
    lib = L.lib()
    ns = lib.s3d_gemm_dgrad_splitk_slices(K, want)
    assert 1 <= ns <= want
    planes = torch.full((ns + 1, rows, N), float('nan'), dtype=torch.float32, device=DEV)
    ga = L.fill(L.S3dGemmArgs(), A_hi=dy, lda=K, B_hi=w, ldb=N, M=rows, N=N, K=K, C=planes, ldc=N, alpha=1.0)
    for rep in range(2):                                                             # overwrites: the second call gives the same planes
        L.check(lib.s3d_gemm_dgrad_splitk(ctypes.byref(ga), ns, ctypes.c_long(rows * N), L.current_stream()), 'dgrad_splitk')
    kchunk = ((K + ns - 1) // ns + 63) // 64 * 64
    for sl in range(ns):
        k0, k1 = sl * kchunk, min(K, (sl + 1) * kchunk)
        ref = dy[:, k0:k1].double() @ w[k0:k1].double()
        assert rel_err(planes[sl], ref) < 1e-5, f'plane {sl}'
    assert bool(torch.isnan(planes[ns]).all()), 'nothing behind the last plane is touched'
    assert rel_err(planes[:ns].double().sum(0), dy.double() @ w.double()) < 1e-5
    if N % 4 == 0 and N <= 1024:
        x = torch.randn(rows, N, generator=g).to(DEV)
        gamma = (1 + 0.1 * torch.randn(N, generator=g)).to(DEV)
        mean = x.mean(1); rstd = (x.var(1, unbiased=False) + 1e-6).rsqrt()
        dres = torch.randn(rows, N, generator=g).to(DEV)
        want_out = ops.layernorm_bwd(planes[:ns].sum(0).contiguous(), x, mean, rstd, gamma, dres=dres)
        dx = torch.empty_like(x); dx_bf = torch.empty(rows, N, dtype=torch.bfloat16, device=DEV)
        dg = torch.zeros(N, device=DEV); db = torch.zeros(N, device=DEV)
        a = L.fill(L.S3dLnBwdArgs(), dy=planes, lddy=N, dy_parts=ns, dy_part_stride=rows * N, x=x, ldx=N, mean=mean, rstd=rstd, gamma=gamma,
                   dres=dres, lddres=N, dx=dx, lddx=N, dx_bf=dx_bf, lddxbf=N, dgamma=dg, dbeta=db, rows=rows, D=N)
        L.check(lib.s3d_layernorm_bwd(ctypes.byref(a), L.current_stream()), 'layernorm_bwd (planes)')
        assert rel_err(dx, want_out[0]) < 2e-6 and rel_err(dg, want_out[2]) < 2e-5 and rel_err(db, want_out[3]) < 2e-5
        assert rel_err(dx_bf.float(), want_out[1].float()) < 1e-2


@pytest.mark.parametrize('rows,D,Hd', [(1664, 384, 1536), (208, 192, 768), (78, 136, 72)])
def test_fused_layernorm_backward_chain_row_statistics(rows, D, Hd):
    """The LayerNorm backward as a dgrad epilogue (s3d_ln_aux -> s3d_gemm_dgrad_dgelu -> s3d_gemm_dgrad_lnbwd): mlp.fc2's dgrad writes
    dh = bf16(dy @ W2 * gelu'(hpre)) and accumulates the two per-row dots, mlp.fc1's dgrad applies norm2's backward to its own output.  Checked
    against the composition the stand-alone kernels compute: fp64 products of the same bf16 operands -> nn.LayerNorm backward in fp64."""
    g = torch.Generator().manual_seed(700 + rows)
    lib = L.lib()
    x_mid = torch.randn(rows, D, generator=g)                                        # LayerNorm-2 input
    gamma = 1 + 0.2 * torch.randn(D, generator=g); beta = 0.1 * torch.randn(D, generator=g)
    W1 = torch.randn(Hd, D, generator=g) * 0.05; b1 = 0.1 * torch.randn(Hd, generator=g)
    W2 = torch.randn(D, Hd, generator=g) * 0.05
    dxo = torch.randn(rows, D, generator=g)                                          # d(x_out)
    dres = torch.randn(rows, D, generator=g)
    mean = x_mid.mean(1); rstd = (x_mid.var(1, unbiased=False) + 1e-6).rsqrt()
    xh = (x_mid - mean[:, None]) * rstd[:, None]
    w1h, w1l = ops.split_bf16(W1.to(DEV))
    pre = ((xh * gamma + beta).double() @ (w1h.double() + w1l.double()).cpu().t() + b1.double())       # what the split-bf16 forward computes
    hpre = pre.float().to(DEV).to(torch.bfloat16)
    w2h = W2.to(DEV).to(torch.bfloat16); dxo_h = dxo.to(DEV).to(torch.bfloat16)
    keep = []                                                                        # device copies must outlive the launches that read them
    def dev(t):
        keep.append(t.to(DEV).contiguous())
        return keep[-1]
    u = torch.empty(Hd, device=DEV); c = torch.empty(Hd, device=DEV)
    lay = (L.S3dLnAuxLayer * 1)()
    L.fill(lay[0], w_hi=w1h, w_lo=w1l, bias=dev(b1), gamma=dev(gamma), beta=dev(beta), u=u, c=c, K=Hd)
    L.check(lib.s3d_ln_aux(lay, 1, D, L.current_stream()), 'ln_aux')
    assert rel_err(u, (w1h.double().cpu() * gamma.double()).mean(1)) < 1e-5
    assert rel_err(c, b1.double() + (w1h.double() + w1l.double()).cpu() @ beta.double()) < 1e-5
    # producer: fc2 dgrad * gelu' + row statistics (and it clears a buffer that is idle)
    rs = torch.zeros(2, rows, device=DEV); idle = torch.full((300,), 7.0, device=DEV)
    dh = torch.full((rows, Hd), float('nan'), dtype=torch.bfloat16, device=DEV)
    ga = L.fill(L.S3dGemmArgs(), A_hi=dxo_h, lda=D, B_hi=w2h, ldb=Hd, M=rows, N=Hd, K=D, alpha=1.0, aux=hpre, ldaux=Hd, O_hi=dh, ldo=Hd)
    st = L.fill(L.S3dRowStats(), u=u, c=c, rs1=rs[0], rs2=rs[1], zero_buf=idle, zero_n=256)
    L.check(lib.s3d_gemm_dgrad_dgelu(ctypes.byref(ga), ctypes.byref(st), L.current_stream()), 'dgrad_dgelu')
    ref_dh = (dxo_h.double() @ w2h.double()) * _gelu_grad(hpre)
    assert rms_err(dh.double(), ref_dh) < 3e-3 and not torch.isnan(dh.float()).any()
    assert bool((idle[:256] == 0).all()) and bool((idle[256:] == 7.0).all())
    gdy = dh.double() @ w1h.double()                                                 # dy of the LayerNorm: what the fc1 dgrad computes from the ROUNDED dh
    gg = gdy.cpu() * gamma.double()
    s1_ref, s2_ref = gg.mean(1), (gg * xh.double()).mean(1)
    scale = float(gg.abs().mean())
    assert float((rs[0].double().cpu() - s1_ref).abs().max()) < 2e-4 * scale, 's1'
    # s2: the kernel's arithmetic exactly (the dot of the rounded dh with the bf16 pre-activation minus c) ...
    s2_formula = (dh.double() * (hpre.double() - c.double())).sum(1).cpu() / D
    assert float((rs[1].double().cpu() / D - s2_formula).abs().max()) < 2e-5 * float(s2_formula.abs().max() + scale), 's2 arithmetic'
    # ... and the identity it stands for, up to the bf16 rounding of the saved pre-activation / the weight's low plane: ~2^-9 sqrt(K) / D of
    # |dh| |z| per row (1e-4 of |dy gamma| at cfg-2's K = 1536, D = 384; the three-tile shapes here average less)
    assert float((rs[1].double().cpu() / D - s2_ref).abs().max()) < (1e-3 if Hd >= 512 else 4e-3) * scale, 's2'
    # consumer: fc1 dgrad + norm2 backward epilogue
    dx = torch.full((rows, D), float('nan'), device=DEV); dx_bf = torch.full((rows, D), float('nan'), dtype=torch.bfloat16, device=DEV)
    nty = (rows + 63) // 64
    part = torch.full((nty + 1, 2, D), float('nan'), device=DEV)
    gb = L.fill(L.S3dGemmArgs(), A_hi=dh, lda=Hd, B_hi=w1h, ldb=D, M=rows, N=D, K=Hd, alpha=1.0)
    ln = L.fill(L.S3dLnBwdArgs(), x=dev(x_mid), ldx=D, mean=dev(mean), rstd=dev(rstd), gamma=dev(gamma), dres=dev(dres), lddres=D, dx=dx, lddx=D,
                dx_bf=dx_bf, lddxbf=D, rows=rows, D=D, partial=part, partial_blocks=nty)
    st2 = L.fill(L.S3dRowStats(), rs1=rs[0], rs2=rs[1])
    L.check(lib.s3d_gemm_dgrad_lnbwd(ctypes.byref(gb), ctypes.byref(ln), ctypes.byref(st2), L.current_stream()), 'dgrad_lnbwd')
    rstd_d, xh_d = rstd.double(), xh.double()
    ref_dx = rstd_d[:, None] * (gg - s1_ref[:, None] - xh_d * s2_ref[:, None]) + dres.double()
    assert rel_err(dx, ref_dx) < (5e-4 if Hd >= 512 else 4e-3), 'dx'    # (s2 from the bf16-rounded pre-activation: 2.2e-4 measured at 1664 x 384)
    ref_dx_k = rstd_d[:, None] * (gg - rs[0].double().cpu()[:, None] - xh_d * (rs[1].double().cpu() / D)[:, None]) + dres.double()
    assert rel_err(dx, ref_dx_k) < 2e-5, 'dx with the statistics the kernel was given'
    assert rms_err(dx_bf.double(), ref_dx) < 3e-3
    assert rel_err(part[:nty, 0].double().sum(0), (gdy.cpu() * xh_d).sum(0)) < 2e-5, 'dgamma partials'
    assert rel_err(part[:nty, 1].double().sum(0), gdy.cpu().sum(0)) < 2e-5, 'dbeta partials'
    assert bool(torch.isnan(part[nty]).all())
    # ... and with atomics instead of partial rows
    dg = torch.zeros(D, device=DEV); db = torch.zeros(D, device=DEV)
    L.fill(ln, partial=None, partial_blocks=0, dgamma=dg, dbeta=db)
    L.check(lib.s3d_gemm_dgrad_lnbwd(ctypes.byref(gb), ctypes.byref(ln), ctypes.byref(st2), L.current_stream()), 'dgrad_lnbwd (atomics)')
    assert rel_err(dg, (gdy.cpu() * xh_d).sum(0)) < 2e-5 and rel_err(db, gdy.cpu().sum(0)) < 2e-5


@pytest.mark.parametrize('rows,K', [(32896, 768), (16416, 576), (1000, 192), (61, 32)])
def test_dgrad_with_whole_row_layernorm_backward(rows, K):
    """s3d_gemm_dgrad_lnrows: dx = LayerNorm'(dy = A @ W) + dres for 192-wide layers on 64 x 192 tiles (whole rows per workgroup: the row
    statistics come from the tile itself) -- the point path's fc1 -> norm2 and qkv -> norm1 backward (models/3DViT/model.py:318-320).  Against
    fp64: the product of the same bf16 operands -> nn.LayerNorm backward; partial rows and atomics for dgamma / dbeta; ragged row counts."""
    D = 192
    g = torch.Generator().manual_seed(40 + rows)
    lib = L.lib()
    x = torch.randn(rows, D, generator=g) * 1.5 + 0.3
    gamma = 1 + 0.2 * torch.randn(D, generator=g)
    W = torch.randn(K, D, generator=g) * 0.05
    dz = torch.randn(rows, K, generator=g)
    dres = torch.randn(rows, D, generator=g)
    mean = x.mean(1); rstd = (x.var(1, unbiased=False) + 1e-6).rsqrt()
    keep = []
    def dev(t):
        keep.append(t.to(DEV).contiguous())
        return keep[-1]
    wh = dev(W).to(torch.bfloat16); dzh = dev(dz).to(torch.bfloat16)
    dx = torch.full((rows, D), float('nan'), device=DEV); dx_bf = torch.full((rows, D + 8), float('nan'), dtype=torch.bfloat16, device=DEV)
    nty = (rows + 63) // 64
    part = torch.full((nty + 1, 2, D), float('nan'), device=DEV)
    ga = L.fill(L.S3dGemmArgs(), A_hi=dzh, lda=K, B_hi=wh, ldb=D, M=rows, N=D, K=K, alpha=1.0)
    ln = L.fill(L.S3dLnBwdArgs(), x=dev(x), ldx=D, mean=dev(mean), rstd=dev(rstd), gamma=dev(gamma), dres=dev(dres), lddres=D, dx=dx, lddx=D,
                dx_bf=dx_bf, lddxbf=D + 8, rows=rows, D=D, partial=part, partial_blocks=nty)
    L.check(lib.s3d_gemm_dgrad_lnrows(ctypes.byref(ga), ctypes.byref(ln), L.current_stream()), 'dgrad_lnrows')
    dy = (dzh.double() @ wh.double()).cpu()
    xh = ((x - mean[:, None]) * rstd[:, None]).double()
    gg = dy * gamma.double()
    ref = rstd.double()[:, None] * (gg - gg.mean(1, keepdim=True) - xh * (gg * xh).mean(1, keepdim=True)) + dres.double()
    assert rel_err(dx, ref) < 2e-5, 'dx'
    assert rms_err(dx_bf[:, :D].double(), ref) < 3e-3 and bool(torch.isnan(dx_bf[:, D:].float()).all())
    assert rel_err(part[:nty, 0].double().sum(0), (dy * xh).sum(0)) < 2e-5, 'dgamma partials'
    assert rel_err(part[:nty, 1].double().sum(0), dy.sum(0)) < 2e-5, 'dbeta partials'
    assert bool(torch.isnan(part[nty]).all())
    dg = torch.zeros(D, device=DEV); db = torch.zeros(D, device=DEV)
    L.fill(ln, partial=None, partial_blocks=0, dgamma=dg, dbeta=db, dres=None)
    L.check(lib.s3d_gemm_dgrad_lnrows(ctypes.byref(ga), ctypes.byref(ln), L.current_stream()), 'dgrad_lnrows (atomics, no residual)')
    assert rel_err(dg, (dy * xh).sum(0)) < 2e-5 and rel_err(db, dy.sum(0)) < 2e-5
    assert rel_err(dx, ref - dres.double()) < 2e-5


@pytest.mark.parametrize('rows,shapes,acc', [
    (1664, [(384, 1536), (1536, 384), (384, 384), (1152, 384)] * 3, 1),       # cfg-2: three blocks' fc2 / fc1 / proj / qkv wgrads in one launch
    (1664, [(384, 1536), (1536, 384), (384, 384), (1152, 384)], 0),           # overwrite mode
    (208, [(192, 768), (768, 192), (192, 192), (576, 192)] * 2, 1),           # cfg-1-sized (partial last k-tile: 208 = 3 * 64 + 16)
    (78, [(136, 72), (72, 136), (8, 8), (264, 40)], 1),                       # ragged edges on every side, odd row count
    (1000, [(304, 320), (64, 192), (192, 256), (448, 128), (320, 72), (136, 200)], 1),   # 1.5 / 2.5 / 3.5 tiles per side, partial last k-tile
    (32896, [(192, 768), (768, 192), (192, 192), (576, 192)] * 2, 0),         # cfg-4: the point path's blocks (257-token sequences x 128 clouds)
    (16416 + 8, [(192, 768), (768, 192), (192, 192), (576, 192)] * 3, 1),     # cfg-5 rows + a k tail, accumulate
])
def test_gemm_wgrad_group_full_k_deterministic(rows, shapes, acc):
    """s3d_gemm_wgrad_group: dW_i (+)= dy_i^T x_i and db_i (+)= colsum(dy_i) for a list of layers in ONE launch, every output tile owned by one
    workgroup over the full k -- against fp64 products of the same bf16-rounded operands, twice (accumulate / overwrite semantics), and
    bitwise equal from run to run (no split-K, no atomics)."""
    g = torch.Generator().manual_seed(900 + rows + len(shapes))
    lib = L.lib()
    items = (L.S3dWgradItem * len(shapes))()
    keep, refs = [], []
    for i, (O, I) in enumerate(shapes):
        pad_o, pad_i = (8 if i % 2 else 0), (16 if i % 3 == 0 else 0)             # row pitches wider than the matrices
        dy = torch.zeros(rows, O + pad_o); dy[:, :O] = torch.randn(rows, O, generator=g)
        x = torch.zeros(rows, I + pad_i); x[:, :I] = torch.randn(rows, I, generator=g)
        dyh, xh = dy.to(DEV).to(torch.bfloat16), x.to(DEV).to(torch.bfloat16)
        dW = torch.full((O, I + 4), 0.5, dtype=torch.float32, device=DEV)         # pre-existing content: kept (accumulate) or replaced
        db = torch.full((O,), 0.25, dtype=torch.float32, device=DEV) if i % 4 != 2 else None
        L.fill(items[i], dy=dyh, ld_dy=O + pad_o, out=O, x=xh, ld_x=I + pad_i, **{'in': I}, dW=dW, ldw=I + 4, db=db)
        keep.append((dyh, xh, dW, db))
        refs.append((dyh[:, :O].double().t() @ xh[:, :I].double(), dyh[:, :O].double().sum(0)))
    snaps = []
    for rep in (1, 2):
        L.check(lib.s3d_gemm_wgrad_group(items, len(shapes), rows, ctypes.c_float(0.5), acc, L.current_stream()), 'wgrad_group')
        for (dyh, xh, dW, db), (rw, rb), (O, I) in zip(keep, refs, shapes):
            n = rep if acc else 1
            base = 0.5 if acc else 0.0
            assert rel_err(dW[:, :I] - base, 0.5 * n * rw) < 2e-5, f'dW pass {rep}'
            assert bool((dW[:, I:] == 0.5).all()), 'pad columns of dW untouched'
            if db is not None:
                assert rel_err(db - (0.25 if acc else 0.0), 0.5 * n * rb) < 2e-5, f'db pass {rep}'
        snaps.append([t.clone() for k in keep for t in k[2:] if t is not None])
    if not acc:                                                                    # overwrite mode: two runs, bitwise the same result
        assert all(torch.equal(a, b) for a, b in zip(*snaps))


@pytest.mark.parametrize('M,N,K', [(65536, 64, 32), (70000, 128, 64), (66048, 256, 128), (300000, 64, 64), (66000, 64, 40), (66000, 192, 192), (33000, 192, 96)])   # k = 40: register-staged kernel
def test_gemm_column_sums_for_the_following_batchnorm(M, N, K):
    """S3dGemmArgs::col_sums: the F32 epilogue of a point-path convolution also accumulates sum(y) and sum(y^2) per output channel
    (fp64 atomics), i.e. the batch statistics of the BatchNorm that follows; the GEMM output itself is unchanged."""
    g = torch.Generator().manual_seed(31)
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.2).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    xh, xl = ops.split_bf16(x); wh, wl = ops.split_bf16(w)
    assert L.lib().s3d_gemm_col_sums_ok(1, M, N) == 1
    y0 = torch.empty(M, N, device=DEV); y1 = torch.empty(M, N, device=DEV)
    sums = torch.zeros(2 * N, dtype=torch.float64, device=DEV)
    ops.gemm(0, 0, 1, 'F32', A_hi=xh, A_lo=xl, lda=K, B_hi=wh, B_lo=wl, ldb=K, M=M, N=N, K=K, bias=b, C=y0, ldc=N)
    ops.gemm(0, 0, 1, 'F32', A_hi=xh, A_lo=xl, lda=K, B_hi=wh, B_lo=wl, ldb=K, M=M, N=N, K=K, bias=b, C=y1, ldc=N, col_sums=sums)
    assert torch.equal(y0, y1)
    yd = y1.double()
    assert rel_err(sums[:N], yd.sum(0)) < 1e-6 and rel_err(sums[N:], (yd * yd).sum(0)) < 1e-6
    # small problems do not take the 128x128 kernels: the library says so instead of silently skipping the sums
    assert L.lib().s3d_gemm_col_sums_ok(1, 1664, 384) == 0


@pytest.mark.parametrize('M,N,K,sums', [(40000, 96, 96, True), (33001, 96, 48, False), (32768 + 5, 192, 96, False), (50000, 48, 48, True),
                                        (1 << 20, 96, 96, True), (33000, 64, 40, True), (32768, 128, 8, False), (65536 + 17, 192, 96, True),
                                        (131072, 48, 24, False)])
def test_gemm_rowstream_convolution(M, N, K, sums):
    """rowstream_gemm.hip: the point path's 1x1 convolutions / per-point projections (pointnet_util.py:238-241 at >= 32768 rows, K <= 96,
    N <= 192) -- every wave multiplies 16-row chunks straight from global memory against weight planes staged in LDS.  Against the fp64
    product of the same split operands; ragged M, K not a multiple of 32, two column passes, the BatchNorm column sums."""
    g = torch.Generator().manual_seed(M % 977 + N + K)
    x = (torch.randn(M, K, generator=g) * 1.3 + 0.2).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.2).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    xh, xl = ops.split_bf16(x); wh, wl = ops.split_bf16(w)
    ldc = N + 4                                                # a padded output pitch: nothing may land in the pad columns
    y = torch.full((M, ldc), 7.0, device=DEV)
    s = torch.zeros(2 * N, dtype=torch.float64, device=DEV) if sums else None
    kw = dict(col_sums=s) if sums else {}
    ops.gemm(0, 0, 1, 'F32', A_hi=xh, A_lo=xl, lda=K, B_hi=wh, B_lo=wl, ldb=K, M=M, N=N, K=K, bias=b, C=y, ldc=ldc, alpha=0.5, **kw)
    ref = 0.5 * ((xh.double() + xl.double()) @ (wh.double() + wl.double()).t()) + b.double()
    assert rel_err(y[:, :N], ref) < 1e-5
    assert float((y[:, N:] - 7.0).abs().max()) == 0.0
    if sums:
        yd = y[:, :N].double()
        assert rel_err(s[:N], yd.sum(0)) < 1e-6 and rel_err(s[N:], (yd * yd).sum(0)) < 1e-6


@pytest.mark.parametrize('groups,ntok,D', [(64, 26, 384), (5, 3, 40), (700, 15, 192)])
def test_token_gradients_default_and_deterministic(groups, ntok, D):
    """s3d_token_grads: d(pos_embed)[t] = sum over groups of dx[g, t], d(cls) = the token-0 sum, d(conv bias) = the sum over the other
    tokens (vit_3d_2d_pretrain.py:455-470 backward) -- the atomic default and the single-writer deterministic kernel against fp64, both
    accumulating onto existing content; the deterministic one bitwise equal from run to run."""
    g = torch.Generator().manual_seed(groups + D)
    dx = torch.randn(groups, ntok, D, generator=g)
    dxd = dx.to(DEV)
    ref = dx.double().sum(0)
    lib = L.lib()
    outs = []
    for det in (0, 1, 1):
        lib.s3d_set_deterministic(det)
        try:
            dpos = torch.full((ntok, D), 0.5, device=DEV); dcls = torch.full((D,), 0.25, device=DEV); dbias = torch.full((D,), -1.0, device=DEV)
            a = L.fill(L.S3dPosGradArgs(), dx=dxd, groups=groups, ntok=ntok, D=D, dpos=dpos, dcls=dcls, dbias=dbias)
            L.check(lib.s3d_token_grads(ctypes.byref(a), L.current_stream()), 'token grads')
            torch.cuda.synchronize()
        finally:
            lib.s3d_set_deterministic(0)
        assert rel_err(dpos - 0.5, ref) < 2e-6 and rel_err(dcls - 0.25, ref[0]) < 2e-6
        assert rel_err(dbias + 1.0, ref[1:].sum(0)) < 2e-6
        outs.append((dpos.clone(), dcls.clone(), dbias.clone()))
    assert all(torch.equal(a, b) for a, b in zip(outs[1], outs[2]))


def test_gemm_wgrad_into_a_sub_matrix():
    """The factored set-abstraction convolution accumulates d(Wf) INSIDE the conv weight's gradient: C = dW + 3 columns, ldc = 3 + I."""
    g = torch.Generator().manual_seed(13)
    rows, O, I = 8192, 96, 48
    dy = torch.randn(rows, O, generator=g).to(DEV); x = torch.randn(rows, I, generator=g).to(DEV)
    dyh, _ = ops.split_bf16(dy); xh, _ = ops.split_bf16(x)
    dW = torch.zeros(O, 3 + I, dtype=torch.float32, device=DEV)
    ops.gemm(1, 1, 0, 'ATOMIC', splitk=0, A_hi=dyh, lda=O, B_hi=xh, ldb=I, M=O, N=I, K=rows, C=dW[:, 3:], ldc=3 + I)
    assert rel_err(dW[:, 3:], dyh.double().t() @ xh.double()) < 2e-5 and float(dW[:, :3].abs().max()) == 0.0


@pytest.mark.parametrize('M,N,K', [(130, 192, 128),
                                   (33003, 768, 512),       # 128x256 sixteen-wave forward tiles / 256x128 eight-wave dgrad tiles
                                   (33003, 768, 768)])      # the encoder layer's linear1 / linear2 at cfg-3 (k = D = 768)
def test_gemm_epilogues_gelu_resid_token_dgelu(M, N, K):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.1).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    xh, xl = ops.split_bf16(x)
    wh, wl = ops.split_bf16(w)
    pre_ref = x.double() @ w.double().t() + b.double()
    # GELU: aux = bf16(pre), O = split(gelu(pre))
    aux = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    oh = torch.zeros_like(aux); ol = torch.zeros_like(aux)
    ops.gemm(0, 0, 1, 'GELU', A_hi=xh, A_lo=xl, lda=K, B_hi=wh, B_lo=wl, ldb=K, M=M, N=N, K=K, bias=b, aux=aux, ldaux=N,
             O_hi=oh, O_lo=ol, ldo=N)
    assert rel_err(oh.float() + ol.float(), F.gelu(pre_ref)) < 5e-5
    assert rel_err(aux.float(), pre_ref) < 1e-2
    # RESID
    R = torch.randn(M, N, generator=g).to(DEV)
    C = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(0, 0, 1, 'RESID', A_hi=xh, A_lo=xl, lda=K, B_hi=wh, B_lo=wl, ldb=K, M=M, N=N, K=K, bias=b, R=R, ldr=N, C=C, ldc=N)
    assert rel_err(C, pre_ref + R.double()) < 2e-5
    # TOKEN (ntok = 10 -> rows 0,10,20,... take cls instead of bias, all rows add pos[t])
    ntok = 10
    cls = torch.randn(N, generator=g).to(DEV); pos = torch.randn(ntok, N, generator=g).to(DEV)
    ops.gemm(0, 0, 1, 'TOKEN', A_hi=xh, A_lo=xl, lda=K, B_hi=wh, B_lo=wl, ldb=K, M=M, N=N, K=K, bias=b, C=C, ldc=N, alpha=0.2,
             cls=cls, pos=pos, ntok=ntok)
    t = torch.arange(M) % ntok
    ref = (x.double() @ w.double().t()) * 0.2 + torch.where((t == 0)[:, None], cls.double().cpu(), b.double().cpu()).to(DEV) \
        + pos.double()[t.to(DEV)]
    assert rel_err(C, ref) < 2e-5
    # DGELU on the NN path: O = bf16(acc * gelu'(aux))
    dy = torch.randn(M, K, generator=g).to(DEV)
    w2 = (torch.randn(K, N, generator=g) * 0.1).to(DEV)
    dyh, _ = ops.split_bf16(dy); w2h, _ = ops.split_bf16(w2)
    o = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm(0, 1, 0, 'DGELU', A_hi=dyh, lda=K, B_hi=w2h, ldb=N, M=M, N=N, K=K, aux=aux, ldaux=N, O_hi=o, ldo=N)
    a = aux.double().requires_grad_(True)
    gp = torch.autograd.grad(F.gelu(a).sum(), a)[0]
    assert rms_err(o.float(), (dyh.double() @ w2h.double()) * gp) < 1e-2
    # RELU (linear1 of the group_embed encoder layer, vit_3d_2d_pretrain.py:381): aux = bf16(pre), O = split(relu(pre)); DRELU: O = bf16(acc * (aux > 0))
    ops.gemm(0, 0, 1, 'RELU', A_hi=xh, A_lo=xl, lda=K, B_hi=wh, B_lo=wl, ldb=K, M=M, N=N, K=K, bias=b, aux=aux, ldaux=N, O_hi=oh, O_lo=ol, ldo=N)
    assert rel_err(oh.float() + ol.float(), F.relu(pre_ref)) < 5e-5
    assert rel_err(aux.float(), pre_ref) < 1e-2
    ops.gemm(0, 1, 0, 'DRELU', A_hi=dyh, lda=K, B_hi=w2h, ldb=N, M=M, N=N, K=K, aux=aux, ldaux=N, O_hi=o, ldo=N)
    assert rms_err(o.float(), (dyh.double() @ w2h.double()) * (aux.double() > 0)) < 1e-2


@pytest.mark.parametrize('M,N,K,epi', [(14200, 2304, 768, 'BF16_BIAS'),      # 56 x 9 tiles of 256x256 (ragged last row tile)
                                       (10700, 3072, 768, 'GELU'),           # 42 x 12
                                       (43500, 768, 768, 'RESID'),           # 170 x 3: attn.proj
                                       (43520, 768, 3072, 'RESID')])         # mlp.fc2 (96 k-steps)
def test_gemm_fat_forward_tile(M, N, K, epi):
    """gemm_nt_fat_kernel (256x256 split-bf16 tile, eight waves; what the pass-1 Linear layers of cfg-3 dispatch): every output against
    the fp64 product of the SAME operand planes (hi + lo, minus the lo x lo term the three-MFMA product drops), every epilogue plane."""
    g = torch.Generator().manual_seed(M + N)
    ah, al = ops.split_bf16(torch.randn(M, K, generator=g).to(DEV))
    bh, bl = ops.split_bf16((torch.randn(N, K, generator=g) * K ** -0.5).to(DEV))
    bias = torch.randn(N, generator=g).to(DEV)
    R = torch.randn(M, N, generator=g).to(DEV)
    C = torch.full((M, N), float('nan'), dtype=torch.float32, device=DEV)
    oh = torch.full((M, N), float('nan'), dtype=torch.bfloat16, device=DEV); ol = oh.clone(); aux = oh.clone()
    L.lib().s3d_cov_enable(1)                                                   # (restarts the fixture's record: this launch only)
    ops.gemm(0, 0, 1, epi, A_hi=ah, A_lo=al, lda=K, B_hi=bh, B_lo=bl, ldb=K, M=M, N=N, K=K, bias=bias, R=R, ldr=N, C=C, ldc=N,
             O_hi=oh, O_lo=ol, ldo=N, aux=aux, ldaux=N)
    from tests import _cov
    assert any(k.startswith('gemm_nt_fat:') for k in _cov.collect(L.lib())), 'the shape did not dispatch the 256x256 tile'
    ref = (ah.double() + al.double()) @ (bh.double() + bl.double()).t() - al.double() @ bl.double().t() + bias.double()
    if epi == 'RESID':
        assert rel_err(C, ref + R.double()) < 3e-6                              # fp32 accumulation over k = 768 .. 3072
        assert rel_err(oh.float() + ol.float(), C.double()) < 2e-5              # the optional split copy of the residual stream
    elif epi == 'BF16_BIAS':
        assert rel_err(oh.float() + ol.float(), ref) < 2e-5
    else:
        assert rel_err(oh.float() + ol.float(), F.gelu(ref)) < 2e-5
        assert float((aux.double() - ref).abs().max()) <= float(ref.abs().max()) * 2.0 ** -8
    del ref


@pytest.mark.parametrize('M,N,K,epi', [(43500, 768, 2304, 'F32'),            # qkv dgrad (ragged last row tile)
                                       (10700, 3072, 768, 'DGELU'),          # mlp.fc2 dgrad
                                       (43520, 768, 768, 'BF16_BIAS')])      # attn.proj dgrad
def test_gemm_fat_dgrad_tile(M, N, K, epi):
    """gemm_nn_fat_kernel (256x256 plain-bf16 dgrad tile; the long backward Linear layers of cfg-3): against the fp64 product of the SAME
    bf16 operands -- fp32 accumulation order for the F32 epilogue, one bf16 ulp for the bf16 outputs."""
    from tests import _cov
    g = torch.Generator().manual_seed(M + K)
    dy = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
    w = (torch.randn(K, N, generator=g) * K ** -0.5).to(DEV).to(torch.bfloat16)         # nn.Linear weight [out = K][in = N]: k-major B
    aux = torch.randn(M, N, generator=g).to(DEV).to(torch.bfloat16)
    out32 = torch.full((M, N), float('nan'), dtype=torch.float32, device=DEV)
    out16 = torch.full((M, N), float('nan'), dtype=torch.bfloat16, device=DEV)
    kw = dict(A_hi=dy, lda=K, B_hi=w, ldb=N, M=M, N=N, K=K)
    if epi == 'F32':
        kw.update(C=out32, ldc=N)
    else:
        kw.update(O_hi=out16, ldo=N)
        if epi == 'DGELU':
            kw.update(aux=aux, ldaux=N)
    L.lib().s3d_cov_enable(1)
    ops.gemm(0, 1, 0, epi, **kw)
    assert any(k.startswith('gemm_nn_fat:') for k in _cov.collect(L.lib())), 'the shape did not dispatch the 256x256 tile'
    ref = dy.double() @ w.double()
    if epi == 'F32':
        assert rel_err(out32, ref) < 1e-5
    else:
        if epi == 'DGELU':
            ref = ref * _gelu_grad(aux)
        got = out16.double()
        assert not torch.isnan(got).any()
        ulp = ref.abs() * 2.0 ** -8 + 1e-6 * float(ref.abs().max())
        assert bool(((got - ref).abs() <= ulp).all()), f'max {(got - ref).abs().max():.3e}'


@pytest.mark.parametrize('rows,D', [(7, 192), (1664, 384), (333, 768), (40000, 192), (9, 256), (64, 1024), (5000, 512)])
def test_layernorm_fwd_bwd(rows, D):
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(rows, D, generator=g) * 2 + 0.5).to(DEV)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(D, generator=g)).to(DEV)
    out, hi, lo, mean, rstd = ops.layernorm_fwd(x, gamma, beta)
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = F.layer_norm(xr, (D,), gr, br, 1e-6)
    assert rel_err(out, ref.detach()) < 1e-5
    assert rel_err(hi.float() + lo.float(), ref.detach()) < 5e-5
    dy = torch.randn(rows, D, generator=g).to(DEV)
    dres = torch.randn(rows, D, generator=g).to(DEV)
    dx, dx_bf, dg, db = ops.layernorm_bwd(dy, x, mean, rstd, gamma, dres)
    ref.backward(dy.double())
    assert rel_err(dx, xr.grad + dres.double()) < 2e-5
    assert rel_err(dx_bf.float(), xr.grad + dres.double()) < 1e-2
    assert rel_err(dg, gr.grad) < 1e-4 and rel_err(db, br.grad) < 1e-4
    # partial-sum mode: per-workgroup column sums, then one reduce launch that ADDS them to dgamma / dbeta
    import ctypes
    from simple3d_former_amd import _lib as L
    nblk = 13
    part = torch.full((nblk, 2, D), float('nan'), device=DEV)                  # every slot must be written
    dg2 = torch.full((D,), 0.5, device=DEV); db2 = torch.full((D,), -0.25, device=DEV)
    dx2 = torch.empty_like(dx)
    a = L.fill(L.S3dLnBwdArgs(), dy=dy, lddy=D, x=x, ldx=D, mean=mean, rstd=rstd, gamma=gamma, dres=dres, lddres=D, dx=dx2,
               lddx=D, rows=rows, D=D, partial=part, partial_blocks=nblk)
    L.check(L.lib().s3d_layernorm_bwd(ctypes.byref(a), L.current_stream()), 'ln_bwd partial')
    P = (ctypes.c_void_p * 1)(part.data_ptr()); G = (ctypes.c_void_p * 1)(dg2.data_ptr()); Bp = (ctypes.c_void_p * 1)(db2.data_ptr())
    L.check(L.lib().s3d_layernorm_grad_reduce(P, G, Bp, 1, nblk, D, L.current_stream()), 'ln_grad_reduce')
    assert torch.equal(dx2, dx)
    assert rel_err(dg2 - 0.5, gr.grad) < 1e-4 and rel_err(db2 + 0.25, br.grad) < 1e-4


def _attn_ref(q, k, v):
    s = (q @ k.transpose(-2, -1)) * q.shape[-1] ** -0.5
    return s.softmax(-1) @ v


@pytest.mark.parametrize('Bb,H,N,hd,seq_first', [(4, 6, 26, 64, False), (3, 3, 15, 256, False), (2, 3, 197, 256, False),
                                                (5, 4, 100, 192, True), (2, 3, 257, 64, False), (64, 6, 26, 64, False),
                                                # 9 / 17 tiles: five waves per workgroup (attention.hip: coop_five); 8 tiles: four
                                                (2, 3, 513, 64, False), (2, 3, 250, 64, False),
                                                (3, 4, 20, 96, True), (5, 4, 32, 48, False), (3, 2, 7, 64, False), (2, 2, 2, 64, False),
                                                # even batch, N <= 16, contiguous: two sequences share one 32-row tile (pack_pairs)
                                                (392, 3, 15, 256, False), (6, 6, 10, 64, False), (4, 4, 16, 48, False), (8, 3, 15, 192, False),
                                                # N <= 32 at hd = 192 / 256: the one-launch backward, gradients two d-blocks at a time
                                                (3, 4, 20, 192, True), (5, 3, 32, 256, False), (7, 3, 29, 256, False),
                                                # hd = 192 long sequences, batch-first and seq-first, ragged last tile, odd tile count: the pipelined forward and
                                                # the cooperative backward with uniform-base staging addresses
                                                (2, 4, 300, 192, False), (2, 4, 330, 192, True), (1, 4, 97, 192, False),
                                                # hd = 256, one tile: the forward that requests Q and K together (attn_fwd_tile256_kernel) -- two tokens,
                                                # odd lengths, a full tile, packed pairs of 16 (N = 32 with two segments), a workgroup with idle waves
                                                (2, 3, 2, 256, False), (3, 3, 7, 256, False), (5, 3, 16, 256, False), (4, 3, 16, 256, False), (1, 1, 31, 256, False)])
def test_attention_fwd_bwd(Bb, H, N, hd, seq_first):
    g = torch.Generator().manual_seed(6)
    D = H * hd
    rows = Bb * N
    qkv = torch.randn(rows, 3 * D, generator=g).to(DEV)
    sb, st = (1, Bb) if seq_first else (N, 1)
    hi, lo = ops.split_bf16(qkv)
    out_hi, out_lo, lse = ops.attention_fwd(hi, lo, Bb, H, N, D, sb, st, split=True)

    def to_bhnd(t, width):          # rows -> [Bb, H, N, hd] for block `which`
        t = t.view(N, Bb, -1).transpose(0, 1) if seq_first else t.view(Bb, N, -1)
        return t

    x = to_bhnd(qkv.double(), 3 * D).requires_grad_(True)
    q, k, v = [x[..., i * D:(i + 1) * D].reshape(Bb, N, H, hd).transpose(1, 2) for i in range(3)]
    ref = _attn_ref(q, k, v)                                   # [Bb,H,N,hd]
    got = to_bhnd(out_hi.float() + out_lo.float(), D).reshape(Bb, N, H, hd).transpose(1, 2)
    e = rel_err(got, ref.detach())
    assert e < 1e-4, f'fwd rel err {e:.3e}'
    # plain-bf16 forward as well
    o2, _, _ = ops.attention_fwd(hi, lo, Bb, H, N, D, sb, st, split=False)
    assert rel_err(to_bhnd(o2.float(), D).reshape(Bb, N, H, hd).transpose(1, 2), ref.detach()) < 3e-2
    # backward (plain bf16 on the hi planes): reference computed from the SAME bf16-rounded q,k,v,dO
    dout = torch.randn(rows, D, generator=g).to(DEV)
    dout_bf = dout.to(torch.bfloat16)
    dqkv = ops.attention_bwd(hi, out_hi, out_lo, lse, dout_bf, Bb, H, N, D, sb, st)
    xb = to_bhnd(hi.double(), 3 * D).requires_grad_(True)
    qb, kb, vb = [xb[..., i * D:(i + 1) * D].reshape(Bb, N, H, hd).transpose(1, 2) for i in range(3)]
    refb = _attn_ref(qb, kb, vb)
    do = to_bhnd(dout_bf.double(), D).reshape(Bb, N, H, hd).transpose(1, 2)
    refb.backward(do)
    gotd = to_bhnd(dqkv.float(), 3 * D)
    for i, name in enumerate('qkv'):
        e = rms_err(gotd[..., i * D:(i + 1) * D], xb.grad[..., i * D:(i + 1) * D])
        assert e < 2e-2, f'd{name} rms err {e:.3e}'


@pytest.mark.parametrize('Bb,H,N,hd,seq_first,stored', [(3, 4, 300, 192, True, False),      # cooperative long-sequence kernels (>= 6 query tiles)
                                                       (3, 4, 300, 192, True, True),       # ... with the forward's 1-bit mask handed to the backward
                                                       (2, 4, 330, 192, True, True),       # ... an odd number of query tiles (the dK / dV loop runs in pairs)
                                                       (2, 4, 300, 192, False, True),      # ... batch-first rows (row pitch = one token)
                                                       (2, 3, 333, 64, False, True), (2, 3, 197, 256, False, True),    # (hd = 256: mask ignored)
                                                       (2, 3, 197, 256, False, False), (4, 6, 26, 64, False, False), (3, 4, 100, 192, True, False),
                                                       (3, 3, 27, 256, False, False), (2, 4, 30, 192, True, False)])     # single-launch backward at hd = 192 / 256
def test_attention_weight_dropout_uses_the_oracle_mask(Bb, H, N, hd, seq_first, stored):
    """Dropout on the attention weights (site 0 of nn.TransformerEncoderLayer): P' = softmax(S) * keep / (1 - p) with the
    counter-based mask of oracle.voxel_oracle.hash_keep_mask over the [Bb*H, N, N] index space -- forward and backward of every
    kernel family (per-wave, single-launch small, cooperative) against autograd on exactly that mask."""
    from oracle import voxel_oracle as vo
    g = torch.Generator().manual_seed(16)
    D, rows, p, seed_v, site = H * hd, Bb * N, 0.1, 4321, 0
    qkv = torch.randn(rows, 3 * D, generator=g).to(DEV)
    sb, st = (1, Bb) if seq_first else (N, 1)
    hi, lo = ops.split_bf16(qkv)
    seed = torch.tensor([seed_v], dtype=torch.int64, device=DEV)
    drop = (p, seed, site)
    T = (N + 31) // 32
    mbuf = torch.full((Bb * H, T, T, 32), 0x5A5A5A5A, dtype=torch.int32, device=DEV) if stored else None   # S3dAttnArgs::drop_mask
    out_hi, out_lo, lse = ops.attention_fwd(hi, lo, Bb, H, N, D, sb, st, split=True, drop=drop, drop_mask=mbuf)
    keep01 = vo.hash_keep_mask((Bb, H, N, N), seed_v, site, p).to(DEV)
    keep = keep01.double() / (1.0 - p)
    if stored and hd < 256:            # the stored bits ARE the oracle's mask: word q of tile (qt, kt), bit = key within the tile
        pad = torch.zeros(Bb * H, T * 32, T * 32, dtype=torch.int64, device=DEV)
        pad[:, :N, :N] = keep01.reshape(Bb * H, N, N).long()
        want = (pad.view(Bb * H, T, 32, T, 32) << torch.arange(32, device=DEV)).sum(-1).permute(0, 1, 3, 2)       # [bh][qt][kt][q]
        got_w = mbuf.long() & 0xFFFFFFFF
        valid_k = (torch.arange(T * 32, device=DEV) < N).view(T, 32).long()
        kmask = (valid_k << torch.arange(32, device=DEV)).sum(-1)                                                  # bits of keys < N, per key tile
        qvalid = (torch.arange(T * 32, device=DEV) < N).view(T, 1, 32)
        assert bool((((got_w ^ want) & kmask.view(1, 1, T, 1)) * qvalid.long()).eq(0).all()), 'stored mask bits differ from the oracle mask'
    elif stored:
        assert bool((mbuf == 0x5A5A5A5A).all())        # hd = 256: the forward is not a cooperative kernel, the buffer is not touched

    def to_b(t):
        return t.view(N, Bb, -1).transpose(0, 1) if seq_first else t.view(Bb, N, -1)

    def ref_attn(x):
        q, k, v = [x[..., i * D:(i + 1) * D].reshape(Bb, N, H, hd).transpose(1, 2) for i in range(3)]
        s = (q @ k.transpose(-2, -1)) * hd ** -0.5
        return (s.softmax(-1) * keep) @ v
    x = to_b(qkv.double()).requires_grad_(True)
    ref = ref_attn(x)
    got = to_b(out_hi.float() + out_lo.float()).reshape(Bb, N, H, hd).transpose(1, 2)
    e = rel_err(got, ref.detach())
    assert e < 1e-4, f'fwd rel err {e:.3e} (a wrong mask gives ~0.3)'
    dout = torch.randn(rows, D, generator=g).to(DEV).to(torch.bfloat16)
    dqkv = ops.attention_bwd(hi, out_hi, out_lo, lse, dout, Bb, H, N, D, sb, st, drop=drop, drop_mask=mbuf)
    if stored:                         # the same gradient as the backward that evaluates the hash: the mask arithmetic differs only in where
        ref_h = ops.attention_bwd(hi, out_hi, out_lo, lse, dout, Bb, H, N, D, sb, st, drop=drop).float()   # a product is fused into an fma
        diff = (dqkv.float() - ref_h).abs()
        assert float(diff.max()) <= 2.0 ** -7 * float(ref_h.abs().max()) and float((diff > 0).float().mean()) < 1e-2
    xb = to_b(hi.double()).requires_grad_(True)
    ref_attn(xb).backward(to_b(dout.double()).reshape(Bb, N, H, hd).transpose(1, 2))
    gotd = to_b(dqkv.float())
    for i, name in enumerate('qkv'):
        e = rms_err(gotd[..., i * D:(i + 1) * D], xb.grad[..., i * D:(i + 1) * D])
        assert e < 2e-2, f'd{name} rms err {e:.3e}'


@pytest.mark.parametrize('stored', [False, True])
def test_attention_single_plane_probabilities_keep_the_row_statistics_and_stay_within_bf16_of_the_full_split(stored):
    """S3dAttnArgs::p_single_plane (the seq-first encoder layer's forward sets it): P enters P V as one bf16 plane.  The log-sum-exp (what the
    backward recomputes P from) and the stored dropout bits are those of the full split bit for bit; the output moves by at most the rounding
    of the weights (2^-9 relative per weight -> <= 2^-9 max|V| per element, far less in rms) and still matches fp64 at 2e-3 relative."""
    g = torch.Generator().manual_seed(23)
    Bb, H, N, hd = 3, 4, 300, 192
    D, rows = H * hd, Bb * N
    qkv = torch.randn(rows, 3 * D, generator=g).to(DEV)
    hi, lo = ops.split_bf16(qkv)
    T = (N + 31) // 32
    seed = torch.tensor([99], dtype=torch.int64, device=DEV)
    kw = dict(drop=(0.1, seed, 0), drop_mask=None) if stored else {}
    outs = []
    for flag in (0, 1):
        if stored: kw['drop_mask'] = torch.zeros((Bb * H, T, T, 32), dtype=torch.int32, device=DEV)
        o_hi, o_lo, lse = ops.attention_fwd(hi, lo, Bb, H, N, D, 1, Bb, split=True, p_single_plane=flag, **kw)
        outs.append((o_hi.float() + o_lo.float(), lse.clone(), kw.get('drop_mask')))
    (full, lse0, m0), (one, lse1, m1) = outs
    assert torch.equal(lse0, lse1)
    if stored: assert torch.equal(m0, m1)
    assert not torch.equal(full, one), 'the flag did not select the single-plane kernel'
    vmax = float(qkv[:, 2 * D:].abs().max())
    assert float((full - one).abs().max()) <= 2.0 ** -9 * vmax * (1.0 / 0.9 if stored else 1.0)
    assert 1e-6 < rel_err(one, full) < 5e-3
    if not stored:                     # against fp64 of the operator (the full split holds 1e-4 in test_attention_fwd_bwd)
        x = qkv.double().view(N, Bb, 3, H, hd).permute(2, 1, 3, 0, 4)
        ref = ((x[0] @ x[1].transpose(-2, -1)) * hd ** -0.5).softmax(-1) @ x[2]
        got = one.view(N, Bb, H, hd).permute(1, 2, 0, 3)
        assert rel_err(got, ref) < 5e-3


@pytest.mark.parametrize('H,N,hd,seg', [(3, 30, 256, 15), (6, 26, 64, 13), (4, 32, 48, 16), (2, 21, 96, 11)])
def test_attention_block_diagonal_segments(H, N, hd, seg):
    """S3dAttnArgs::seg: a query attends to the keys of its own segment only (what the launchers use to pack two short
    sequences into one tile); here requested explicitly, incl. unequal segments (N < 2*seg)."""
    g = torch.Generator().manual_seed(16)
    Bb, D = 3, H * hd
    qkv = torch.randn(Bb * N, 3 * D, generator=g).to(DEV)
    hi, lo = ops.split_bf16(qkv)
    out_hi, out_lo, lse = ops.attention_fwd(hi, lo, Bb, H, N, D, N, 1, split=True, seg=seg)
    mask = (torch.arange(N)[:, None] >= seg) == (torch.arange(N)[None, :] >= seg)

    def ref_fn(x):
        q, k, v = [x[..., i * D:(i + 1) * D].reshape(Bb, N, H, hd).transpose(1, 2) for i in range(3)]
        s = (q @ k.transpose(-2, -1)) * hd ** -0.5
        return s.masked_fill(~mask.to(s.device), float('-inf')).softmax(-1) @ v

    ref = ref_fn(qkv.double().view(Bb, N, -1))
    got = (out_hi.float() + out_lo.float()).view(Bb, N, H, hd).transpose(1, 2)
    assert rel_err(got, ref) < 1e-4
    dout_bf = torch.randn(Bb * N, D, generator=g).to(DEV).to(torch.bfloat16)
    dqkv = ops.attention_bwd(hi, out_hi, out_lo, lse, dout_bf, Bb, H, N, D, N, 1, seg=seg)
    xb = hi.double().view(Bb, N, -1).requires_grad_(True)
    ref_fn(xb).backward(dout_bf.double().view(Bb, N, H, hd).transpose(1, 2))
    for i, name in enumerate('qkv'):
        e = rms_err(dqkv.float().view(Bb, N, -1)[..., i * D:(i + 1) * D], xb.grad[..., i * D:(i + 1) * D])
        assert e < 2e-2, f'd{name} rms err {e:.3e}'
    with pytest.raises(RuntimeError, match='seg'):
        ops.attention_fwd(hi, lo, Bb, H, N, D, N, 1, seg=N // 2 - 1)


@pytest.mark.parametrize('kind,V,c,P,D', [('VoxelEmbed', 30, 6, 5, 384), ('VoxelEmbed', 32, 6, 5, 384),
                                          ('VoxelEmbed_no_average', 12, 4, 3, 192), ('VoxelNaiveProjection', 30, 6, 5, 384),
                                          ('VoxelEmbed', 128, 16, 8, 768), ('VoxelEmbed_no_average', 36, 9, 4, 192)])
def test_tokenizer_modules_match_oracle(kind, V, c, P, D):
    from oracle import voxel_oracle as vo
    g = torch.Generator().manual_seed(7)
    mod = getattr(s3d, kind)(voxel_size=V, cell_size=c, patch_size=P, embed_dim=D).to(DEV)
    x = (torch.rand(2, 1, V, V, V, generator=g) < 0.1).float()
    w, b = mod.proj[0].weight.detach().cpu(), mod.proj[0].bias.detach().cpu()
    fn = {'VoxelEmbed': vo.voxel_embed, 'VoxelEmbed_no_average': vo.voxel_embed_no_average,
          'VoxelNaiveProjection': vo.voxel_naive_projection}[kind]
    ref = fn(x, w, b, c)
    with torch.no_grad():
        got = mod(x.to(DEV))
    assert tuple(got.shape) == tuple(ref.shape)
    assert rel_err(got, ref) < 2e-5
    # non-binary (general float) grids go through the same path
    xf = torch.rand(2, 1, V, V, V, generator=g)
    with torch.no_grad():
        assert rel_err(mod(xf.to(DEV)), fn(xf, w, b, c)) < 5e-5
    with pytest.raises(AssertionError):
        mod(torch.zeros(1, 1, V + 2, V + 2, V + 2, device=DEV))


@pytest.mark.parametrize('am', [False, True])
def test_head_and_cross_entropy(am):
    from oracle import voxel_oracle as vo
    g = torch.Generator().manual_seed(8)
    B, D, C = 8, 384, 40
    feat = torch.randn(B, D, generator=g)
    y = torch.randint(0, C, (B,), generator=g)
    wgt = torch.rand(C, generator=g) + 0.5
    if am:
        W = torch.randn(D, C, generator=g) * 0.1
        sd = {'voxel_head.W': W}
    else:
        W = torch.randn(C, D, generator=g) * 0.1
        bias = torch.randn(C, generator=g) * 0.1
        sd = {'voxel_head.weight': W, 'voxel_head.bias': bias}
    fr = feat.clone().requires_grad_(True)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    logits_ref = vo.voxel_head(fr, sdr)
    loss_ref = vo.cross_entropy(logits_ref, y, wgt)
    loss_ref.backward()

    featd = feat.to(DEV)
    logits = torch.empty(B, C, device=DEV)
    dfeat = torch.empty(B, D, device=DEV)
    dW = torch.zeros_like(W, device=DEV)
    dbias = torch.zeros(C, device=DEV)
    scratch = torch.zeros(C + B, device=DEV)
    Wd = W.to(DEV)
    h = L.fill(L.S3dHeadArgs(), feat=featd, B=B, D=D, C=C, W=Wd, bias=None if am else bias.to(DEV), logits=logits,
               am_softmax=int(am), am_scale=30.0, dfeat=dfeat, dW=dW, dbias=dbias, scratch=scratch)
    if not am:
        bd = bias.to(DEV); h.bias = bd.data_ptr()
    L.check(L.lib().s3d_head_fwd(ctypes.byref(h), L.current_stream()), 'head_fwd')
    assert rel_err(logits, logits_ref.detach()) < 1e-5
    loss, dl = ops.cross_entropy(logits, y.to(DEV), wgt.to(DEV))
    assert abs(float(loss) - float(loss_ref)) < 1e-5
    h.dlogits = dl.data_ptr()
    L.check(L.lib().s3d_head_bwd(ctypes.byref(h), L.current_stream()), 'head_bwd')
    assert rel_err(dfeat, fr.grad) < 1e-4
    key = 'voxel_head.W' if am else 'voxel_head.weight'
    assert rel_err(dW, sdr[key].grad) < 1e-4
    if not am:
        assert rel_err(dbias, sdr['voxel_head.bias'].grad) < 1e-4
    # unweighted
    loss2, _ = ops.cross_entropy(logits, y.to(DEV))
    assert abs(float(loss2) - float(vo.cross_entropy(logits_ref.detach(), y))) < 1e-5


def test_adam_matches_torch_and_refreshes_planes():
    g = torch.Generator().manual_seed(9)
    n = 4096
    p0 = torch.randn(n, generator=g)
    q = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([q], lr=1e-3)
    p = p0.clone().to(DEV); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    hi = torch.zeros(n, dtype=torch.bfloat16, device=DEV); lo = torch.zeros_like(hi)
    st = np.zeros(9, dtype=np.int32)
    st[:7] = np.array([1e-3, 0.9, 0.999, 1e-8, 1.0, 0, 0], dtype=np.float32).view(np.int32)
    state = torch.from_numpy(st).to(DEV)
    for step in range(1, 5):
        grad = torch.randn(n, generator=g) * (0.1 * step)
        q.grad = grad.clone(); opt.step()
        gd = grad.to(DEV)
        L.check(L.lib().s3d_adam_step(L.ptr(p), L.ptr(gd), L.ptr(m), L.ptr(v), L.ptr(hi), L.ptr(lo), ctypes.c_long(n),
                                      L.ptr(state), 1, L.current_stream()), 'adam')
        assert float((p.cpu() - q.detach()).abs().max()) < 2e-7
        assert float(gd.abs().max()) == 0.0                                   # zero_grad fused
        assert float((hi.float() + lo.float() - p).abs().max()) < 1e-4
    assert int(state[7]) == 4


@pytest.mark.parametrize('D,H,N,Bb', [(384, 6, 26, 8), (192, 3, 10, 3), (768, 3, 15, 5),
                                      (384, 6, 26, 64), (192, 3, 513, 8),      # the benched cfg-2 block (1664 rows: gemm_pair_dmat_kernel, fused forward) and a cfg-5 slice (513 tokens)
                                      # the point path's block at >= 16 384 rows: norm2 + fc1 + GELU + fc2 + residual as ONE launch (fused_mlp.hip),
                                      # bands of eight waves (16 448 rows) and of nine (cfg-4's 32 896 rows: 229 workgroups instead of 257)
                                      (192, 3, 257, 64), (192, 3, 257, 128)])
def test_block_fwd_bwd_matches_oracle(D, H, N, Bb):
    """One timm Block through s3d_block_fwd / s3d_block_bwd vs autograd on the oracle restatement."""
    from oracle import voxel_oracle as vo
    from simple3d_former_amd.engine import ParamArena, _BlockWorkspace, _BlockScratch, LN_EPS
    g = torch.Generator().manual_seed(10)
    M, Hd = Bb * N, 4 * D
    shapes = {'blocks.0.norm1.weight': (D,), 'blocks.0.norm1.bias': (D,), 'blocks.0.attn.qkv.weight': (3 * D, D),
              'blocks.0.attn.qkv.bias': (3 * D,), 'blocks.0.attn.proj.weight': (D, D), 'blocks.0.attn.proj.bias': (D,),
              'blocks.0.norm2.weight': (D,), 'blocks.0.norm2.bias': (D,), 'blocks.0.mlp.fc1.weight': (Hd, D),
              'blocks.0.mlp.fc1.bias': (Hd,), 'blocks.0.mlp.fc2.weight': (D, Hd), 'blocks.0.mlp.fc2.bias': (D,)}
    sd = {}
    for k, shp in shapes.items():
        if 'norm' in k and k.endswith('weight'):
            sd[k] = 1 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith('weight'):
            sd[k] = torch.randn(shp, generator=g) * 0.05
        else:
            sd[k] = torch.randn(shp, generator=g) * 0.05
    x = torch.randn(Bb, N, D, generator=g)
    dy = torch.randn(Bb, N, D, generator=g) * 0.1
    # oracle
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    y_ref = vo.vit_block(xr, leaf, 0, H)
    y_ref.backward(dy)
    # engine pieces
    arena = ParamArena(shapes, torch.device(DEV)); arena.load(sd); arena.refresh_planes()
    ws = _BlockWorkspace(1, Bb, N, D, H, Hd, DEV, True)
    sc = _BlockScratch(M, D, H, Hd, Bb * H * N, DEV)
    p = 'blocks.0.'
    bp = L.fill(L.S3dBlockParams(), ln1_w=arena.param(p + 'norm1.weight'), ln1_b=arena.param(p + 'norm1.bias'),
                ln2_w=arena.param(p + 'norm2.weight'), ln2_b=arena.param(p + 'norm2.bias'), qkv_b=arena.param(p + 'attn.qkv.bias'),
                proj_b=arena.param(p + 'attn.proj.bias'), fc1_b=arena.param(p + 'mlp.fc1.bias'), fc2_b=arena.param(p + 'mlp.fc2.bias'),
                qkv_w_hi=arena.hi_of(p + 'attn.qkv.weight'), qkv_w_lo=arena.lo_of(p + 'attn.qkv.weight'),
                proj_w_hi=arena.hi_of(p + 'attn.proj.weight'), proj_w_lo=arena.lo_of(p + 'attn.proj.weight'),
                fc1_w_hi=arena.hi_of(p + 'mlp.fc1.weight'), fc1_w_lo=arena.lo_of(p + 'mlp.fc1.weight'),
                fc2_w_hi=arena.hi_of(p + 'mlp.fc2.weight'), fc2_w_lo=arena.lo_of(p + 'mlp.fc2.weight'))
    bg = L.fill(L.S3dBlockGrads(), ln1_w=arena.grad(p + 'norm1.weight'), ln1_b=arena.grad(p + 'norm1.bias'),
                ln2_w=arena.grad(p + 'norm2.weight'), ln2_b=arena.grad(p + 'norm2.bias'), qkv_w=arena.grad(p + 'attn.qkv.weight'),
                qkv_b=arena.grad(p + 'attn.qkv.bias'), proj_w=arena.grad(p + 'attn.proj.weight'), proj_b=arena.grad(p + 'attn.proj.bias'),
                fc1_w=arena.grad(p + 'mlp.fc1.weight'), fc1_b=arena.grad(p + 'mlp.fc1.bias'), fc2_w=arena.grad(p + 'mlp.fc2.weight'),
                fc2_b=arena.grad(p + 'mlp.fc2.bias'))
    ws.x[0].copy_(x.reshape(M, D))
    L.check(L.lib().s3d_block_fwd(ctypes.byref(ws.shape), ctypes.byref(bp), ctypes.byref(ws.acts[0]), L.current_stream()), 'block_fwd')
    e = rel_err(ws.x[1], y_ref.detach().reshape(M, D))
    assert e < 1e-4, f'block fwd rel err {e:.3e}'                       # split-bf16 forward
    sc.dx_a.copy_(dy.reshape(M, D)); sc.dx_a_bf.copy_(dy.reshape(M, D).to(torch.bfloat16))
    L.check(L.lib().s3d_block_bwd(ctypes.byref(ws.shape), ctypes.byref(bp), ctypes.byref(bg), ctypes.byref(ws.acts[0]),
                                  ctypes.byref(sc.c), L.current_stream()), 'block_bwd')
    e = rms_err(sc.dx_a, xr.grad.reshape(M, D))
    assert e < 2e-2, f'block dx rms err {e:.3e}'                        # plain-bf16 backward
    for k in shapes:
        e = rms_err(arena.grad(k), leaf[k].grad)
        assert e < 3e-2, f'{k}: grad rms err {e:.3e}'


@pytest.mark.parametrize('D,H,N,Bb,depth', [(384, 6, 26, 64, 3), (384, 6, 26, 5, 2), (192, 3, 10, 3, 2), (192, 3, 10, 4, 2), (192, 3, 32, 2, 1), (384, 6, 1, 7, 1)])
def test_fused_block_launches_equal_the_seven_launch_sequence(D, H, N, Bb, depth):
    """s3d_blocks_fwd with S3dBlockShape::fuse = 0 (norm1 + qkv + attention in one launch, norm2 + fc1 + GELU in another) against
    fuse = -1 (one launch per operator) on the same parameters: the residual stream and EVERY saved activation the backward reads.
    Same arithmetic (split-bf16 products, fp32 accumulation / LayerNorm / softmax / GELU), different summation order: fp32 values to
    5e-6 of their scale (hi + lo sums: 2e-5, the resolution of the split), bf16 planes to one bf16 ulp of a few entries; the forward stays bitwise reproducible run to run."""
    from simple3d_former_amd.engine import ParamArena, _BlockWorkspace
    g = torch.Generator().manual_seed(11)
    Hd, M = 4 * D, Bb * N
    shapes = {}
    for i in range(depth):
        p = f'blocks.{i}.'
        shapes.update({p + 'norm1.weight': (D,), p + 'norm1.bias': (D,), p + 'attn.qkv.weight': (3 * D, D), p + 'attn.qkv.bias': (3 * D,),
                       p + 'attn.proj.weight': (D, D), p + 'attn.proj.bias': (D,), p + 'norm2.weight': (D,), p + 'norm2.bias': (D,),
                       p + 'mlp.fc1.weight': (Hd, D), p + 'mlp.fc1.bias': (Hd,), p + 'mlp.fc2.weight': (D, Hd), p + 'mlp.fc2.bias': (D,)})
    sd = {k: (1 + 0.1 * torch.randn(shp, generator=g)) if ('norm' in k and k.endswith('weight')) else torch.randn(shp, generator=g) * 0.05
          for k, shp in shapes.items()}
    arena = ParamArena(shapes, torch.device(DEV)); arena.load(sd); arena.refresh_planes()
    bp = (L.S3dBlockParams * depth)()
    for i in range(depth):
        p = f'blocks.{i}.'
        L.fill(bp[i], ln1_w=arena.param(p + 'norm1.weight'), ln1_b=arena.param(p + 'norm1.bias'), ln2_w=arena.param(p + 'norm2.weight'),
               ln2_b=arena.param(p + 'norm2.bias'), qkv_b=arena.param(p + 'attn.qkv.bias'), proj_b=arena.param(p + 'attn.proj.bias'),
               fc1_b=arena.param(p + 'mlp.fc1.bias'), fc2_b=arena.param(p + 'mlp.fc2.bias'),
               qkv_w_hi=arena.hi_of(p + 'attn.qkv.weight'), qkv_w_lo=arena.lo_of(p + 'attn.qkv.weight'),
               proj_w_hi=arena.hi_of(p + 'attn.proj.weight'), proj_w_lo=arena.lo_of(p + 'attn.proj.weight'),
               fc1_w_hi=arena.hi_of(p + 'mlp.fc1.weight'), fc1_w_lo=arena.lo_of(p + 'mlp.fc1.weight'),
               fc2_w_hi=arena.hi_of(p + 'mlp.fc2.weight'), fc2_w_lo=arena.lo_of(p + 'mlp.fc2.weight'))
    x = torch.randn(M, D, generator=g)
    runs = {}
    for fuse in (False, True, True):
        ws = _BlockWorkspace(depth, Bb, N, D, H, Hd, DEV, True, fuse=fuse, shared_lo=False)
        for t in (ws.stats, ws.lse, ws.xn1, ws.xn1_lo, ws.qkv, ws.att, ws.att_lo, ws.xn2, ws.xn2_lo, ws.hpre, ws.hact, ws.hact_lo):
            t.fill_(float('nan'))                                       # whatever the backward reads must have been written
        ws.x[0].copy_(x)
        L.check(L.lib().s3d_blocks_fwd(ctypes.byref(ws.shape), bp, ws.acts, depth, L.current_stream()), 'blocks_fwd')
        torch.cuda.synchronize()
        runs.setdefault(fuse, []).append(ws)
    ref, (got, again) = runs[False][0], runs[True]

    def close32(a, b, what, rel=5e-6):
        scale = float(b.abs().max())
        e = float((a - b).abs().max())
        assert e <= rel * scale + 1e-7, f'{what}: {e:.3e} of scale {scale:.3e}'

    def close16(a, b, what):                                            # bf16 planes: a rounding boundary may flip by one ulp
        a, b = a.float(), b.float()
        assert not torch.isnan(a).any(), f'{what}: not written'
        e = (a - b).abs()
        tol = b.abs() * 2.0 ** -7 + 5e-6 * float(b.abs().max())         # one bf16 ulp, or the fp32 difference itself near zero
        assert bool((e <= tol).all()), f'{what}: max {float(e.max()):.3e}'
        assert float((e > 0).float().mean()) < 0.02, f'{what}: {float((e > 0).float().mean()):.3%} of the entries differ'

    for i in range(depth):
        close32(got.x[i + 1], ref.x[i + 1], f'x_out[{i}]')
        close32(got.x_mid[i], ref.x_mid[i], f'x_mid[{i}]')
        close32(got.stats[i], ref.stats[i], f'mean / rstd [{i}]')
        close32(got.lse[i], ref.lse[i], f'lse[{i}]')
        close16(got.xn1[i], ref.xn1[i], 'xn1_hi'); close16(got.xn2[i], ref.xn2[i], 'xn2_hi')
        close32(got.xn1[i].float() + got.xn1_lo[i].float(), ref.xn1[i].float() + ref.xn1_lo[i].float(), 'xn1 hi + lo', rel=2e-5)        # the split itself resolves 2^-16 of the value
        close32(got.xn2[i].float() + got.xn2_lo[i].float(), ref.xn2[i].float() + ref.xn2_lo[i].float(), 'xn2 hi + lo', rel=2e-5)        # the split itself resolves 2^-16 of the value
        close16(got.qkv[i], ref.qkv[i], 'qkv_hi'); close16(got.att[i], ref.att[i], 'att_hi')
        close32(got.att[i].float() + got.att_lo[i].float(), ref.att[i].float() + ref.att_lo[i].float(), 'att hi + lo', rel=2e-5)        # the split itself resolves 2^-16 of the value
        close16(got.hpre[i], ref.hpre[i], 'hpre'); close16(got.hact[i], ref.hact[i], 'hact_hi')
        close32(got.hact[i].float() + got.hact_lo[i].float(), ref.hact[i].float() + ref.hact_lo[i].float(), 'hact hi + lo', rel=2e-5)        # the split itself resolves 2^-16 of the value
        assert torch.equal(got.x[i + 1], again.x[i + 1]) and torch.equal(got.hact[i], again.hact[i]) and torch.equal(got.hact_lo[i], again.hact_lo[i]) and torch.equal(got.att[i], again.att[i])


@pytest.mark.parametrize('D,G,Nb', [(192, 27, 4), (768, 40, 15), (384, 50, 6)])
def test_group_encoder_layer_fwd_bwd_matches_oracle(D, G, Nb):
    """nn.TransformerEncoderLayer(d_model=D, dim_feedforward=D, nhead=4), seq-first as fed at vit_3d_2d_pretrain.py:479,
    through s3d_encoder_layer_fwd/bwd vs autograd on the oracle restatement (eval-mode dropout)."""
    from oracle import voxel_oracle as vo
    from simple3d_former_amd.engine import ParamArena, _BlockScratch
    g = torch.Generator().manual_seed(11)
    M = G * Nb
    ge = 'group_embed.'
    shapes = {ge + 'self_attn.in_proj_weight': (3 * D, D), ge + 'self_attn.in_proj_bias': (3 * D,),
              ge + 'self_attn.out_proj.weight': (D, D), ge + 'self_attn.out_proj.bias': (D,),
              ge + 'linear1.weight': (D, D), ge + 'linear1.bias': (D,), ge + 'linear2.weight': (D, D), ge + 'linear2.bias': (D,),
              ge + 'norm1.weight': (D,), ge + 'norm1.bias': (D,), ge + 'norm2.weight': (D,), ge + 'norm2.bias': (D,)}
    sd = {k: (1 + 0.1 * torch.randn(shp, generator=g)) if ('norm' in k and k.endswith('weight')) else torch.randn(shp, generator=g) * 0.06
          for k, shp in shapes.items()}
    x = torch.randn(G, Nb, D, generator=g)
    dy = torch.randn(G, Nb, D, generator=g) * 0.1
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    y_ref = vo.group_encoder_layer(xr, leaf)
    y_ref.backward(dy)

    arena = ParamArena(shapes, torch.device(DEV)); arena.load(sd); arena.refresh_planes()
    f32 = dict(dtype=torch.float32, device=DEV); b16 = dict(dtype=torch.bfloat16, device=DEV)
    t = {n: torch.empty(M, D, **f32) for n in ('x_in', 's1', 'x1', 's2', 'x_out')}
    stats = torch.empty(4, M, **f32); lse = torch.empty(Nb * 4 * G, **f32)
    pl = {n: torch.empty(2, M, w, **b16) for n, w in (('xin', D), ('qkv', 3 * D), ('att', D), ('x1p', D), ('f', D))}
    fpre = torch.empty(M, D, **b16)
    acts = L.fill(L.S3dEncActs(), x_in=t['x_in'], s1=t['s1'], x1=t['x1'], s2=t['s2'], x_out=t['x_out'], mean1=stats[0], rstd1=stats[1],
                  mean2=stats[2], rstd2=stats[3], lse=lse, xin_hi=pl['xin'][0], xin_lo=pl['xin'][1], qkv_hi=pl['qkv'][0],
                  qkv_lo=pl['qkv'][1], att_hi=pl['att'][0], att_lo=pl['att'][1], x1_hi=pl['x1p'][0], x1_lo=pl['x1p'][1], fpre=fpre,
                  f_hi=pl['f'][0], f_lo=pl['f'][1])
    shape = L.S3dEncShape(G=G, Nb=Nb, D=D, H=4, Dff=D, eps=1e-5, split=1)
    P = lambda k: arena.param(ge + k)
    ep = L.fill(L.S3dEncParams(), in_b=P('self_attn.in_proj_bias'), out_b=P('self_attn.out_proj.bias'), l1_b=P('linear1.bias'),
                l2_b=P('linear2.bias'), n1_w=P('norm1.weight'), n1_b=P('norm1.bias'), n2_w=P('norm2.weight'), n2_b=P('norm2.bias'),
                in_w_hi=arena.hi_of(ge + 'self_attn.in_proj_weight'), in_w_lo=arena.lo_of(ge + 'self_attn.in_proj_weight'),
                out_w_hi=arena.hi_of(ge + 'self_attn.out_proj.weight'), out_w_lo=arena.lo_of(ge + 'self_attn.out_proj.weight'),
                l1_w_hi=arena.hi_of(ge + 'linear1.weight'), l1_w_lo=arena.lo_of(ge + 'linear1.weight'),
                l2_w_hi=arena.hi_of(ge + 'linear2.weight'), l2_w_lo=arena.lo_of(ge + 'linear2.weight'))
    Gd = lambda k: arena.grad(ge + k)
    eg = L.fill(L.S3dEncGrads(), in_w=Gd('self_attn.in_proj_weight'), in_b=Gd('self_attn.in_proj_bias'), out_w=Gd('self_attn.out_proj.weight'),
                out_b=Gd('self_attn.out_proj.bias'), l1_w=Gd('linear1.weight'), l1_b=Gd('linear1.bias'), l2_w=Gd('linear2.weight'),
                l2_b=Gd('linear2.bias'), n1_w=Gd('norm1.weight'), n1_b=Gd('norm1.bias'), n2_w=Gd('norm2.weight'), n2_b=Gd('norm2.bias'))
    t['x_in'].copy_(x.reshape(M, D))
    L.check(L.lib().s3d_encoder_layer_fwd(ctypes.byref(shape), ctypes.byref(ep), ctypes.byref(acts), L.current_stream()), 'enc fwd')
    e = rel_err(t['x_out'], y_ref.detach().reshape(M, D))
    assert e < 1e-4, f'encoder fwd rel err {e:.3e}'
    sc = _BlockScratch(M, D, 4, 4 * D, Nb * 4 * G, DEV)
    sc.dx_a.copy_(dy.reshape(M, D))
    L.check(L.lib().s3d_encoder_layer_bwd(ctypes.byref(shape), ctypes.byref(ep), ctypes.byref(eg), ctypes.byref(acts),
                                          ctypes.byref(sc.c), L.current_stream()), 'enc bwd')
    e = rms_err(sc.dx_b, xr.grad.reshape(M, D))
    assert e < 2e-2, f'encoder dx rms err {e:.3e}'
    assert rms_err(sc.dx_b_bf.float(), xr.grad.reshape(M, D)) < 2e-2
    for k in shapes:
        e = rms_err(arena.grad(k), leaf[k].grad)
        assert e < 3e-2, f'{k}: grad rms err {e:.3e}'


def test_assemble_tokens_fwd_bwd():
    B, n, D = 3, 9, 192
    g = torch.Generator().manual_seed(12)
    src = torch.randn(B * n, D, generator=g).to(DEV); cls = torch.randn(D, generator=g).to(DEV)
    pos = torch.randn(n + 1, D, generator=g).to(DEV)
    out = torch.empty(B * (n + 1), D, device=DEV)
    L.check(L.lib().s3d_assemble_tokens(L.ptr(src), L.ptr(cls), L.ptr(pos), L.ptr(out), ctypes.c_long(B), n, D, L.current_stream()), 'asm')
    ref = torch.cat((cls.expand(B, 1, D), src.view(B, n, D)), dim=1) + pos
    assert torch.equal(out.view(B, n + 1, D), ref)
    back = torch.empty_like(src)
    L.check(L.lib().s3d_assemble_tokens_bwd(L.ptr(out), L.ptr(back), ctypes.c_long(B), n, D, L.current_stream()), 'asm bwd')
    assert torch.equal(back.view(B, n, D), ref[:, 1:])


def test_block_stack_on_a_carved_workspace_equals_the_engine_layout():
    """s3d_block_workspace_bytes / _carve: a depth-3 block stack (cfg-2 block shape, class-rows-only last block) forward + backward on ONE
    caller allocation laid out by the library vs the same calls on the Python host's own buffers (deterministic mode: bitwise)."""
    from simple3d_former_amd.engine import ParamArena, _BlockWorkspace, _BlockScratch, _cls_scratch
    lib = L.lib()
    lib.s3d_block_workspace_bytes.restype = ctypes.c_size_t
    g = torch.Generator().manual_seed(12)
    D, H, N, Bb, depth = 384, 6, 26, 16, 3
    Hd, M = 4 * D, Bb * N
    shapes = {}
    for i in range(depth):
        p = f'blocks.{i}.'
        shapes.update({p + 'norm1.weight': (D,), p + 'norm1.bias': (D,), p + 'attn.qkv.weight': (3 * D, D), p + 'attn.qkv.bias': (3 * D,),
                       p + 'attn.proj.weight': (D, D), p + 'attn.proj.bias': (D,), p + 'norm2.weight': (D,), p + 'norm2.bias': (D,),
                       p + 'mlp.fc1.weight': (Hd, D), p + 'mlp.fc1.bias': (Hd,), p + 'mlp.fc2.weight': (D, Hd), p + 'mlp.fc2.bias': (D,)})
    sd = {k: (1 + 0.1 * torch.randn(shp, generator=g)) if ('norm' in k and k.endswith('weight')) else torch.randn(shp, generator=g) * 0.05
          for k, shp in shapes.items()}
    x = torch.randn(M, D, generator=g)
    dy = torch.zeros(M, D); dy[::N] = torch.randn(Bb, D, generator=g) * 0.1           # d(x_out) lives on the class rows only
    was = lib.s3d_get_deterministic()
    lib.s3d_set_deterministic(1)
    try:
        results = []
        for carved in (False, True):
            arena = ParamArena(shapes, torch.device(DEV)); arena.load(sd); arena.refresh_planes()
            bp, bg = (L.S3dBlockParams * depth)(), (L.S3dBlockGrads * depth)()
            for i in range(depth):
                p = f'blocks.{i}.'
                L.fill(bp[i], ln1_w=arena.param(p + 'norm1.weight'), ln1_b=arena.param(p + 'norm1.bias'), ln2_w=arena.param(p + 'norm2.weight'),
                       ln2_b=arena.param(p + 'norm2.bias'), qkv_b=arena.param(p + 'attn.qkv.bias'), proj_b=arena.param(p + 'attn.proj.bias'),
                       fc1_b=arena.param(p + 'mlp.fc1.bias'), fc2_b=arena.param(p + 'mlp.fc2.bias'),
                       qkv_w_hi=arena.hi_of(p + 'attn.qkv.weight'), qkv_w_lo=arena.lo_of(p + 'attn.qkv.weight'),
                       proj_w_hi=arena.hi_of(p + 'attn.proj.weight'), proj_w_lo=arena.lo_of(p + 'attn.proj.weight'),
                       fc1_w_hi=arena.hi_of(p + 'mlp.fc1.weight'), fc1_w_lo=arena.lo_of(p + 'mlp.fc1.weight'),
                       fc2_w_hi=arena.hi_of(p + 'mlp.fc2.weight'), fc2_w_lo=arena.lo_of(p + 'mlp.fc2.weight'))
                L.fill(bg[i], ln1_w=arena.grad(p + 'norm1.weight'), ln1_b=arena.grad(p + 'norm1.bias'), ln2_w=arena.grad(p + 'norm2.weight'),
                       ln2_b=arena.grad(p + 'norm2.bias'), qkv_w=arena.grad(p + 'attn.qkv.weight'), qkv_b=arena.grad(p + 'attn.qkv.bias'),
                       proj_w=arena.grad(p + 'attn.proj.weight'), proj_b=arena.grad(p + 'attn.proj.bias'), fc1_w=arena.grad(p + 'mlp.fc1.weight'),
                       fc1_b=arena.grad(p + 'mlp.fc1.bias'), fc2_w=arena.grad(p + 'mlp.fc2.weight'), fc2_b=arena.grad(p + 'mlp.fc2.bias'))
            if carved:
                shape = L.S3dBlockShape(Bb=Bb, N=N, D=D, H=H, hidden=Hd, eps=1e-6, split=1, cls_only_block=depth, fuse=0)
                nbytes = lib.s3d_block_workspace_bytes(ctypes.byref(shape), depth, 1)
                buf = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
                assert buf.data_ptr() % 256 == 0
                acts, scr = (L.S3dBlockActs * depth)(), L.S3dBlockScratch()
                zo, zb = ctypes.c_size_t(0), ctypes.c_size_t(0)
                L.check(lib.s3d_block_workspace_carve(ctypes.byref(shape), depth, 1, L.ptr(buf), ctypes.c_size_t(nbytes), acts, ctypes.byref(scr),
                                                      ctypes.byref(zo), ctypes.byref(zb)), 'carve')
                buf.fill_(0xff)                                                    # poison: whatever is read must have been written
                buf[zo.value:zo.value + zb.value].zero_()
                view = lambda ptr, n, dt: buf[ptr - buf.data_ptr():ptr - buf.data_ptr() + n * torch.empty(0, dtype=dt).element_size()].view(dt)
                x0, dxa, dxa_bf = view(acts[0].x_in, M * D, torch.float32), view(scr.dx_a, M * D, torch.float32), view(scr.dx_a_bf, M * D, torch.bfloat16)
                xl = view(acts[depth - 1].x_out, M * D, torch.float32)
                keep = buf
            else:
                ws = _BlockWorkspace(depth, Bb, N, D, H, Hd, DEV, True, cls_only=True)
                scp = _BlockScratch(M, D, H, Hd, Bb * H * N, DEV, depth=depth)
                scr, keep = _cls_scratch(scp.c, M, D, DEV)
                shape, acts = ws.shape, ws.acts
                x0, dxa, dxa_bf, xl = ws.x[0].view(-1), scp.dx_a.view(-1), scp.dx_a_bf.view(-1), ws.x[depth].view(-1)
            x0.copy_(x.reshape(-1))
            L.check(lib.s3d_blocks_fwd(ctypes.byref(shape), bp, acts, depth, L.current_stream()), 'blocks_fwd')
            dxa.copy_(dy.reshape(-1)); dxa_bf.copy_(dy.reshape(-1).to(torch.bfloat16))
            L.check(lib.s3d_blocks_bwd(ctypes.byref(shape), bp, bg, acts, ctypes.byref(scr), depth - 1, 0, L.current_stream()), 'blocks_bwd')
            torch.cuda.synchronize()
            results.append((xl.view(M, D)[::N].clone(), dxa.clone(), arena.g.clone()))
        (y0, d0, g0), (y1, d1, g1) = results
        assert torch.equal(y0, y1) and torch.isfinite(y1).all()                    # class rows of the stack's output
        assert torch.equal(d0, d1) and torch.equal(g0, g1) and float(g1.abs().sum()) > 0
    finally:
        lib.s3d_set_deterministic(was)


@pytest.mark.parametrize('D,H,N,Bb', [(384, 6, 26, 64), (384, 6, 26, 5), (192, 3, 10, 4), (192, 3, 32, 3), (384, 6, 1, 7)])
def test_fused_attention_backward_equals_the_unfused_launches(D, H, N, Bb):
    """S3dBlockShape::fuse = 0 (attn.proj dgrad inside the attention-backward launch -- blk_attn_bwd_kernel -- and attn.proj's wgrad as a
    third problem of the qkv pair launch) against fuse = 1 (proj dgrad || wgrad pair, attention backward, qkv pair) on the same saved
    activations: d(x_in) and every parameter gradient.  Same arithmetic (bf16 operands, fp32 accumulation, the bf16 rounding of d(att)
    at the same place), different summation order: rms difference below 2e-3 of the tensor's rms (a dropped tile / head / k-slice would
    be ~1e-1); 5e-3 where the softmax statistic enters: the fused kernel takes delta = sum_k P dP from the P it recomputes, the unfused one
    rowsum(dO * O) from the saved output -- equal in exact arithmetic, bf16-rounding apart here (with N = 1 the fused dS is exactly 0)."""
    from simple3d_former_amd.engine import ParamArena, _BlockWorkspace, _BlockScratch
    g = torch.Generator().manual_seed(21)
    Hd, M = 4 * D, Bb * N
    p = 'blocks.0.'
    shapes = {p + 'norm1.weight': (D,), p + 'norm1.bias': (D,), p + 'attn.qkv.weight': (3 * D, D), p + 'attn.qkv.bias': (3 * D,),
              p + 'attn.proj.weight': (D, D), p + 'attn.proj.bias': (D,), p + 'norm2.weight': (D,), p + 'norm2.bias': (D,),
              p + 'mlp.fc1.weight': (Hd, D), p + 'mlp.fc1.bias': (Hd,), p + 'mlp.fc2.weight': (D, Hd), p + 'mlp.fc2.bias': (D,)}
    sd = {k: (1 + 0.1 * torch.randn(shp, generator=g)) if ('norm' in k and k.endswith('weight')) else torch.randn(shp, generator=g) * 0.05
          for k, shp in shapes.items()}
    x = torch.randn(M, D, generator=g)
    dy = torch.randn(M, D, generator=g) * 0.1
    lib = L.lib()
    was = lib.s3d_get_deterministic()
    lib.s3d_set_deterministic(1)
    try:
        res = []
        for fuse in (1, 0):
            arena = ParamArena(shapes, torch.device(DEV)); arena.load(sd); arena.refresh_planes()
            bp = L.fill(L.S3dBlockParams(), ln1_w=arena.param(p + 'norm1.weight'), ln1_b=arena.param(p + 'norm1.bias'),
                        ln2_w=arena.param(p + 'norm2.weight'), ln2_b=arena.param(p + 'norm2.bias'), qkv_b=arena.param(p + 'attn.qkv.bias'),
                        proj_b=arena.param(p + 'attn.proj.bias'), fc1_b=arena.param(p + 'mlp.fc1.bias'), fc2_b=arena.param(p + 'mlp.fc2.bias'),
                        qkv_w_hi=arena.hi_of(p + 'attn.qkv.weight'), qkv_w_lo=arena.lo_of(p + 'attn.qkv.weight'),
                        proj_w_hi=arena.hi_of(p + 'attn.proj.weight'), proj_w_lo=arena.lo_of(p + 'attn.proj.weight'),
                        fc1_w_hi=arena.hi_of(p + 'mlp.fc1.weight'), fc1_w_lo=arena.lo_of(p + 'mlp.fc1.weight'),
                        fc2_w_hi=arena.hi_of(p + 'mlp.fc2.weight'), fc2_w_lo=arena.lo_of(p + 'mlp.fc2.weight'))
            bg = L.fill(L.S3dBlockGrads(), ln1_w=arena.grad(p + 'norm1.weight'), ln1_b=arena.grad(p + 'norm1.bias'),
                        ln2_w=arena.grad(p + 'norm2.weight'), ln2_b=arena.grad(p + 'norm2.bias'), qkv_w=arena.grad(p + 'attn.qkv.weight'),
                        qkv_b=arena.grad(p + 'attn.qkv.bias'), proj_w=arena.grad(p + 'attn.proj.weight'), proj_b=arena.grad(p + 'attn.proj.bias'),
                        fc1_w=arena.grad(p + 'mlp.fc1.weight'), fc1_b=arena.grad(p + 'mlp.fc1.bias'), fc2_w=arena.grad(p + 'mlp.fc2.weight'),
                        fc2_b=arena.grad(p + 'mlp.fc2.bias'))
            ws = _BlockWorkspace(1, Bb, N, D, H, Hd, DEV, True)
            ws.shape.fuse = fuse
            sc = _BlockScratch(M, D, H, Hd, Bb * H * N, DEV)
            sc.dqkv.fill_(float('nan'))                                  # every d(q | k | v) entry the qkv GEMMs read must have been written
            ws.x[0].copy_(x.to(DEV))
            L.check(lib.s3d_block_fwd(ctypes.byref(ws.shape), ctypes.byref(bp), ctypes.byref(ws.acts[0]), L.current_stream()), 'block_fwd')
            sc.dx_a.copy_(dy.to(DEV)); sc.dx_a_bf.copy_(dy.to(DEV).to(torch.bfloat16))
            L.check(lib.s3d_block_bwd(ctypes.byref(ws.shape), ctypes.byref(bp), ctypes.byref(bg), ctypes.byref(ws.acts[0]), ctypes.byref(sc.c),
                                      L.current_stream()), 'block_bwd')
            torch.cuda.synchronize()
            res.append((sc.dx_a.clone(), sc.dqkv.float().clone(), {k: arena.grad(k).clone() for k in shapes}))
        (dx0, dq0, g0), (dx1, dq1, g1) = res
        assert torch.isfinite(dq1).all() and torch.isfinite(dx1).all()
        assert rms_err(dq1, dq0) < 5e-3, f'd(qkv): {rms_err(dq1, dq0):.3e}'
        assert rms_err(dx1, dx0) < 5e-3, f'd(x_in): {rms_err(dx1, dx0):.3e}'
        for k in shapes:
            e = rms_err(g1[k], g0[k])
            assert e < (5e-3 if ('qkv' in k or 'norm1' in k) else 2e-3), f'{k}: {e:.3e}'
        assert rms_err(g1[p + 'attn.proj.weight'], g0[p + 'attn.proj.weight']) < 2e-5       # the same wgrad, riding on another launch
    finally:
        lib.s3d_set_deterministic(was)
