// Backward GEMMs of the small-batch block stack (cfg-1 / cfg-2: <= 8192 token rows), round 5.
//
// Until round 4 every Linear backward was ONE launch holding the dgrad (NN) and the wgrad (TN, split-K, fp32 atomics) that consume the
// same dy (gemm.hip: gemm_pair_dmat_kernel).  Only the dgrad is on the critical path of loss.backward() (train_cls_voxel.py:287); the
// wgrad half made every such launch 3 - 7 us longer than the dgrad alone (profiles/r05_nopair_kernel_stats.txt: 13.6 / 17.3 us for the
// pairs, 10.0 - 10.4 us for the dgrads alone).  Here the two are separated:
//
//   * dgrad_splitk_kernel  -- dx = dy @ W (NN: dy k-contiguous, W read k-major as it lies in memory), 64 x 64 output tiles, k split in
//     `nslice` slices that write `nslice` fp32 partial planes; the consumer (ln_bwd_kernel, S3dLnBwdArgs::dy_parts) adds the planes while
//     it loads them.  At 1664 rows a 64 x 64 tiling of a [1664 x 384] output has 156 workgroups for 256 CUs and each of them is a serial
//     chain of 18 - 24 k-steps paced by what ONE workgroup pulls from L2 (~22 B/clk); the slices triple the workgroups
//     (two or three per CU pull ~39 B/clk) and cut the chain to 6 - 8 k-steps, with no atomics and no extra launch for the reduction.
//   * wgrad_group_kernel   -- dW[i] (+)= dy[i]^T x[i], db[i] (+)= colsum(dy[i]) for a LIST of layers (the four Linear layers of several
//     blocks) in one launch: 128 x 128 output tiles over the FULL k (token rows), one workgroup per tile, plain read-modify-write stores.
//     No split-K, no fp32 atomics -> the weight gradients are bitwise reproducible, and the launch sits where nothing waits for it
//     (once per group of blocks, before the group's last LayerNorm backward).
//
// Both run on the LDS-DMA pipeline of gemm.hip (global_load_lds_dwordx4 pieces of 1 KB per wave instruction, source-side swizzle,
// k-major tiles read back with ds_read_b64_tr_b16, counted s_waitcnt vmcnt + one barrier per k-tile).
#include "bwd_gemm.h"
#include "gemm.h"
#include "kmajor.h"
#include "kernels.h"
#include "adam_fill.h"

#include <string.h>

namespace {

__device__ __attribute__((aligned(16))) const unsigned int g_bwd_zeros[4] = {0u, 0u, 0u, 0u};   // DMA source of a zero chunk (partial k-tiles)

// ---------------------------------------------------------------------------------------------------------------------
// dgrad (NN): C = A @ B      A = dy [M][K] (k-contiguous, row pitch lda), B = W [K][N] (k-major, row pitch ldb)
// 256 threads = 2 x 2 waves of 32 x 32; stage = A 64 rows x 128 B + B 64 k-rows x 128 B = 16 KB, NS stages.  Three epilogues:
//   DG_PLANES  k-slices stored as fp32 partial planes (the LayerNorm backward kernel adds them)
//   DG_DGELU   mlp.fc2's dgrad: dh = bf16(acc * gelu'(hpre)), and the two per-row dot products of dh that the LayerNorm-2 backward needs
//              (row statistics, below), accumulated with fp32 atomics into rs1 / rs2
//   DG_LNBWD   the dgrad whose output feeds a LayerNorm backward (mlp.fc1 -> norm2, attn.qkv -> norm1) with that LayerNorm backward as
//              its epilogue: dx = rstd (dy gamma - s1 - xh s2) + dres, bf16 copy, column partials of dgamma / dbeta
//
// Row statistics.  The LayerNorm backward of row m needs s1 = mean_n(dy gamma) and s2 = mean_n(dy gamma xh) over the WHOLE row, which no
// 64-column tile of dy = dz @ W has.  But dy is linear in dz:  s1 = sum_k dz[m][k] u[k] with u[k] = mean_n(W[k][n] gamma[n]) (weights only),
// and s2 = (1/D) sum_k dz[m][k] zz[m][k] with zz[m][k] = sum_n W[k][n] gamma[n] xh[m][n] = pre[m][k] - c[k], c[k] = b[k] + sum_n W[k][n] beta[n]:
// the layer's own saved pre-activation (hpre resp. qkv) minus a weights-only vector.  So the PRODUCER of dz (this file's DG_DGELU epilogue,
// fused_block.hip's attention backward) accumulates both dots per row while it has dz in registers, and the LayerNorm backward becomes
// element-wise -- an epilogue.  Removes two launches per block from the backward chain (ln_bwd_kernel: 6.8 us each at cfg-2).
enum { DG_PLANES = 0, DG_DGELU = 1, DG_LNBWD = 2 };
struct DgradArgs {
    const bf16_t* A; const bf16_t* B; float* C;
    long lda, ldb, ldc, slice_stride;
    int M, N, K, kchunk, ntx, nty, nslice;
    float alpha;
    // DG_DGELU
    const bf16_t* aux; long ldaux; bf16_t* O; long ldo;
    const float* st_u; const float* st_c; float* rs1; float* rs2;       // row statistics out (atomics); vectors of the NEXT dgrad's weight [N]
    float* zero_buf; int zero_n;                                        // a row-statistics buffer nobody uses during this launch: cleared
    // DG_LNBWD
    const float* x; long ldx; const float* mean; const float* rstd; const float* gamma; const float* dres; long lddres;
    float* dx; long lddx; bf16_t* dx_bf; long lddxbf;
    float* partial; float* dgamma; float* dbeta;                         // partial: [nty][2][N] column sums (else atomics into dgamma / dbeta)
    const float* in_s1; const float* in_s2; float inv_d;
};

// KS = 1: four waves (2 x 2 of 32 x 32).  KS = 2: eight waves -- a second group of four takes every other k-tile into accumulators of its own
// (stage = two k-tiles, one barrier per pair) and hands them over through LDS at the end: the k-loop of these launches is a serial chain of
// wait -> barrier -> fragment reads -> MFMAs per k-tile (~0.3 us each, 18 - 24 of them at cfg-2), not a bandwidth limit (a load-only replica of
// the same tile streams its operands twice as fast, profiles/r02_dma_bw_probe.txt), so two groups halve it.
template <int NS, int KS, bool KTAIL>
__device__ __forceinline__ void dgrad_mainloop(const DgradArgs& p, unsigned char* smem, const int m0, const int n0, const int kbeg, const int kslice,
                                               f32x4 (&acc)[2][2]) {
    constexpr int TILE = 16384, STAGE = TILE * KS, PPW = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3, wm = w4 >> 1, wn = w4 & 1;
    const int ntiles = (kslice + 63) >> 6, ktail = KTAIL ? (kslice & 63) : 0;
    const int nsteps = (ntiles + KS - 1) / KS;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    // waves 0, 1 of a group: the A pieces (8 rows x 128 B each); waves 2, 3: the B pieces (8 k-rows x 64 columns each)
    const bf16_t* gp[PPW];
    int gk[(KTAIL || KS > 1) ? PPW : 1];
    const bool isB = w4 >= 2;
    const int k0 = kbeg + grp * 64;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int q = (w4 & 1) * PPW + j;                              // piece 0 .. 7 of the operand
        const int r = q * 8 + (lane >> 3), c = lane & 7;
        if (isB) {
            const int cg = c ^ kmajor_swz<64>(r);
            gp[j] = p.B + (long)(k0 + r) * p.ldb + min(n0 + cg * 8, p.N - 8);
            if constexpr (KTAIL || KS > 1) gk[j] = r;
        } else {
            const int sw = dma_swz64(r);
            gp[j] = p.A + (long)min(m0 + r, p.M - 1) * p.lda + k0 + ((c ^ sw) << 3);
            if constexpr (KTAIL || KS > 1) gk[j] = (c ^ sw) << 3;
        }
    }
    const long gstep = isB ? (long)KS * 64 * p.ldb : (long)KS * 64;
    auto issue = [&](int t) {
        const unsigned dst = lds0 + (unsigned)((t % NS) * STAGE + grp * TILE + w4 * PPW * 1024);
        if constexpr (KTAIL || KS > 1) {
            const int kt = t * KS + grp;                               // this group's k-tile: beyond the slice (odd tile count) / partial -> zeros
            const int valid = kt >= ntiles ? 0 : (KTAIL && ktail != 0 && kt == ntiles - 1) ? ktail : 64;
            if (valid != 64) {                                         // wave-uniform, at most once per launch
#pragma unroll
                for (int j = 0; j < PPW; ++j)
                    glds16(gk[j] < valid ? gp[j] + (long)t * gstep : reinterpret_cast<const bf16_t*>(g_bwd_zeros), dst + j * 1024);
                return;
            }
        }
#pragma unroll
        for (int j = 0; j < PPW; ++j) glds16(gp[j] + (long)t * gstep, dst + j * 1024);
    };
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NS - 1; ++u)
        if (u < nsteps) issue(u);
    for (int t = 0; t < nsteps; ++t) {
        if (t + NS - 1 <= nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + NS - 1 < nsteps) issue(t + NS - 1);
        const unsigned char* sA = smem + (t % NS) * STAGE + grp * TILE;
        const unsigned char* sB = sA + 8192;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = read_frag_dma(sA, wm * 32 + i * 16 + (lane & 15), ks * 4 + (lane >> 4));
            frags_kmajor<64, 2>(sB, wn * 32, ks * 32 + (lane >> 4) * 8, lane, b);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
    }
}

// sum over the 8 lanes of an aligned group (lanes that differ in lane & 7), result in lane & 7 == 0 (and its mirror)
__device__ __forceinline__ float oct_sum(float v) {
    int x = __float_as_int(v);
#define S3D_OCT_STEP(ctrl) x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(x, x, ctrl, 0xf, 0xf, false)));
    S3D_OCT_STEP(0xB1) S3D_OCT_STEP(0x4E) S3D_OCT_STEP(0x141)          // quad_perm xor 1, xor 2, row_half_mirror
#undef S3D_OCT_STEP
    return __int_as_float(x);
}
__device__ __forceinline__ void bf8_to_f32(const u32x4 v, float (&o)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[2 * e] = __uint_as_float(v[e] << 16); o[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u); }
}

template <int MODE, int NS, int KS, bool KTAIL>
__global__ __launch_bounds__(256 * KS) void dgrad_kernel(const DgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3, wm = w4 >> 1, wn = w4 & 1;
    // virtual tile = (slice, row tile, column tile), slice-major; every XCD (workgroup id mod 8) takes a contiguous run, so that the
    // tiles of a row panel (same dy rows) and the row panels of a slice (same W rows) meet in one L2
    int v;
    {
        const int total = p.ntx * p.nty * p.nslice, bid = blockIdx.x;
        const int q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
        v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ntile = p.ntx * p.nty;
    const int bz = v / ntile, tile = v - bz * ntile;
    const int ty = tile / p.ntx, m0 = ty * 64, n0 = (tile % p.ntx) * 64;
    const int kbeg = bz * p.kchunk;
    const int kslice = min(p.K, kbeg + p.kchunk) - kbeg;
    // staged epilogues (DG_DGELU, DG_LNBWD; group 0 only): thread t owns columns n0 + 8 (t & 7) .. + 7 of rows (t >> 3) and (t >> 3) + 32
    const int ecol = n0 + 8 * (tid & 7);
    const bool ecol_ok = ecol < p.N;
    const int ec = min(ecol, p.N - 8);
    int erow[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) erow[it] = m0 + (tid >> 3) + 32 * it;

    // ---- what the epilogue reads, requested before the k-loop (a load issued next to its use costs its whole latency)
    u32x4 pre8[2];                                                      // DG_DGELU: saved pre-activation, 8 bf16
    f32x4 x8[2][2], r8[2][2];                                           // DG_LNBWD: LayerNorm input row, residual gradient
    float rmean[2], rrstd[2], rs1[2], rs2[2];
    if constexpr (MODE != DG_PLANES) {
        if (p.zero_buf != nullptr && (int)blockIdx.x * (256 * KS) + tid < p.zero_n) p.zero_buf[(int)blockIdx.x * (256 * KS) + tid] = 0.f;
    }
    if constexpr (MODE == DG_DGELU) {
        if (grp == 0) {
#pragma unroll
            for (int it = 0; it < 2; ++it)
                pre8[it] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p.aux + (long)min(erow[it], p.M - 1) * p.ldaux + ec));
        }
    }
    if constexpr (MODE == DG_LNBWD) {
        if (grp == 0) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const long mr = min(erow[it], p.M - 1);
                rmean[it] = p.mean[mr]; rrstd[it] = p.rstd[mr]; rs1[it] = p.in_s1[mr]; rs2[it] = p.in_s2[mr] * p.inv_d;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    x8[it][h] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.x + mr * p.ldx + ec + 4 * h));
                    r8[it][h] = p.dres ? *reinterpret_cast<const f32x4*>(p.dres + mr * p.lddres + ec + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    }

    f32x4 acc[2][2];
    dgrad_mainloop<NS, KS, KTAIL>(p, smem, m0, n0, kbeg, kslice, acc);

    constexpr int LDC = 68;
    float* ct = reinterpret_cast<float*>(smem + (KS > 1 ? 16384 : 0));   // fp32 [64][LDC] staging tile (behind the hand-over area)
    if constexpr (KS > 1) {
        f32x4* hand = reinterpret_cast<f32x4*>(smem);                  // [4 waves][2][2][64 lanes]
        __syncthreads();                                               // every wave is done with the ring
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) hand[((w4 * 2 + i) * 2 + j) * 64 + lane] = acc[i][j];
        }
        __syncthreads();
        if (grp == 1) return;                                          // (waves that have ended are not counted by later barriers)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] += hand[((w4 * 2 + i) * 2 + j) * 64 + lane];
    }

    if constexpr (MODE == DG_PLANES) {
        // lane holds row m and four consecutive columns: one 16-byte store per fragment (16 rows x 64 B per instruction)
        float* C = p.C + (long)bz * p.slice_stride;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int m = m0 + wm * 32 + i * 16 + (lane & 15), n = n0 + wn * 32 + j * 16 + (lane >> 4) * 4;
                if (m < p.M && n < p.N) *reinterpret_cast<f32x4*>(C + (long)m * p.ldc + n) = acc[i][j] * p.alpha;
            }
        return;
    }
    // park the accumulators: row-contiguous 16 / 32-byte accesses from here on
    if constexpr (KS == 1) __syncthreads();                            // every wave is done with the ring
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            *reinterpret_cast<f32x4*>(ct + (wm * 32 + i * 16 + (lane & 15)) * LDC + wn * 32 + j * 16 + (lane >> 4) * 4) = acc[i][j] * p.alpha;
    __syncthreads();
    if constexpr (MODE == DG_DGELU) {
        const bool stats = p.rs1 != nullptr;                           // uniform
        float u8[8], c8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { u8[e] = 0.f; c8[e] = 0.f; }
        if (stats && ecol_ok) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(p.st_u + ecol + 4 * h), b = *reinterpret_cast<const f32x4*>(p.st_c + ecol + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) { u8[4 * h + e] = a[e]; c8[4 * h + e] = b[e]; }
            }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int r = (tid >> 3) + 32 * it;
            float pr[8], d[8];
            bf8_to_f32(pre8[it], pr);
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(ct + r * LDC + 8 * (tid & 7)), v1 = *reinterpret_cast<const f32x4*>(ct + r * LDC + 8 * (tid & 7) + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { d[e] = v0[e] * gelu_erf_grad(pr[e]); d[4 + e] = v1[e] * gelu_erf_grad(pr[4 + e]); }
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf2(d[2 * e], d[2 * e + 1]);
            const bool ok = erow[it] < p.M && ecol_ok;
            if (ok) *reinterpret_cast<u32x4*>(p.O + (long)erow[it] * p.ldo + ecol) = o;
            if (stats) {                                               // dots of the ROUNDED gradient: what the next dgrad multiplies
                float q[8], s1 = 0.f, s2 = 0.f;
                bf8_to_f32(o, q);
                if (ok) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { s1 = fmaf(q[e], u8[e], s1); s2 = fmaf(q[e], pr[e] - c8[e], s2); }
                }
                s1 = oct_sum(s1); s2 = oct_sum(s2);
                if ((tid & 7) == 0 && erow[it] < p.M) { atomic_add_f32(p.rs1 + erow[it], s1); atomic_add_f32(p.rs2 + erow[it], s2); }
            }
        }
    } else {
        float g8[8], cg[8], cb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { g8[e] = 0.f; cg[e] = 0.f; cb[e] = 0.f; }
        if (ecol_ok) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(p.gamma + ecol + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) g8[4 * h + e] = a[e];
            }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int r = (tid >> 3) + 32 * it;
            const bool ok = erow[it] < p.M && ecol_ok;
            float d[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 dy4 = *reinterpret_cast<const f32x4*>(ct + r * LDC + 8 * (tid & 7) + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dy = dy4[e];
                    const float xh = (x8[it][h][e] - rmean[it]) * rrstd[it];
                    d[4 * h + e] = rrstd[it] * (dy * g8[4 * h + e] - rs1[it] - xh * rs2[it]) + r8[it][h][e];
                    if (ok) { cg[4 * h + e] += dy * xh; cb[4 * h + e] += dy; }
                }
            }
            if (ok) {
                if (p.dx) {
                    *reinterpret_cast<f32x4*>(p.dx + (long)erow[it] * p.lddx + ecol) = f32x4{d[0], d[1], d[2], d[3]};
                    *reinterpret_cast<f32x4*>(p.dx + (long)erow[it] * p.lddx + ecol + 4) = f32x4{d[4], d[5], d[6], d[7]};
                }
                if (p.dx_bf) {
                    u32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = f2bf2(d[2 * e], d[2 * e + 1]);
                    *reinterpret_cast<u32x4*>(p.dx_bf + (long)erow[it] * p.lddxbf + ecol) = o;
                }
            }
        }
        if (p.partial != nullptr || p.dgamma != nullptr) {             // uniform: column sums of dy xh / dy over the tile's rows
            float* red = ct + 64 * LDC;                                // [32 row pairs][2][64 columns]
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                *reinterpret_cast<f32x4*>(red + ((tid >> 3) * 2 + 0) * 64 + 8 * (tid & 7) + 4 * h) = f32x4{cg[4 * h], cg[4 * h + 1], cg[4 * h + 2], cg[4 * h + 3]};
                *reinterpret_cast<f32x4*>(red + ((tid >> 3) * 2 + 1) * 64 + 8 * (tid & 7) + 4 * h) = f32x4{cb[4 * h], cb[4 * h + 1], cb[4 * h + 2], cb[4 * h + 3]};
            }
            __syncthreads();
            if (tid < 128) {
                const int which = tid >> 6, col = tid & 63, n = n0 + col;
                float sum = 0.f;
#pragma unroll 8
                for (int rp = 0; rp < 32; ++rp) sum += red[(rp * 2 + which) * 64 + col];
                if (n < p.N) {
                    if (p.partial) p.partial[((long)ty * 2 + which) * p.N + n] = sum;
                    else atomic_add_f32((which ? p.dbeta : p.dgamma) + n, sum);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dgrad + LayerNorm backward for 192-wide layers (deit_tiny: the point path's 32 896 / 16 416 token rows), WHOLE rows per workgroup:
// dx = LayerNorm'(dy = A @ B) + dres with the row statistics taken from the tile itself -- a 64 x 192 tile IS 64 complete rows, so no
// producer-side statistics are needed (fc1 -> norm2 and qkv -> norm1 alike).  Replaces a 128 x 128-tile dgrad that writes dy as fp32
// (25 MB) plus the ln_bwd_kernel that reads it back: 30 + 29 us -> one launch (cfg-4).
// 256 threads = 2 x 2 waves of 32 x 96; k-tiles of 32: A 64 rows x 64 B + B three k-major panels of 32 x 64 = 16 KB per stage, NS stages.
struct DgradLnRowsArgs {
    const bf16_t* A; const bf16_t* B;
    long lda, ldb;
    int M, K;
    float alpha;
    const float* x; long ldx; const float* mean; const float* rstd; const float* gamma; const float* dres; long lddres;
    float* dx; long lddx; bf16_t* dx_bf; long lddxbf;
    float* partial; float* dgamma; float* dbeta;                         // partial: [row tiles][2][192] column sums (else atomics)
};

template <int NS>
__global__ __launch_bounds__(256) void dgrad_lnrows_kernel(const DgradLnRowsArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int N = 192, STAGE = 16384, PPW = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ty = blockIdx.x, m0 = ty * 64;
    const int ntiles = p.K >> 5;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    // DMA pieces of a stage (1 KB each): wave w -> A rows 16 w .. + 15 (64 B each, chunk swizzle (row >> 2) & 3) and k-rows 8 w .. + 7 of the
    // three B panels (128 B each, kmajor_swz<64>)
    const bf16_t* gpa;
    const bf16_t* gpb[3];
    {
        const int row = wave * 16 + (lane >> 2), slot = lane & 3;
        gpa = p.A + (long)min(m0 + row, p.M - 1) * p.lda + ((slot ^ ((row >> 2) & 3)) << 3);
        const int r = wave * 8 + (lane >> 3), c = lane & 7, cg = c ^ kmajor_swz<64>(r);
#pragma unroll
        for (int q = 0; q < 3; ++q) gpb[q] = p.B + (long)r * p.ldb + q * 64 + cg * 8;
    }
    const long gstep_b = 32 * p.ldb;
    auto issue = [&](int t) {
        const unsigned dst = lds0 + (unsigned)((t % NS) * STAGE + wave * 1024);
        glds16(gpa + (long)t * 32, dst);
#pragma unroll
        for (int q = 0; q < 3; ++q) glds16(gpb[q] + (long)t * gstep_b, dst + 4096 + q * 4096);
    };
    f32x4 acc[2][6];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NS - 1; ++u)
        if (u < ntiles) issue(u);
    for (int t = 0; t < ntiles; ++t) {
        if (t + NS - 1 <= ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + NS - 1 < ntiles) issue(t + NS - 1);
        const unsigned char* sA = smem + (t % NS) * STAGE;
        const unsigned char* sB = sA + 4096;
        bf16x8 a[2], b[6];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wm * 32 + i * 16 + (lane & 15);
            a[i] = *reinterpret_cast<const bf16x8*>(sA + row * 64 + (((lane >> 4) ^ ((row >> 2) & 3)) << 4));
        }
        const int kq8 = (lane >> 4) * 8;
        if (wn == 0) {                                                  // columns 0 .. 95: panel 0, first half of panel 1
            bf16x8 b4[4], b2[2];
            frags_kmajor<64, 4>(sB, 0, kq8, lane, b4);
            frags_kmajor<64, 2>(sB + 4096, 0, kq8, lane, b2);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = b4[j];
            b[4] = b2[0]; b[5] = b2[1];
        } else {                                                        // columns 96 .. 191: second half of panel 1, panel 2
            bf16x8 b4[4], b2[2];
            frags_kmajor<64, 2>(sB + 4096, 32, kq8, lane, b2);
            frags_kmajor<64, 4>(sB + 8192, 0, kq8, lane, b4);
            b[0] = b2[0]; b[1] = b2[1];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[2 + j] = b4[j];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    // park the accumulators (lane = row, four consecutive columns per fragment): complete rows of dy in LDS
    constexpr int LDC = N + 4;
    float* ct = reinterpret_cast<float*>(smem);                         // fp32 [64][LDC] = 50 KB of the 64 KB ring
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j)
            *reinterpret_cast<f32x4*>(ct + (wm * 32 + i * 16 + (lane & 15)) * LDC + wn * 96 + j * 16 + (lane >> 4) * 4) = acc[i][j] * p.alpha;
    __syncthreads();
    // thread t: columns 64 q + 8 (t & 7) .. + 7 (q = 0, 1, 2) of rows (t >> 3) and (t >> 3) + 32 -- the eight lanes of an octet hold a whole row
    const int c8 = 8 * (tid & 7);
    const float* __restrict__ xin = p.x;                                // (restrict: the second row's loads may pass the first row's stores)
    const float* __restrict__ rin = p.dres;
    float* __restrict__ dxo = p.dx;
    bf16_t* __restrict__ dxb = p.dx_bf;
    float g8[3][8], cg[3][8], cb[3][8];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + 64 * q + c8 + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) { g8[q][4 * h + e] = gm[e]; cg[q][4 * h + e] = 0.f; cb[q][4 * h + e] = 0.f; }
        }
    const float inv_n = 1.0f / (float)N;
    // both rows' inputs are requested before either is used (a load issued next to its use costs its whole latency, twice per workgroup)
    f32x4 xr[2][3][2], rr[2][3][2];
    float rmean[2], rrstd[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const long mr = min((long)m0 + (tid >> 3) + 32 * it, (long)p.M - 1);
        rmean[it] = p.mean[mr]; rrstd[it] = p.rstd[mr];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                xr[it][q][h] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xin + mr * p.ldx + 64 * q + c8 + 4 * h));
                rr[it][q][h] = rin ? *reinterpret_cast<const f32x4*>(rin + mr * p.lddres + 64 * q + c8 + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int r = (tid >> 3) + 32 * it;
        const bool ok = m0 + r < p.M;
        const float mean = rmean[it], rstd = rrstd[it];
        float dy[3][8], xh[3][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(ct + r * LDC + 64 * q + c8 + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = d4[e] * g8[q][4 * h + e], xv = (xr[it][q][h][e] - mean) * rstd;
                    dy[q][4 * h + e] = d4[e]; xh[q][4 * h + e] = xv;
                    s1 += g; s2 = fmaf(g, xv, s2);
                }
            }
        s1 = oct_sum(s1) * inv_n; s2 = oct_sum(s2) * inv_n;
        s1 = __shfl(s1, lane & ~7, 64); s2 = __shfl(s2, lane & ~7, 64);  // (oct_sum leaves the total in lane & 7 == 0 and its mirror)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float d[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = 4 * h + e;
                    d[k] = rstd * (dy[q][k] * g8[q][k] - s1 - xh[q][k] * s2) + rr[it][q][h][e];
                    if (ok) { cg[q][k] = fmaf(dy[q][k], xh[q][k], cg[q][k]); cb[q][k] += dy[q][k]; }
                }
            }
            if (ok) {
                if (dxo) {
                    *reinterpret_cast<f32x4*>(dxo + (long)(m0 + r) * p.lddx + 64 * q + c8) = f32x4{d[0], d[1], d[2], d[3]};
                    *reinterpret_cast<f32x4*>(dxo + (long)(m0 + r) * p.lddx + 64 * q + c8 + 4) = f32x4{d[4], d[5], d[6], d[7]};
                }
                if (dxb) {
                    u32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = f2bf2(d[2 * e], d[2 * e + 1]);
                    *reinterpret_cast<u32x4*>(dxb + (long)(m0 + r) * p.lddxbf + 64 * q + c8) = o;
                }
            }
        }
    }
    if (p.partial != nullptr || p.dgamma != nullptr) {                 // uniform: column sums of dy xh / dy over the tile's rows
        __syncthreads();                                               // every thread has read its rows of ct
        float* red = reinterpret_cast<float*>(smem);                    // [32 row pairs][2][192]
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                *reinterpret_cast<f32x4*>(red + ((tid >> 3) * 2 + 0) * N + 64 * q + c8 + 4 * h) = f32x4{cg[q][4 * h], cg[q][4 * h + 1], cg[q][4 * h + 2], cg[q][4 * h + 3]};
                *reinterpret_cast<f32x4*>(red + ((tid >> 3) * 2 + 1) * N + 64 * q + c8 + 4 * h) = f32x4{cb[q][4 * h], cb[q][4 * h + 1], cb[q][4 * h + 2], cb[q][4 * h + 3]};
            }
        __syncthreads();
        for (int idx = tid; idx < 2 * N; idx += 256) {
            const int which = idx / N, col = idx % N;
            float sum = 0.f;
#pragma unroll 8
            for (int rp = 0; rp < 32; ++rp) sum += red[(rp * 2 + which) * N + col];
            if (p.partial) p.partial[((long)ty * 2 + which) * N + col] = sum;
            else atomic_add_f32((which ? p.dbeta : p.dgamma) + col, sum);
        }
    }
}

// ---- weights-only vectors of the row statistics (see above): for every row k of a Linear weight W [K][D] that follows a LayerNorm
//      (gamma, beta):  u[k] = mean_n(W_hi[k][n] gamma[n])  (the dgrad multiplies by the HIGH plane),  c[k] = b[k] + sum_n (W_hi + W_lo)[k][n] beta[n]
//      (the forward's pre-activation was computed from both planes).  One wave per row, eight rows per workgroup.
struct LnAuxLayer { const bf16_t* w_hi; const bf16_t* w_lo; const float* bias; const float* gamma; const float* beta; float* u; float* c; int K; };
struct LnAuxArgs { LnAuxLayer l[32]; int n, D; };
__global__ __launch_bounds__(256) void ln_aux_kernel(const LnAuxArgs a) {
    const LnAuxLayer& L = a.l[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_d = 1.0f / (float)a.D;
    // a wave's four rows together: eight 16-byte loads in flight per lane and trip (the launch reads 50 MB of weight planes at cfg-2 and is on
    // the step's launch chain; one row at a time it ran at 3.3 TB/s)
    constexpr int R = 4;
    const int k0 = ((int)blockIdx.x * 4 + wave) * R;
    if (k0 >= L.K) return;
    float su[R], sc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { su[r] = 0.f; sc[r] = 0.f; }
    for (int n = lane * 8; n < a.D; n += 512) {
        U128 h[R], l[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long kr = min(k0 + r, L.K - 1);
            h[r].u = *reinterpret_cast<const u32x4*>(L.w_hi + kr * a.D + n);
            l[r].u = *reinterpret_cast<const u32x4*>(L.w_lo + kr * a.D + n);
        }
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(L.gamma + n), g1 = *reinterpret_cast<const f32x4*>(L.gamma + n + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(L.beta + n), b1 = *reinterpret_cast<const f32x4*>(L.beta + n + 4);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float wh = bf2f(h[r].h[e]);
                su[r] = fmaf(wh, e < 4 ? g0[e] : g1[e - 4], su[r]);
                sc[r] = fmaf(wh + bf2f(l[r].h[e]), e < 4 ? b0[e] : b1[e - 4], sc[r]);
            }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { su[r] = wave_sum(su[r]); sc[r] = wave_sum(sc[r]); }
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (k0 + r < L.K) { L.u[k0 + r] = su[r] * inv_d; L.c[k0 + r] = sc[r] + (L.bias ? L.bias[k0 + r] : 0.f); }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// grouped wgrad: for every problem i,  dW_i[Mo][No] (+)= alpha * dy_i[K][Mo]^T x_i[K][No],  db_i[Mo] (+)= alpha * colsum(dy_i)
constexpr int WG_MAX = 48;                                             // 48 x 64-byte entries: the launch's kernel arguments stay under 4 KB
struct WgProb {
    const bf16_t* dy; const bf16_t* x; float* dW; float* db;
    int ld_dy, ld_x, ldw;
    int Mo, No, ntx, tile0, pad_;                                            // tiles [tile0, tile0 + ntx * nty) of the launch
};
struct WgGroup {
    WgProb p[WG_MAX];
    int n, K, total;
    float alpha;
    int beta;                                                          // 1: accumulate into dW / db, 0: overwrite
};

// optimizer shares riding on the launch (adam_fill.h): ranges of the arena whose gradients are final -- the blocks of the PREVIOUS group --
// updated by `blocks` filler workgroups behind the tiles.  The tiles are paced by LDS / the L2 -> LDS path with the HBM nearly idle, the
// update is a pure HBM stream without LDS: the two share a CU without competing for the same resource.
struct WgFill { AdamFill f[6]; int n, blocks; };

template <int NS, int BK, bool KTAIL>
__global__ __launch_bounds__(256) void wgrad_group_kernel(const WgGroup g, const WgFill fill) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((int)blockIdx.x >= g.total) {                                   // filler workgroup: every range, grid-strided over the fillers
        const int fb = (int)blockIdx.x - g.total;
        for (int r = 0; r < fill.n; ++r) {
            AdamFill f = fill.f[r];
            f.blocks = fill.blocks;
            adam_fill_run(f, fb);
        }
        return;
    }
    static_assert(BK == 32 || BK == 64, "k-tiles of 32 or 64 rows");
    constexpr int A_BYTES = BK * 256, STAGE = 2 * A_BYTES, PPW = BK / 8;   // A and B: BK k-rows x 128 columns each, BK / 4 pieces each
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int v;
    {
        const int bid = blockIdx.x, q = g.total >> 3, r = g.total & 7, xcd = bid & 7, idx = bid >> 3;
        v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int pi = 0;
    while (pi + 1 < g.n && v >= g.p[pi + 1].tile0) ++pi;               // uniform
    const WgProb& pr = g.p[pi];
    const int tile = v - pr.tile0;
    const int tx = tile % pr.ntx, m0 = (tile / pr.ntx) * 128, n0 = tx * 128;
    const int K = g.K;
    const int ntiles = (K + BK - 1) / BK, ktail = KTAIL ? (K % BK) : 0;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);

    // waves 0, 1: the dy pieces (4 k-rows x 256 B each); waves 2, 3: the x pieces
    const bf16_t* gp[PPW];
    int gk[KTAIL ? PPW : 1];
    const bool isB = wave >= 2;
    const bf16_t* base = isB ? pr.x : pr.dy;
    const long ld = isB ? pr.ld_x : pr.ld_dy;
    const int R = isB ? pr.No : pr.Mo, r0 = isB ? n0 : m0;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int q = (wave & 1) * PPW + j;                            // piece 0 .. BK / 4 - 1 of the operand
        const int r = q * 4 + (lane >> 4), c = lane & 15;
        const int cg = c ^ kmajor_swz<128>(r);
        gp[j] = base + (long)r * ld + min(r0 + cg * 8, R - 8);
        if constexpr (KTAIL) gk[j] = r;
    }
    const long gstep = BK * ld;
    auto issue = [&](int t) {
        const unsigned dst = lds0 + (unsigned)((t % NS) * STAGE + wave * PPW * 1024);
        if constexpr (KTAIL) {
            if (ktail != 0 && t == ntiles - 1) {
#pragma unroll
                for (int j = 0; j < PPW; ++j)
                    glds16(gk[j] < ktail ? gp[j] + (long)t * gstep : reinterpret_cast<const bf16_t*>(g_bwd_zeros), dst + j * 1024);
                return;
            }
        }
#pragma unroll
        for (int j = 0; j < PPW; ++j) glds16(gp[j] + (long)t * gstep, dst + j * 1024);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias gradient: column sums of dy = dy^T . ones, one extra MFMA per dy fragment in the first tile column
    const bool want_bsum = pr.db != nullptr && tx == 0 && wn == 0;     // wave-uniform
    f32x4 bacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    U128 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones.h[i] = (bf16_t)0x3F80;

#pragma unroll
    for (int u = 0; u < NS - 1; ++u)
        if (u < ntiles) issue(u);
    for (int t = 0; t < ntiles; ++t) {
        if (t + NS - 1 <= ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + NS - 1 < ntiles) issue(t + NS - 1);
        const unsigned char* sA = smem + (t % NS) * STAGE;
        const unsigned char* sB = sA + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 a[4], b[4];
            frags_kmajor_ab<128, 4>(sA, wm * 64, sB, wn * 64, ks * 32 + (lane >> 4) * 8, lane, a, b);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            if (want_bsum) {
#pragma unroll
                for (int i = 0; i < 4; ++i) bacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], ones.v, bacc[i], 0, 0, 0);
            }
        }
    }
    // one writer per element: plain 16-byte read-modify-write (lane = row m of dW, four consecutive columns)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + (lane & 15);
        f32x4 old[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            old[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (g.beta && m < pr.Mo && n < pr.No) old[j] = *reinterpret_cast<const f32x4*>(pr.dW + (long)m * pr.ldw + n);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            if (m < pr.Mo && n < pr.No) *reinterpret_cast<f32x4*>(pr.dW + (long)m * pr.ldw + n) = old[j] + acc[i][j] * g.alpha;
        }
    }
    if (want_bsum && (lane & 15) == 0) {                               // bacc: every column equal; lane 16 q holds rows 4 q .. 4 q + 3
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
                if (m < pr.Mo) pr.db[m] = (g.beta ? pr.db[m] : 0.f) + bacc[i][r] * g.alpha;
            }
    }
}

template <typename K>
void set_lds_once(K kern, int bytes, bool& done) {
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        done = true;
    }
}

}  // namespace

bool s3d_dgrad_splitk_ok(const GemmArgs& a) {
    return a.M > 0 && a.N >= 8 && a.K >= 8 && (a.K & 7) == 0 && (a.N & 7) == 0 && (a.lda & 7) == 0 && (a.ldb & 7) == 0 && (a.ldc & 3) == 0 &&
           a.C != nullptr && a.A_hi != nullptr && a.B_hi != nullptr && a.bias == nullptr;
}

// k-slices a request for `want` slices of K really gives (slices are whole 64-tiles): the count the consumer has to add
int s3d_dgrad_splitk_slices(int K, int want) {
    if (want < 1) want = 1;
    if (want > 4) want = 4;
    const int kchunk = ((K + want - 1) / want + 63) / 64 * 64;
    return (K + kchunk - 1) / kchunk;
}

namespace {
template <int MODE, int KS>
int launch_dgrad_ks(const DgradArgs& d, hipStream_t s) {
    const bool ktail = (d.K & 63) != 0;
    constexpr int NS = 3, LDS = NS * 16384 * KS;
    static bool set0 = false, set1 = false;
    const dim3 grid((unsigned)(d.ntx * d.nty * d.nslice));
    if (ktail) {
        set_lds_once(dgrad_kernel<MODE, NS, KS, true>, LDS, set1);
        hipLaunchKernelGGL((dgrad_kernel<MODE, NS, KS, true>), grid, dim3(256 * KS), LDS, s, d);
    } else {
        set_lds_once(dgrad_kernel<MODE, NS, KS, false>, LDS, set0);
        hipLaunchKernelGGL((dgrad_kernel<MODE, NS, KS, false>), grid, dim3(256 * KS), LDS, s, d);
    }
    return 0;
}
template <int MODE>
int launch_dgrad(DgradArgs& d, long long key, double flops, const char* name, int cov, hipStream_t s) {
    d.ntx = (d.N + 63) / 64; d.nty = (d.M + 63) / 64;
    if (s3d_prof_skipped(key)) return 0;
    // two wave groups when a k-slice is a long chain and the grid leaves CUs to spare (profiles/r05_dgrad_wave_groups.txt)
    static const int ks_env = s3d_tune_int("S3D_DGRAD_KS");
    const int chain = (d.kchunk + 63) / 64;
    const bool two = ks_env > 0 ? ks_env == 2 : (chain >= 8 && (long)d.ntx * d.nty * d.nslice <= 320);
    s3d_prof_begin(key, flops, s);
    if (two) launch_dgrad_ks<MODE, 2>(d, s);
    else launch_dgrad_ks<MODE, 1>(d, s);
    s3d_prof_end(s);
    S3D_CHECK_LAUNCH_V(name, cov * 100 + (two ? 10 : 0) + ((d.K & 63) != 0 ? 1 : 0));
    return 0;
}
DgradArgs dgrad_base(const GemmArgs& a) {
    DgradArgs d;
    memset(&d, 0, sizeof(d));
    d.A = a.A_hi; d.B = a.B_hi; d.lda = a.lda; d.ldb = a.ldb; d.M = a.M; d.N = a.N; d.K = a.K; d.alpha = a.alpha;
    d.kchunk = (a.K + 63) / 64 * 64; d.nslice = 1;
    return d;
}
bool dgrad_operands_ok(const GemmArgs& a) {
    return a.M > 0 && a.N >= 8 && a.K >= 8 && (a.K & 7) == 0 && (a.N & 7) == 0 && (a.lda & 7) == 0 && (a.ldb & 7) == 0 && a.A_hi != nullptr &&
           a.B_hi != nullptr && a.bias == nullptr;
}
}  // namespace

int s3d_launch_dgrad_splitk(const GemmArgs& a, int nslice, long slice_stride, hipStream_t s) {
    S3D_REQUIRE(s3d_dgrad_splitk_ok(a), "dgrad_splitk: M=%d N=%d K=%d lda=%ld ldb=%ld ldc=%ld: K, N, lda, ldb multiples of 8, fp32 output C, no bias",
                a.M, a.N, a.K, a.lda, a.ldb, a.ldc);
    S3D_REQUIRE(nslice >= 1 && nslice <= 4 && (nslice == 1 || slice_stride >= (long)a.M * a.ldc), "dgrad_splitk: nslice=%d (1 .. 4), slice_stride=%ld", nslice,
                slice_stride);
    DgradArgs d = dgrad_base(a);
    d.C = a.C; d.ldc = a.ldc; d.slice_stride = slice_stride;
    d.kchunk = ((a.K + nslice - 1) / nslice + 63) / 64 * 64;
    d.nslice = (a.K + d.kchunk - 1) / d.kchunk;                         // (never more than asked for)
    S3D_REQUIRE(d.nslice == nslice, "dgrad_splitk: K=%d does not give %d non-empty slices of whole 64-tiles (the consumer adds exactly nslice planes)", a.K, nslice);
    return launch_dgrad<DG_PLANES>(d, 1100000000000LL + 64064, 2.0 * a.M * a.N * a.K, "dgrad_splitk", nslice, s);   // bench.py: 11 = split-K dgrad
}

// dh = bf16((dy @ W) * gelu'(aux)) (a.O_hi, a.aux), and -- when st->rs1 is set -- the LayerNorm row statistics of dh (see the kernel)
int s3d_launch_dgrad_dgelu(const GemmArgs& a, const S3dRowStats* st, hipStream_t s) {
    S3D_REQUIRE(dgrad_operands_ok(a) && a.O_hi && a.aux && (a.ldo & 3) == 0 && (a.ldaux & 3) == 0 && a.aux_lo == nullptr && a.O_lo == nullptr,
                "dgrad_dgelu: M=%d N=%d K=%d: K, N, lda, ldb multiples of 8; bf16 output O_hi and saved pre-activation aux required", a.M, a.N, a.K);
    DgradArgs d = dgrad_base(a);
    d.aux = a.aux; d.ldaux = a.ldaux; d.O = a.O_hi; d.ldo = a.ldo;
    if (st) {
        S3D_REQUIRE(st->rs1 == nullptr || (st->u && st->c && st->rs2), "dgrad_dgelu: row statistics need u, c, rs1 and rs2");
        d.st_u = st->u; d.st_c = st->c; d.rs1 = st->rs1; d.rs2 = st->rs2; d.zero_buf = st->zero_buf; d.zero_n = st->zero_n;
        S3D_REQUIRE(st->zero_n <= ((a.N + 63) / 64) * ((a.M + 63) / 64) * 256, "dgrad_dgelu: zero_n=%d exceeds the launch's threads", st->zero_n);
    }
    return launch_dgrad<DG_DGELU>(d, 1200000000000LL + 64064, 2.0 * a.M * a.N * a.K, "dgrad_dgelu", st && st->rs1 ? 1 : 0, s);   // bench.py: 12
}

// dx = LayerNorm'(dy = a.A @ a.B) + dres as the dgrad's epilogue; ln->dy is ignored, the row statistics come from st->rs1 / rs2
int s3d_launch_dgrad_lnbwd(const GemmArgs& a, const LnBwdArgs& ln, const S3dRowStats* st, hipStream_t s) {
    S3D_REQUIRE(dgrad_operands_ok(a), "dgrad_lnbwd: M=%d N=%d K=%d: K, N, lda, ldb multiples of 8", a.M, a.N, a.K);
    S3D_REQUIRE(st && st->rs1 && st->rs2, "dgrad_lnbwd: the row statistics of the producer are required");
    S3D_REQUIRE(ln.D == a.N && ln.rows == a.M && ln.x && ln.mean && ln.rstd && ln.gamma && (ln.dx || ln.dx_bf) && ln.drop_thr == 0 && ln.dx_bf_lo == nullptr &&
                    (ln.ldx & 3) == 0 && (ln.lddres & 3) == 0 && (ln.lddx & 3) == 0 && (ln.lddxbf & 3) == 0,
                "dgrad_lnbwd: LayerNorm over the dgrad's %d columns / %d rows (got D=%d rows=%ld), no dropout, no lo plane", a.N, a.M, ln.D, ln.rows);
    DgradArgs d = dgrad_base(a);
    d.x = ln.x; d.ldx = ln.ldx; d.mean = ln.mean; d.rstd = ln.rstd; d.gamma = ln.gamma; d.dres = ln.dres; d.lddres = ln.lddres;
    d.dx = ln.dx; d.lddx = ln.lddx; d.dx_bf = ln.dx_bf; d.lddxbf = ln.lddxbf;
    d.partial = ln.partial; d.dgamma = ln.dgamma; d.dbeta = ln.dbeta;
    if (ln.partial) S3D_REQUIRE(ln.partial_blocks >= (a.M + 63) / 64, "dgrad_lnbwd: the partial buffer needs >= %d rows of [2][D]", (a.M + 63) / 64);
    d.in_s1 = st->rs1; d.in_s2 = st->rs2; d.inv_d = 1.0f / (float)a.N; d.zero_buf = st->zero_buf; d.zero_n = st->zero_n;
    S3D_REQUIRE(st->zero_n <= ((a.N + 63) / 64) * ((a.M + 63) / 64) * 256, "dgrad_lnbwd: zero_n=%d exceeds the launch's threads", st->zero_n);
    return launch_dgrad<DG_LNBWD>(d, 1300000000000LL + 64064, 2.0 * a.M * a.N * a.K, "dgrad_lnbwd", ln.partial ? 1 : 0, s);      // bench.py: 13
}

// dx = LayerNorm'(dy = a.A @ a.B) + dres for 192-wide layers, whole rows per workgroup (no producer statistics); ln->dy is ignored
bool s3d_dgrad_lnrows_ok(const GemmArgs& a, const LnBwdArgs& ln) {
    return a.N == 192 && ln.D == 192 && a.M > 0 && ln.rows == a.M && a.K >= 32 && (a.K & 31) == 0 && (a.lda & 7) == 0 && (a.ldb & 7) == 0 && a.A_hi && a.B_hi &&
           a.bias == nullptr && ln.x && ln.mean && ln.rstd && ln.gamma && (ln.dx || ln.dx_bf) && ln.drop_thr == 0 && ln.dx_bf_lo == nullptr && ln.dy_parts == 0 &&
           (ln.ldx & 3) == 0 && (ln.lddres & 3) == 0 && (ln.lddx & 3) == 0 && (ln.lddxbf & 7) == 0 && (ln.partial == nullptr || ln.partial_blocks >= (a.M + 63) / 64);
}
int s3d_launch_dgrad_lnrows(const GemmArgs& a, const LnBwdArgs& ln, hipStream_t s) {
    S3D_REQUIRE(s3d_dgrad_lnrows_ok(a, ln), "dgrad_lnrows: M=%d N=%d K=%d: 192 columns, K a multiple of 32, a LayerNorm over the same rows", a.M, a.N, a.K);
    DgradLnRowsArgs d;
    memset(&d, 0, sizeof(d));
    d.A = a.A_hi; d.B = a.B_hi; d.lda = a.lda; d.ldb = a.ldb; d.M = a.M; d.K = a.K; d.alpha = a.alpha;
    d.x = ln.x; d.ldx = ln.ldx; d.mean = ln.mean; d.rstd = ln.rstd; d.gamma = ln.gamma; d.dres = ln.dres; d.lddres = ln.lddres;
    d.dx = ln.dx; d.lddx = ln.lddx; d.dx_bf = ln.dx_bf; d.lddxbf = ln.lddxbf;
    d.partial = ln.partial; d.dgamma = ln.dgamma; d.dbeta = ln.dbeta;
    constexpr long long KEY = 1400000000000LL + 64192;                  // bench.py: 14 = dgrad + LayerNorm backward, whole rows
    if (s3d_prof_skipped(KEY)) return 0;
    s3d_prof_begin(KEY, 2.0 * a.M * a.N * a.K, s);
    // three stages = 48 KB of ring, 50 KB with the fp32 row tile of the epilogue: THREE workgroups per CU, so that cfg-4's 514 tiles are one
    // round of resident workgroups (at four stages = two per CU, 512 slots, the last two tiles ran alone: 45 us per launch)
    constexpr int NS = 3, LDS = 64 * 196 * 4;
    static_assert(LDS >= NS * 16384, "ring fits");
    static bool set = false;
    set_lds_once(dgrad_lnrows_kernel<NS>, LDS, set);
    hipLaunchKernelGGL((dgrad_lnrows_kernel<NS>), dim3((unsigned)((a.M + 63) / 64)), dim3(256), LDS, s, d);
    s3d_prof_end(s);
    S3D_CHECK_LAUNCH_V("dgrad_lnrows", ln.partial ? 1 : 0);
    return 0;
}

int s3d_launch_ln_aux(const S3dLnAuxLayer* layers, int n, int D, hipStream_t s) {
    S3D_REQUIRE(layers != nullptr && n >= 1 && n <= 32 && D >= 8 && (D & 7) == 0, "ln_aux: 1 .. 32 layers, D a multiple of 8 (n=%d, D=%d)", n, D);
    LnAuxArgs a;
    memset(&a, 0, sizeof(a));
    int kmax = 0;
    for (int i = 0; i < n; ++i) {
        const S3dLnAuxLayer& q = layers[i];
        S3D_REQUIRE(q.w_hi && q.w_lo && q.gamma && q.beta && q.u && q.c && q.K > 0, "ln_aux: layer %d: weight planes, gamma, beta, u, c required", i);
        a.l[i] = LnAuxLayer{q.w_hi, q.w_lo, q.bias, q.gamma, q.beta, q.u, q.c, q.K};
        kmax = q.K > kmax ? q.K : kmax;
    }
    a.n = n; a.D = D;
    hipLaunchKernelGGL(ln_aux_kernel, dim3((unsigned)((kmax + 15) / 16), (unsigned)n), dim3(256), 0, s, a);
    S3D_CHECK_LAUNCH("ln_aux");
    return 0;
}

int s3d_launch_wgrad_group(const S3dWgradItem* it, int n, int K, float alpha, int accumulate, hipStream_t s, const AdamFill* fills, int nfill) {
    S3D_REQUIRE(it != nullptr && n >= 1 && n <= WG_MAX, "wgrad_group: 1 .. %d problems per launch (got %d)", WG_MAX, n);
    S3D_REQUIRE(K >= 1, "wgrad_group: K=%d rows", K);
    WgGroup g;
    memset(&g, 0, sizeof(g));
    int total = 0;
    double flops = 0.0;
    for (int i = 0; i < n; ++i) {
        const S3dWgradItem& q = it[i];
        S3D_REQUIRE(q.dy && q.x && q.dW && q.out >= 8 && q.in >= 8 && (q.out & 7) == 0 && (q.in & 7) == 0 && (q.ld_dy & 7) == 0 && (q.ld_x & 7) == 0 &&
                        (q.ldw & 3) == 0 && q.ld_dy >= q.out && q.ld_x >= q.in && q.ldw >= q.in && q.ld_dy < (1L << 31) && q.ld_x < (1L << 31) && q.ldw < (1L << 31),
                    "wgrad_group: problem %d: out=%d in=%d ld_dy=%ld ld_x=%ld ldw=%ld (multiples of 8; ldw of 4)", i, q.out, q.in, q.ld_dy, q.ld_x, q.ldw);
        WgProb& p = g.p[i];
        p.dy = q.dy; p.x = q.x; p.dW = q.dW; p.db = q.db; p.ld_dy = (int)q.ld_dy; p.ld_x = (int)q.ld_x; p.ldw = (int)q.ldw; p.Mo = q.out; p.No = q.in;
        p.ntx = (q.in + 127) / 128; p.tile0 = total;
        total += p.ntx * ((q.out + 127) / 128);
        flops += 2.0 * q.out * q.in * (double)K;
    }
    g.n = n; g.K = K; g.total = total; g.alpha = alpha; g.beta = accumulate ? 1 : 0;
    WgFill fill;
    memset(&fill, 0, sizeof(fill));
    if (fills != nullptr && nfill > 0) {
        S3D_REQUIRE(nfill <= 6, "wgrad_group: at most 6 optimizer ranges per launch (got %d)", nfill);
        long n4 = 0;
        for (int i = 0; i < nfill; ++i) { fill.f[i] = fills[i]; n4 += fills[i].n4; }
        fill.n = nfill;
        static const int fb_env = s3d_tune_int("S3D_WGRAD_FILL_BLOCKS");
        long blocks = fb_env > 0 ? fb_env : 256;
        if (blocks > (n4 + 255) / 256) blocks = (n4 + 255) / 256;
        fill.blocks = (int)blocks;
    }
    const unsigned grid = (unsigned)(total + fill.blocks);
    // variant = ring depth x k-tile: in-flight bytes per CU are what paces this launch (every operand byte is a first touch from the
    // Infinity Cache / HBM at ~1.5 us; profiles/r05_wgrad_group_variants.txt)
    static const int v_env = s3d_tune_int("S3D_WGRAD_VARIANT");
    const int variant = v_env >= 0 ? v_env : 0;
    constexpr long long KEY = 1000000000000LL + 128128;                 // bench.py: 10 = grouped full-k wgrad, 128 x 128 tiles
    if (s3d_prof_skipped(KEY)) return 0;
    s3d_prof_begin(KEY, flops, s);
#define S3D_WG_LAUNCH(NS_, BK_)                                                                                              \
    do {                                                                                                                     \
        constexpr int LDS = NS_ * BK_ * 512;                                                                                 \
        static bool set0 = false, set1 = false;                                                                              \
        if ((K % BK_) != 0) {                                                                                                \
            set_lds_once(wgrad_group_kernel<NS_, BK_, true>, LDS, set1);                                                     \
            hipLaunchKernelGGL((wgrad_group_kernel<NS_, BK_, true>), dim3(grid), dim3(256), LDS, s, g, fill);               \
        } else {                                                                                                             \
            set_lds_once(wgrad_group_kernel<NS_, BK_, false>, LDS, set0);                                                    \
            hipLaunchKernelGGL((wgrad_group_kernel<NS_, BK_, false>), dim3(grid), dim3(256), LDS, s, g, fill);              \
        }                                                                                                                    \
    } while (0)
    // measured (profiles/r05_wgrad_group_variants.txt; 324 tiles = three cfg-2 blocks per launch): two workgroups per CU beat one with a
    // deeper ring (39 - 41 us against 52: the second resident workgroup also absorbs the 324 / 256 tile quantisation)
    switch (variant) {
        case 1: S3D_WG_LAUNCH(2, 64); break;          // 64 KB: two workgroups per CU, one stage in flight each           40.7 us
        case 2: S3D_WG_LAUNCH(3, 64); break;          // 96 KB: one workgroup per CU, two stages in flight                52.4 us
        case 3: S3D_WG_LAUNCH(5, 32); break;          // 80 KB: two workgroups per CU, four 16 KB stages in flight each   41.4 us
        case 5: S3D_WG_LAUNCH(4, 64); break;          // 128 KB: one workgroup per CU, three stages in flight             51.6 us
        default: S3D_WG_LAUNCH(4, 32); break;         // 64 KB: two workgroups per CU, three 16 KB stages in flight each  39.3 us
    }
#undef S3D_WG_LAUNCH
    s3d_prof_end(s);
    const bool ktail = (K & 63) != 0;
    S3D_CHECK_LAUNCH_V("wgrad_group", (ktail ? 1 : 0) + (fill.n ? 10 : 0));
    return 0;
}
