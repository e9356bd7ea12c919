// LayerNorm forward / backward for gfx950: one 64-lane wave per row, float4 accesses, fp32 statistics.
// Forward emits the normalised row as split-bf16 planes (GEMM A operand) and optionally fp32; backward fuses the
// residual-gradient add, a bf16 copy of dx (operand of the next wgrad/dgrad GEMM) and the gamma/beta gradients.
#include "kernels.h"
#include "adam_fill.h"
#include "ln_row.h"

#include <stdlib.h>

namespace {

constexpr int MAXC = 4;   // float4 chunks per lane -> D <= 1024
// The kernels are templated on the chunk count MC actually needed (D <= 256 / 512 / 1024): with MC = 4 a D = 192 row (deit_tiny,
// 33 k - 188 k rows per launch on the point and group_embed paths) still issued four clamped loads per array and held 176 VGPRs
// = two waves per SIMD, i.e. half the HBM rate the same kernel reaches at D = 768.

template <int MC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnArgs p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const float* x = p.x + row * p.ldx;
    float4 v[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) {
        const int col = c * 256 + lane * 4;
        const float4 t = *reinterpret_cast<const float4*>(x + min(col, p.D - 4));      // unconditional, clamped
        const float keep = (col < p.D) ? 1.f : 0.f;
        v[c] = make_float4(t.x * keep, t.y * keep, t.z * keep, t.w * keep);
    }
    ln_row_finish<MC>(v, lane, p.D, p.eps, p.gamma, p.beta, p.mean ? p.mean + row : nullptr, p.rstd ? p.rstd + row : nullptr,
                      p.out_hi ? p.out_hi + row * p.ldo : nullptr, p.out_lo ? p.out_lo + row * p.ldo : nullptr,
                      p.out_f32 ? p.out_f32 + row * p.ldo : nullptr);
}

// Backward: each wave owns RPW consecutive-by-stride rows and issues ALL their loads before any reduction, so it pays one
// memory latency instead of RPW (the kernel is latency-bound at M ~ 1.7k rows); dgamma/dbeta partials are reduced across the
// block's waves in LDS and added with one atomic per column per block (same-address atomics serialise at ~12 ns each, so the
// block count is kept at rows/16).
constexpr int MAX_RPW = 4;

template <int RPW, int MC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwdArgs p, const AdamFill fill) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nmain = (int)gridDim.x - fill.blocks;       // workgroups behind the main grid run a share of the optimizer update (adam_fill.h)
    if ((int)blockIdx.x >= nmain) { adam_fill_run(fill, (int)blockIdx.x - nmain); return; }
    float* red = reinterpret_cast<float*>(smem);          // [2][4 waves][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nw = (long)nmain * 4;
    float4 dg[MC], db[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) dg[c] = db[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int nchunk = (p.D + 255) / 256;                 // float4 chunks per lane actually used (<= MC)
    const float inv_d = 1.0f / (float)p.D;

    for (long row0 = (long)blockIdx.x * 4 + wave; row0 < p.rows; row0 += nw * RPW) {
        float4 X[RPW][MC], DY[RPW][MC], R[RPW][MC];
        float mean[RPW], rstd[RPW];
        // phase 1: every load of every row of this batch
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const long row = min(row0 + (long)i * nw, p.rows - 1);      // clamped duplicate rows are not stored
            mean[i] = p.mean[row];
            rstd[i] = p.rstd[row];
#pragma unroll
            for (int c = 0; c < MC; ++c) {
                const int col = min(c * 256 + lane * 4, p.D - 4);
                if (c < nchunk) {
                    {   // the saved block input is dead after this read: non-temporal, so that it does not outlive live data in L2
                        const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.x + row * p.ldx + col));
                        X[i][c] = make_float4(t[0], t[1], t[2], t[3]);
                    }
                    DY[i][c] = *reinterpret_cast<const float4*>(p.dy + row * p.lddy + col);
                    if (p.dy_parts > 1) {                   // uniform: k-slices of a split-K dgrad, added in plane order (all loads in flight together)
                        float4 e[3];
#pragma unroll
                        for (int k = 1; k < 4; ++k)
                            if (k < p.dy_parts) e[k - 1] = *reinterpret_cast<const float4*>(p.dy + (long)k * p.dy_part_stride + row * p.lddy + col);
#pragma unroll
                        for (int k = 1; k < 4; ++k)
                            if (k < p.dy_parts) { DY[i][c].x += e[k - 1].x; DY[i][c].y += e[k - 1].y; DY[i][c].z += e[k - 1].z; DY[i][c].w += e[k - 1].w; }
                    }
                    R[i][c] = p.dres ? *reinterpret_cast<const float4*>(p.dres + row * p.lddres + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
        // phase 2: per-row reductions and stores
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const long row = row0 + (long)i * nw;
            const bool live = row < p.rows;
            float4 xh[MC], g[MC];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int c = 0; c < MC; ++c) {
                const int col = c * 256 + lane * 4;
                if (c < nchunk && col < p.D) {
                    const float4 w = *reinterpret_cast<const float4*>(p.gamma + col);
                    const float4 x = X[i][c], dy = DY[i][c];
                    xh[c] = make_float4((x.x - mean[i]) * rstd[i], (x.y - mean[i]) * rstd[i], (x.z - mean[i]) * rstd[i], (x.w - mean[i]) * rstd[i]);
                    g[c] = make_float4(dy.x * w.x, dy.y * w.y, dy.z * w.z, dy.w * w.w);
                    s1 += g[c].x + g[c].y + g[c].z + g[c].w;
                    s2 += g[c].x * xh[c].x + g[c].y * xh[c].y + g[c].z * xh[c].z + g[c].w * xh[c].w;
                    if (live) {
                        dg[c].x += dy.x * xh[c].x; dg[c].y += dy.y * xh[c].y; dg[c].z += dy.z * xh[c].z; dg[c].w += dy.w * xh[c].w;
                        db[c].x += dy.x; db[c].y += dy.y; db[c].z += dy.z; db[c].w += dy.w;
                    }
                }
            }
            s1 = wave_sum(s1) * inv_d;                  // (a division here is ~10 instructions on every lane, twice per row)
            s2 = wave_sum(s2) * inv_d;
            if (!live) continue;                          // wave-uniform
#pragma unroll
            for (int c = 0; c < MC; ++c) {
                const int col = c * 256 + lane * 4;
                if (c < nchunk && col < p.D) {
                    const float4 r = R[i][c];
                    float d[4] = {rstd[i] * (g[c].x - s1 - xh[c].x * s2) + r.x, rstd[i] * (g[c].y - s1 - xh[c].y * s2) + r.y,
                                  rstd[i] * (g[c].z - s1 - xh[c].z * s2) + r.z, rstd[i] * (g[c].w - s1 - xh[c].w * s2) + r.w};
                    if (p.dx) *reinterpret_cast<float4*>(p.dx + row * p.lddx + col) = make_float4(d[0], d[1], d[2], d[3]);
                    if (p.dx_bf) {
                        union { uint2 u; bf16_t h[4]; } o;
                        if (p.drop_thr) {                  // masked branch gradient of a post-norm residual
                            const unsigned long long key = drop_key(p.drop_seed, p.drop_site);
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                d[k] = drop_keep(key, (unsigned long long)row * p.D + col + k, p.drop_thr) ? d[k] * p.drop_scale : 0.f;
                        }
                        if (p.dx_bf_lo) {                  // parity mode: the gradient as a hi + lo pair
                            union { uint2 u; bf16_t h[4]; } ol;
#pragma unroll
                            for (int k = 0; k < 4; ++k) split_bf16(d[k], o.h[k], ol.h[k]);
                            *reinterpret_cast<uint2*>(p.dx_bf_lo + row * p.lddxbf + col) = ol.u;
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k) o.h[k] = f2bf(d[k]);
                        }
                        *reinterpret_cast<uint2*>(p.dx_bf + row * p.lddxbf + col) = o.u;
                    }
                }
            }
        }
    }
    if (p.dgamma == nullptr && p.partial == nullptr) return;   // uniform
#pragma unroll
    for (int c = 0; c < MC; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < p.D) {
            *reinterpret_cast<float4*>(red + (0 * 4 + wave) * p.D + col) = dg[c];
            *reinterpret_cast<float4*>(red + (1 * 4 + wave) * p.D + col) = db[c];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * p.D; i += 256) {
        const int which = i / p.D, col = i % p.D;
        const float s = red[(which * 4 + 0) * p.D + col] + red[(which * 4 + 1) * p.D + col] +
                        red[(which * 4 + 2) * p.D + col] + red[(which * 4 + 3) * p.D + col];
        if (p.partial) p.partial[((long)blockIdx.x * 2 + which) * p.D + col] = s;      // every block stores (zeros if it had no row)
        else atomic_add_f32((which ? p.dbeta : p.dgamma) + col, s);
    }
}

// dgamma / dbeta += column sums of the [nblk][2][D] partials of up to 64 LayerNorms.  One workgroup per (LayerNorm, 64
// columns of the 2*D): thread (c, rg) sums rows rg, rg+4, .. with 13 independent loads in flight, LDS folds the four groups.
struct LnReduceArgs { const float* part[64]; float* dg[64]; float* db[64]; int rows[64]; int D; };
__global__ __launch_bounds__(256) void ln_grad_reduce_kernel(const LnReduceArgs a) {
    __shared__ float red[4][64];
    const int ln = blockIdx.y, c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;                   // 0 .. 2D-1: [gamma | beta]
    const float* src = a.part[ln];
    const int nblk = a.rows[ln];
    float s = 0.f;
    if (col < 2 * a.D) {
        const int which = col / a.D, cc = col % a.D;
        // eight independent (clamped, then masked) loads per trip: the reduction is latency-bound, not bandwidth-bound
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int b0 = rg; b0 < nblk; b0 += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + 4 * u;
                const float v = src[((long)min(b, nblk - 1) * 2 + which) * a.D + cc];
                acc[u] += (b < nblk) ? v : 0.f;
            }
        }
        s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
    red[rg][c] = s;
    __syncthreads();
    if (rg == 0 && col < 2 * a.D) {
        const float t = red[0][c] + red[1][c] + red[2][c] + red[3][c];
        float* dst = (col < a.D) ? a.dg[ln] + col : a.db[ln] + (col - a.D);
        *dst += t;
    }
}

}  // namespace

int s3d_launch_ln_fwd(const LnArgs& a, hipStream_t s) {
    S3D_REQUIRE(a.D % 4 == 0 && a.D <= 1024 && a.ldx % 4 == 0, "layernorm: D=%d must be a multiple of 4 and <= 1024", a.D);
    if (a.rows <= 0) return 0;
    const dim3 grid((unsigned)((a.rows + 3) / 4));
    if (a.D <= 256) hipLaunchKernelGGL(ln_fwd_kernel<1>, grid, dim3(256), 0, s, a);
    else if (a.D <= 512) hipLaunchKernelGGL(ln_fwd_kernel<2>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(ln_fwd_kernel<4>, grid, dim3(256), 0, s, a);
    S3D_CHECK_LAUNCH_V("ln_fwd", a.D <= 256 ? 1 : a.D <= 512 ? 2 : 4);
    return 0;
}

int s3d_launch_ln_bwd(const LnBwdArgs& a, hipStream_t s, AdamFillQueue* fillq) {
    S3D_REQUIRE(a.D % 4 == 0 && a.D <= 1024, "layernorm bwd: D=%d must be a multiple of 4 and <= 1024", a.D);
    S3D_REQUIRE(a.dy_parts >= 0 && a.dy_parts <= 4 && (a.dy_parts <= 1 || (a.dy_part_stride & 3) == 0), "layernorm bwd: dy_parts=%d (<= 4), dy_part_stride=%ld (multiple of 4)",
                a.dy_parts, a.dy_part_stride);
    if (a.rows <= 0) return 0;
    long blocks = (a.rows + 4 * MAX_RPW - 1) / (4 * MAX_RPW);
    if (a.rows <= 512) blocks = (a.rows + 3) / 4;         // few rows (final norm on the cls rows): one row per wave
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    if (a.partial) {
        S3D_REQUIRE(a.partial_blocks > 0, "layernorm bwd: partial mode needs partial_blocks > 0");
        blocks = a.partial_blocks;                       // the caller sized the buffer; the kernel grid-strides over rows
    } else if (s3d_deterministic() && (a.dgamma || a.dbeta)) {
        blocks = 1;                                      // single writer of dgamma / dbeta (the atomic path's order is free otherwise)
    }
    // rows each wave handles per trip: as many as the grid leaves it (clamped duplicate rows would only add loads)
    const long per_wave = (a.rows + blocks * 4 - 1) / (blocks * 4);
    const size_t lds = 2 * 4 * a.D * sizeof(float);
    AdamFill fill = adam_fill_none();
    if (fillq) fill = fillq->take(256);
    const unsigned grid = (unsigned)blocks + (unsigned)fill.blocks;
#define S3D_LN_BWD(RPW)                                                                                          \
    do {                                                                                                         \
        if (a.D <= 256) hipLaunchKernelGGL((ln_bwd_kernel<RPW, 1>), dim3(grid), dim3(256), lds, s, a, fill);      \
        else if (a.D <= 512) hipLaunchKernelGGL((ln_bwd_kernel<RPW, 2>), dim3(grid), dim3(256), lds, s, a, fill); \
        else hipLaunchKernelGGL((ln_bwd_kernel<RPW, 4>), dim3(grid), dim3(256), lds, s, a, fill);                 \
    } while (0)
    static const int forced_rpw = s3d_tune_int("S3D_LN_RPW") > 0 ? s3d_tune_int("S3D_LN_RPW") : 0;      // tuning builds only
    const int rpw = forced_rpw ? forced_rpw : (per_wave >= 3 ? 4 : per_wave == 2 ? 2 : 1);
    if (rpw >= 4) S3D_LN_BWD(4);
    else if (rpw == 2) S3D_LN_BWD(2);
    else S3D_LN_BWD(1);
#undef S3D_LN_BWD
    S3D_CHECK_LAUNCH_V("ln_bwd", (rpw >= 4 ? 4 : rpw) * 100 + (a.D <= 256 ? 1 : a.D <= 512 ? 2 : 4) * 10 + (a.partial ? 1 : 0) + (a.dy_parts > 1 ? 1000 : 0));
    return 0;
}

int s3d_launch_ln_grad_reduce(const float* const* partial, float* const* dgamma, float* const* dbeta, int n_ln, int nblk, int D,
                              hipStream_t s) {
    S3D_REQUIRE(n_ln >= 0 && n_ln <= 64 && nblk > 0 && D > 0, "layernorm_grad_reduce: n_ln=%d (<= 64), nblk=%d, D=%d", n_ln, nblk, D);
    if (n_ln == 0) return 0;
    int rows[64];
    for (int i = 0; i < n_ln; ++i) rows[i] = nblk;
    return s3d_launch_ln_grad_reduce_rows(partial, dgamma, dbeta, rows, n_ln, D, s);
}

// ... with a row count per LayerNorm (the dgrad-epilogue LayerNorm backward writes one partial row per 64-row tile)
int s3d_launch_ln_grad_reduce_rows(const float* const* partial, float* const* dgamma, float* const* dbeta, const int* rows, int n_ln, int D,
                                   hipStream_t s) {
    S3D_REQUIRE(n_ln >= 0 && n_ln <= 64 && D > 0, "layernorm_grad_reduce: n_ln=%d (<= 64), D=%d", n_ln, D);
    if (n_ln == 0) return 0;
    LnReduceArgs a;
    for (int i = 0; i < n_ln; ++i) {
        S3D_REQUIRE(rows[i] > 0, "layernorm_grad_reduce: LayerNorm %d has no partial rows", i);
        a.part[i] = partial[i]; a.dg[i] = dgamma[i]; a.db[i] = dbeta[i]; a.rows[i] = rows[i];
    }
    a.D = D;
    hipLaunchKernelGGL(ln_grad_reduce_kernel, dim3((2 * D + 63) / 64, n_ln), dim3(256), 0, s, a);
    S3D_CHECK_LAUNCH("ln_grad_reduce");
    return 0;
}
