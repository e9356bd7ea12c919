// Row-stream GEMM: C[M][N] = A[M][K] @ W[N][K]^T * alpha + bias for MILLIONS of rows and a weight matrix that fits in LDS
// (K <= 96, N <= 192): the 1x1 convolutions and per-point projections of the point path's set abstraction
// (/root/reference/data/pointnet_util.py:238-241, models/3DViT/model.py:52-64 -- 2.1 M grouped rows x 96 channels at cfg-4).
//
// On the 128 x 128 LDS-DMA tiles of gemm.hip these launches are three k-tiles of prologue + epilogue per workgroup, two workgroups per CU:
// 2.5 TB/s on a product that is a pure HBM stream (read the hi + lo planes of A, write fp32 C; 1.6 GB at cfg-4's conv).  Here the weight
// planes are staged once per workgroup (40 KB: four workgroups = sixteen waves per CU), every wave walks 16-row chunks on its own -- the
// rows come straight from global memory as MFMA operand fragments (a row-major bf16 row IS the k-contiguous fragment layout: 16 bytes per
// lane), no LDS round trip, no barrier in the loop -- and C leaves as 16-byte stores (the product is taken transposed, C^T = W A^T, so a lane
// holds four consecutive columns of one row).  Split precision like every forward GEMM: hi*lo + lo*hi + hi*hi on bf16 MFMA, fp32 accumulate.
// Optional S3dGemmArgs::col_sums (the following train-mode BatchNorm's statistics): per-lane fp32 partials over the wave's rows, folded
// over the 16 row lanes and the four waves, one fp64 atomic per column and statistic per workgroup.
#include "gemm.h"
#include "kernels.h"

#include <string.h>

namespace {

struct RsArgs {
    const bf16_t* A_hi; const bf16_t* A_lo; const bf16_t* W_hi; const bf16_t* W_lo; const float* bias; float* C; double* col_sums;
    long lda, ldb, ldc, M;
    int N, K, chunks_per_wg;
    float alpha;
};

constexpr int RS_NH = 96;                                               // output columns per pass (at most six 16-column fragments)

// One 16-row chunk: acc[f] = W_f . A^T (split product), then C rows and the column-sum partials.  MASKED: the ragged last chunk of the
// problem (rows clamped for the loads, stores under a row mask); the main loop is branch-free so that the compiler can count the vector
// memory operations in flight (s_waitcnt vmcnt(NF): the stores of the previous chunk are never waited for).
template <int KS, int NF, bool SUMS, bool MASKED>
__device__ __forceinline__ void rs_chunk(const RsArgs& p, const bf16_t* wh, const bf16_t* wl, const float* sb, int wbase, const u32x4 (&xh)[KS],
                                         const u32x4 (&xl)[KS], long row, bool row_ok, int nb, f32x4 (&cs)[SUMS ? NF : 1], f32x4 (&cq)[SUMS ? NF : 1]) {
    constexpr int KP = KS * 32 + 8;
    const int kg = (threadIdx.x & 63) >> 4;
    f32x4 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int off = wbase + f * 16 * KP + s * 32;
            const bf16x8 fh = *reinterpret_cast<const bf16x8*>(wh + off);
            const bf16x8 fl = *reinterpret_cast<const bf16x8*>(wl + off);
            const bf16x8 ah = __builtin_bit_cast(bf16x8, xh[s]), al = __builtin_bit_cast(bf16x8, xl[s]);
            acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl, ah, acc[f], 0, 0, 0);
            acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, al, acc[f], 0, 0, 0);
            acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, ah, acc[f], 0, 0, 0);
        }
    }
    float* crow = p.C + row * p.ldc + nb + kg * 4;                      // lane = row (lane & 15), columns f * 16 + 4 kg .. + 3
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(sb + f * 16 + kg * 4);
        const f32x4 v = acc[f] * p.alpha + b;
        if constexpr (MASKED) {
            if (row_ok) {
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(crow + f * 16));
                if constexpr (SUMS) { cs[f] += v; cq[f] += v * v; }
            }
        } else {
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(crow + f * 16));
            if constexpr (SUMS) { cs[f] += v; cq[f] += v * v; }
        }
    }
}

template <int KS, int NF, bool SUMS>
__global__ __launch_bounds__(256, SUMS ? 2 : 3) void rowstream_gemm_kernel(const RsArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KP = KS * 32 + 8;                                     // staged weight row in bf16 elements: + 16 bytes, conflict-free 16-byte reads
    constexpr int CPR = KS * 4, NR = NF * 16;                           // 16-byte chunks per weight row; weight rows = output columns per pass
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    bf16_t* wh = reinterpret_cast<bf16_t*>(smem);
    bf16_t* wl = wh + NR * KP;
    float* sb = reinterpret_cast<float*>(wl + NR * KP);                 // bias of the pass [NR]
    float* red = reinterpret_cast<float*>(smem);                        // column-sum fold [4][2][NR], over the weight planes once they are done
    const long chunk0 = (long)blockIdx.x * p.chunks_per_wg;
    const long full = p.M >> 4;                                         // whole 16-row chunks of the problem
    // this wave's whole chunks: chunk0 + wave, + 4, ...
    const long mine_end = min(chunk0 + p.chunks_per_wg, full);
    int koff[KS];                                                       // k >= K: a re-read of the row's last 8 columns against ZERO weight columns
#pragma unroll
    for (int s = 0; s < KS; ++s) koff[s] = min(s * 32 + kg * 8, p.K - 8);
    const int passes = p.N / NR;
    for (int nh = 0; nh < passes; ++nh) {
        const int nb = nh * NR;
        if (nh > 0) __syncthreads();
        for (int i = tid; i < NR * CPR; i += 256) {
            const int n = i / CPR, c = i % CPR;
            u32x4 h = {0u, 0u, 0u, 0u}, l = {0u, 0u, 0u, 0u};
            if (c * 8 < p.K) {
                h = *reinterpret_cast<const u32x4*>(p.W_hi + (long)(nb + n) * p.ldb + c * 8);
                l = *reinterpret_cast<const u32x4*>(p.W_lo + (long)(nb + n) * p.ldb + c * 8);
            }
            *reinterpret_cast<u32x4*>(wh + n * KP + c * 8) = h;
            *reinterpret_cast<u32x4*>(wl + n * KP + c * 8) = l;
        }
        if (tid < NR) sb[tid] = p.bias != nullptr ? p.bias[nb + tid] : 0.f;
        __syncthreads();
        f32x4 cs[SUMS ? NF : 1], cq[SUMS ? NF : 1];
        if constexpr (SUMS) {
#pragma unroll
            for (int f = 0; f < NF; ++f) { cs[f] = f32x4{0.f, 0.f, 0.f, 0.f}; cq[f] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
        // the weight fragments are re-read from LDS for every chunk (an opaque base keeps the compiler from hoisting 36 loop-invariant
        // fragments into 144 registers: this kernel lives on waves -- bytes in flight -- per CU)
        int wbase = r16 * KP + kg * 8;
        asm volatile("" : "+v"(wbase));
        u32x4 nxh[KS], nxl[KS];
        long c = chunk0 + wave;
        if (c < mine_end) {
            // the rows of chunk c + 4 are requested before chunk c is multiplied and stored: two chunks (12 KB at K = 96) in flight per wave
            const bf16_t* ah = p.A_hi + (c * 16 + r16) * p.lda;
            const bf16_t* al = p.A_lo + (c * 16 + r16) * p.lda;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                nxh[s] = *reinterpret_cast<const u32x4*>(ah + koff[s]);
                nxl[s] = *reinterpret_cast<const u32x4*>(al + koff[s]);
            }
#pragma unroll 1
            for (; c < mine_end; c += 4) {
                u32x4 xh[KS], xl[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) { xh[s] = nxh[s]; xl[s] = nxl[s]; }
                const long cn = min(c + 4, mine_end - 1);               // (past the end: a harmless re-read of the wave group's last chunk)
                ah = p.A_hi + (cn * 16 + r16) * p.lda;
                al = p.A_lo + (cn * 16 + r16) * p.lda;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    nxh[s] = *reinterpret_cast<const u32x4*>(ah + koff[s]);
                    nxl[s] = *reinterpret_cast<const u32x4*>(al + koff[s]);
                }
                asm volatile("" : "+v"(wbase));
                rs_chunk<KS, NF, SUMS, false>(p, wh, wl, sb, wbase, xh, xl, c * 16 + r16, true, nb, cs, cq);
            }
        }
        // the ragged last chunk of the problem: the wave whose turn it would be
        if ((p.M & 15) != 0 && full >= chunk0 && full < chunk0 + p.chunks_per_wg && (int)((full - chunk0) & 3) == wave) {
            const long row = full * 16 + r16, rc = min(row, p.M - 1);
            u32x4 xh[KS], xl[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                xh[s] = *reinterpret_cast<const u32x4*>(p.A_hi + rc * p.lda + koff[s]);
                xl[s] = *reinterpret_cast<const u32x4*>(p.A_lo + rc * p.lda + koff[s]);
            }
            rs_chunk<KS, NF, SUMS, true>(p, wh, wl, sb, wbase, xh, xl, rc, row < p.M, nb, cs, cq);
        }
        if constexpr (SUMS) {
            if (p.col_sums != nullptr) {                                // uniform
                __syncthreads();                                        // every wave is done with the weight planes
#pragma unroll
                for (int f = 0; f < NF; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float a = cs[f][r], q = cq[f][r];
#pragma unroll
                        for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m, 64); q += __shfl_xor(q, m, 64); }
                        if (r16 == 0) {
                            red[(wave * 2 + 0) * NR + f * 16 + kg * 4 + r] = a;
                            red[(wave * 2 + 1) * NR + f * 16 + kg * 4 + r] = q;
                        }
                    }
                __syncthreads();
                if (tid < 2 * NR) {
                    const int which = tid / NR, cc = tid % NR;
                    const float t = red[(0 * 2 + which) * NR + cc] + red[(1 * 2 + which) * NR + cc] + red[(2 * 2 + which) * NR + cc] +
                                    red[(3 * 2 + which) * NR + cc];
                    unsafeAtomicAdd(p.col_sums + (long)which * p.N + nb + cc, (double)t);
                }
            }
        }
    }
}

template <int KS, int NF, bool SUMS>
void rs_launch(const RsArgs& r, unsigned grid, hipStream_t s) {
    constexpr int LDS = 2 * NF * 16 * (KS * 32 + 8) * 2 + NF * 16 * 4;
    hipLaunchKernelGGL((rowstream_gemm_kernel<KS, NF, SUMS>), dim3(grid), dim3(256), LDS, s, r);
}
template <int KS, int NF>
void rs_launch_s(const RsArgs& r, unsigned grid, bool sums, hipStream_t s) {
    if (sums) rs_launch<KS, NF, true>(r, grid, s);
    else rs_launch<KS, NF, false>(r, grid, s);
}
template <int KS>
void rs_launch_n(const RsArgs& r, unsigned grid, int nf, bool sums, hipStream_t s) {
    if (nf == 6) rs_launch_s<KS, 6>(r, grid, sums, s);
    else if (nf == 4) rs_launch_s<KS, 4>(r, grid, sums, s);
    else rs_launch_s<KS, 3>(r, grid, sums, s);
}
// 16-column fragments per pass: N = 48 -> 3, 64 / 128 -> 4, 96 / 192 -> 6; 0 = not a shape of this kernel
int rs_frags(int N) { return (N == 96 || N == 192) ? 6 : (N == 64 || N == 128) ? 4 : N == 48 ? 3 : 0; }

}  // namespace

// Shapes the row-stream kernel takes from s3d_launch_gemm (forward NT, split precision, F32 epilogue): a weight matrix that fits in LDS and
// enough rows to fill the chip several times over
bool s3d_rowstream_gemm_ok(bool split, int epi, const GemmArgs& a) {
    return split && epi == EPI_F32 && a.M >= 32768 && a.K >= 8 && a.K <= 96 && (a.K & 7) == 0 && rs_frags(a.N) != 0 && a.C != nullptr && a.A_lo != nullptr &&
           a.B_lo != nullptr && (a.ldc & 3) == 0 && (a.lda & 7) == 0 && (a.ldb & 7) == 0 && a.drop_thr == 0 && a.ln_tickets == nullptr;
}

int s3d_launch_rowstream_gemm(const GemmArgs& a, hipStream_t s) {
    S3D_REQUIRE(s3d_rowstream_gemm_ok(true, EPI_F32, a), "rowstream gemm: M=%d N=%d K=%d: K <= 96, N in {48, 64, 96, 128, 192}, split planes, fp32 output", a.M, a.N,
                a.K);
    RsArgs r;
    memset(&r, 0, sizeof(r));
    r.A_hi = a.A_hi; r.A_lo = a.A_lo; r.W_hi = a.B_hi; r.W_lo = a.B_lo; r.bias = a.bias; r.C = a.C; r.col_sums = a.col_sums;
    r.lda = a.lda; r.ldb = a.ldb; r.ldc = a.ldc; r.M = a.M; r.N = a.N; r.K = a.K; r.alpha = a.alpha;
    const long chunks = ((long)a.M + 15) / 16;
    long per = (chunks + 2047) / 2048;                                  // about two rounds of the 1024 resident workgroups
    per = (per + 3) / 4 * 4;
    if (per < 4) per = 4;
    r.chunks_per_wg = (int)per;
    const unsigned grid = (unsigned)((chunks + per - 1) / per);
    const int ks = (a.K + 31) / 32, nf = rs_frags(a.N);
    const bool sums = a.col_sums != nullptr;
    if (ks == 1) rs_launch_n<1>(r, grid, nf, sums, s);
    else if (ks == 2) rs_launch_n<2>(r, grid, nf, sums, s);
    else rs_launch_n<3>(r, grid, nf, sums, s);
    S3D_CHECK_LAUNCH_V("gemm_rowstream", ks * 100 + nf * 10 + (sums ? 1 : 0));
    return 0;
}
