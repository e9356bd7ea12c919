// Point-cloud kernels of the PointTransformerCls/Seg path for gfx950: farthest-point sampling, k-nearest-neighbour
// selection, neighbourhood gather / scatter-add, train-mode BatchNorm (+ReLU, + max over the k neighbours) forward and
// backward, 3-NN inverse-distance interpolation, mean pooling.  These are HBM/latency-bound integer + fp32 kernels (no
// MFMA); the 1x1 convolutions / Linear layers between them run on the MFMA GEMM (gemm.hip).
//
// Distances are computed as ((dx*dx + dy*dy) + dz*dz) with explicitly NON-fused fp32 operations so that neighbour order
// is bit-identical to the reference's torch.sum((src - dst) ** 2, -1) (data/pointnet_util.py:22-36) -- an FMA contraction
// would flip near-ties at the 16th/17th neighbour.
#include "kernels.h"

#include <stdint.h>
#include <stdlib.h>

namespace {

__device__ __forceinline__ float sqdist(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// ------------------------------------------------------------------------------------------- FPS
// One workgroup (4 waves) per cloud.  npoint strictly sequential iterations of {update the running min distance to the newest
// centroid, argmax}; ties resolve to the smallest index (torch.max's first maximum).  An iteration is pure latency, so the
// kernel is built around its critical path: coordinates and running distances live in registers (PPT points per thread), the
// wave maximum is six DPP max steps (row_shr 1/2/4/8, row_bcast 15/31 -- max is idempotent, so no row / bank masks are needed)
// read back from lane 63, the winning lane is found by ballot (exact ties, rare, fall back to a scalar loop), and the four
// wave results meet in a double-buffered LDS slot with ONE barrier per iteration: every thread then reduces the four slots
// itself and fetches the new centroid with a broadcast LDS read.  (First version: cloud re-read from LDS every iteration, 12
// ds_bpermute shuffles and three barriers per iteration -- 1.28 us per iteration.)
__device__ __forceinline__ float wave_max_nonneg(float v) {
    int x = __float_as_int(v);
#define S3D_DPP_MAX(ctrl)                                                                  \
    {                                                                                      \
        const int t = __builtin_amdgcn_update_dpp(x, x, ctrl, 0xf, 0xf, false);            \
        x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(t)));                   \
    }
    S3D_DPP_MAX(0x111) S3D_DPP_MAX(0x112) S3D_DPP_MAX(0x114) S3D_DPP_MAX(0x118)           // row_shr:1,2,4,8 -> lane 15 of each row
    S3D_DPP_MAX(0x142) S3D_DPP_MAX(0x143)                                                 // row_bcast:15, row_bcast:31 -> lane 63
#undef S3D_DPP_MAX
    return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}
__device__ __forceinline__ int wave_min_i32(int x) {
#define S3D_DPP_MIN(ctrl) x = min(x, __builtin_amdgcn_update_dpp(x, x, ctrl, 0xf, 0xf, false));
    S3D_DPP_MIN(0x111) S3D_DPP_MIN(0x112) S3D_DPP_MIN(0x114) S3D_DPP_MIN(0x118) S3D_DPP_MIN(0x142) S3D_DPP_MIN(0x143)
#undef S3D_DPP_MIN
    return __builtin_amdgcn_readlane(x, 63);
}

constexpr int FPS_MAX_PPT = 8;   // points per thread -> N <= 2048

template <int PPT>
__global__ __launch_bounds__(256) void fps_kernel(const float* __restrict__ xyz, long xyz_ld, const long long* __restrict__ start,
                                                  int N, int npoint, int* __restrict__ out_idx, float* __restrict__ new_xyz) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4* sx = reinterpret_cast<f32x4*>(smem);                         // [N] (x, y, z, -)
    f32x4* slot = sx + N;                                               // [2][2]: per buffer (v0 i0 v1 i1) (v2 i2 v3 i3)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* src = xyz + (long)b * N * xyz_ld;
    float px[PPT], py[PPT], pz[PPT], dist[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int i = tid + j * 256;
        const bool in = i < N;
        const float* q = src + (long)(in ? i : 0) * xyz_ld;
        px[j] = q[0]; py[j] = q[1]; pz[j] = q[2];
        dist[j] = in ? 1e10f : -1.f;                                    // padding lanes can never win (distances are >= 0)
        if (in) sx[i] = f32x4{px[j], py[j], pz[j], 0.f};
    }
    int far = (int)start[b];
    __syncthreads();
    for (int it = 0; it < npoint; ++it) {
        const f32x4 c = sx[far];
        if (tid == 0) {
            out_idx[(long)b * npoint + it] = far;
            float* o = new_xyz + ((long)b * npoint + it) * 3;
            o[0] = c[0]; o[1] = c[1]; o[2] = c[2];
        }
        float best = -1.f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const float d = sqdist(px[j], py[j], pz[j], c[0], c[1], c[2]);
            dist[j] = fminf(dist[j], d);                                // -1 stays -1 for padding
            if (dist[j] > best) { best = dist[j]; bi = tid + j * 256; } // strict >: first (smallest) index wins
        }
        const float wmax = wave_max_nonneg(best);
        const unsigned long long tied = __ballot(best == wmax);
        int wbi;
        if ((tied & (tied - 1)) == 0) wbi = __builtin_amdgcn_readlane(bi, __ffsll((long long)tied) - 1);
        else wbi = wave_min_i32(best == wmax ? bi : 0x7fffffff);       // exact ties across lanes (duplicates, padding): smallest index
        float* sl = reinterpret_cast<float*>(slot + 2 * (it & 1));
        if (lane == 0) { sl[2 * wave] = wmax; sl[2 * wave + 1] = __int_as_float(wbi); }
        __syncthreads();
        const f32x4 a = slot[2 * (it & 1)], e = slot[2 * (it & 1) + 1];
        float v = a[0]; int ix = __float_as_int(a[1]);
        { const float ov = a[2]; const int oi = __float_as_int(a[3]); if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; } }
        { const float ov = e[0]; const int oi = __float_as_int(e[1]); if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; } }
        { const float ov = e[2]; const int oi = __float_as_int(e[3]); if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; } }
        far = ix;
    }
}

// ------------------------------------------------------------------------------------------- kNN
// One wave per query point: distances to all N reference points in registers (PPL <= 32 per lane, templated so that small
// clouds do not scan dead registers), then K rounds of {lane-local min over not-yet-taken, wave argmin}.  The wave argmin
// is a DPP min over the distance bits (distances are >= 0, so their bit patterns order like the floats) + a ballot for the
// owning lane; exact ties fall back to a DPP min over the indices.  Ascending distance; ties -> smaller index.  K == 3 also
// emits the normalised inverse-distance weights of PointNetFeaturePropagation (data/pointnet_util.py:401-408).
constexpr int KNN_MAX_PPL = 32;  // N <= 2048

template <int K, int KNN_PPL>
__global__ __launch_bounds__(256) void knn_kernel(const float* __restrict__ query, const float* __restrict__ ref, int S, int N,
                                                  long total, int* __restrict__ out_idx, float* __restrict__ out_w) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= total) return;
    const long b = item / S;
    const float qx = query[item * 3], qy = query[item * 3 + 1], qz = query[item * 3 + 2];
    const float* r = ref + b * N * 3;
    float d[KNN_PPL];
#pragma unroll
    for (int j = 0; j < KNN_PPL; ++j) {
        const int pidx = lane + j * 64;
        d[j] = INFINITY;
        if (pidx < N) d[j] = sqdist(qx, qy, qz, r[3 * pidx], r[3 * pidx + 1], r[3 * pidx + 2]);
    }
    unsigned taken = 0u;
    float wsum = 0.f, wv[K];
    int wi[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float best = INFINITY;
        int bj = -1;
#pragma unroll
        for (int j = 0; j < KNN_PPL; ++j)
            if (!((taken >> j) & 1u) && d[j] < best) { best = d[j]; bj = j; }
        int bi = (bj < 0) ? 0x7fffffff : lane + bj * 64;
        const float v = __int_as_float(wave_min_i32(__float_as_int(best)));
        const unsigned long long tied = __ballot(best == v);
        int ix;
        if ((tied & (tied - 1)) == 0) ix = __builtin_amdgcn_readlane(bi, __ffsll((long long)tied) - 1);
        else ix = wave_min_i32(best == v ? bi : 0x7fffffff);
        if (ix == bi && bj >= 0) taken |= (1u << bj);
        wi[k] = ix;
        wv[k] = 1.0f / (v + 1e-8f);
        wsum += wv[k];
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            out_idx[item * K + k] = wi[k];
            if (out_w) out_w[item * K + k] = wv[k] / wsum;
        }
    }
}

// ------------------------------------------------------------------------------------------- neighbourhood gather
// rows r = (b, s, j): A[r] = [xyz[b, idx] - new_xyz[b, s] (3) | feats[b, idx] (C) | 0 pad] as split-bf16 planes
// (sample_and_group(knn=True), data/pointnet_util.py:126-134)
__global__ void group_gather_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                    const float* __restrict__ feats, const int* __restrict__ idx, int N, int S, int K, int C,
                                    long rows, bf16_t* __restrict__ a_hi, bf16_t* __restrict__ a_lo, int lda) {
    const long total = rows * lda;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / lda;
        const int c = (int)(i % lda);
        const long bs = r / K;
        const long b = bs / S;
        const int p = idx[r];
        float v = 0.f;
        if (c < 3) v = xyz[(b * N + p) * 3 + c] - new_xyz[bs * 3 + c];
        else if (c < 3 + C) v = feats[(b * N + p) * (long)C + (c - 3)];
        bf16_t h, l;
        split_bf16(v, h, l);
        a_hi[i] = h;
        a_lo[i] = l;
    }
}
// backward: dfeats[b, idx[r], c] += dA[r][3 + c]
__global__ void group_scatter_kernel(const float* __restrict__ dA, int ldd, const int* __restrict__ idx, int N, int S, int K,
                                     int C, long rows, float* __restrict__ dfeats) {
    const long total = rows * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        const int c = (int)(i % C);
        const long b = r / ((long)S * K);
        atomic_add_f32(dfeats + (b * N + idx[r]) * (long)C + c, dA[r * ldd + 3 + c]);
    }
}

// ------------------------------------------------------------------------------------------- BatchNorm (train mode)
// statistics over rows per channel in double (fp64 atomics): sums[c] = sum x, sums[C + c] = sum x^2
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, long rows, int C, int ld, double* __restrict__ sums) {
    // thread t handles channel t % C of rows (t / C) + i * (256 / C * gridDim.x); C <= 256 and 256 % C need not hold
    const int per = 256 / C;                       // rows handled per block iteration
    const int c = threadIdx.x % C, sub = threadIdx.x / C;
    if (sub >= per) return;
    double s = 0.0, q = 0.0;
    for (long r = (long)blockIdx.x * per + sub; r < rows; r += (long)gridDim.x * per) {
        const double v = x[r * ld + c];
        s += v; q += v * v;
    }
    atomicAdd(sums + c, s);
    atomicAdd(sums + C + c, q);
}
// mean / rstd + running statistics (momentum m, unbiased running variance), nn.BatchNorm semantics
__global__ void bn_finalize_kernel(const double* __restrict__ sums, long rows, int C, float eps, float momentum,
                                   float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ run_mean,
                                   float* __restrict__ run_var, const float* __restrict__ momentum_dev) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (momentum_dev) momentum = *momentum_dev;        // device-resident hyper-parameter: HIP-graph replays see updates
    const double m = sums[c] / rows;
    double var = sums[C + c] / rows - m * m;
    if (var < 0) var = 0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)m;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)(var * ((double)rows / (double)(rows > 1 ? rows - 1 : 1)));
    }
}
__global__ void bn_eval_prepare_kernel(const float* __restrict__ run_mean, const float* __restrict__ run_var, int C, float eps,
                                       float* __restrict__ mean, float* __restrict__ rstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) { mean[c] = run_mean[c]; rstd[c] = 1.0f / sqrtf(run_var[c] + eps); }
}
// y = relu((x - mean) * rstd * gamma + beta) -> fp32 and/or split planes (row pitch ldo)
__global__ void bn_relu_kernel(const float* __restrict__ x, long rows, int C, int ld, const float* __restrict__ mean,
                               const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                               float* __restrict__ y, bf16_t* __restrict__ y_hi, bf16_t* __restrict__ y_lo, int ldo) {
    const long total = rows * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        const int c = (int)(i % C);
        const float v = fmaxf((x[r * ld + c] - mean[c]) * rstd[c] * gamma[c] + beta[c], 0.f);
        if (y) y[r * ldo + c] = v;
        if (y_hi) {
            bf16_t h, l;
            split_bf16(v, h, l);
            y_hi[r * ldo + c] = h;
            if (y_lo) y_lo[r * ldo + c] = l;
        }
    }
}
// fused BN + ReLU + max over the K neighbours: out[s][c] = max_k relu(bn(x[(s,k)][c])), arg[s][c] = first argmax
__global__ void bn_relu_max_kernel(const float* __restrict__ x, long groups, int K, int C, const float* __restrict__ mean,
                                   const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ out, unsigned char* __restrict__ arg) {
    const long total = groups * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long s = i / C;
        const int c = (int)(i % C);
        const float a = rstd[c] * gamma[c], b = beta[c] - mean[c] * a;
        float best = -INFINITY;
        int bk = 0;
        for (int k = 0; k < K; ++k) {
            const float v = fmaxf(x[(s * K + k) * C + c] * a + b, 0.f);
            if (v > best) { best = v; bk = k; }
        }
        out[i] = best;
        arg[i] = (unsigned char)bk;
    }
}
// backward statistics.  mode 0: g = dy * (y > 0) per row.  mode 1 (max): g[(s,k)] = (k == arg[s]) ? dy[s] * (y > 0) : 0.
// sums[c] = sum g, sums[C + c] = sum g * xhat
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const float* __restrict__ x, int ld, const float* __restrict__ dy, int lddy,
                                                           const unsigned char* __restrict__ arg, int K, long rows, int C,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           double* __restrict__ sums) {
    const int per = 256 / C;
    const int c = threadIdx.x % C, sub = threadIdx.x / C;
    if (sub >= per) return;
    const float a = rstd[c] * gamma[c], b = beta[c] - mean[c] * a;
    double s = 0.0, q = 0.0;
    const long n = arg ? rows / K : rows;          // mode 1 iterates groups
    for (long r = (long)blockIdx.x * per + sub; r < n; r += (long)gridDim.x * per) {
        const long xr = arg ? r * K + arg[r * C + c] : r;
        const float xv = x[xr * ld + c];
        const float g = (xv * a + b > 0.f) ? dy[r * lddy + c] : 0.f;
        s += g;
        q += (double)g * (double)((xv - mean[c]) * rstd[c]);
    }
    atomicAdd(sums + c, s);
    atomicAdd(sums + C + c, q);
}
// dx = gamma * rstd * (g - sum_g / R - xhat * sum_gxhat / R)  -> bf16 (operand of the following wgrad / dgrad GEMMs);
// dgamma += sum_gxhat, dbeta += sum_g (block 0 only)
__global__ void bn_bwd_apply_kernel(const float* __restrict__ x, int ld, const float* __restrict__ dy, int lddy,
                                    const unsigned char* __restrict__ arg, int K, long rows, int C,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const double* __restrict__ sums, bf16_t* __restrict__ dx, int lddx,
                                    float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const long total = rows * C;
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            atomic_add_f32(dgamma + c, (float)sums[C + c]);
            atomic_add_f32(dbeta + c, (float)sums[c]);
        }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        const int c = (int)(i % C);
        const float a = rstd[c] * gamma[c], b = beta[c] - mean[c] * a;
        const float xv = x[r * ld + c];
        float g = 0.f;
        if (arg) {
            const long s = r / K;
            if ((int)(r % K) == (int)arg[s * C + c] && xv * a + b > 0.f) g = dy[s * lddy + c];
        } else if (xv * a + b > 0.f) {
            g = dy[r * lddy + c];
        }
        const float xh = (xv - mean[c]) * rstd[c];
        const float v = a * (g - (float)(sums[c] / rows) - xh * (float)(sums[C + c] / rows));
        dx[r * lddx + c] = f2bf(v);
    }
}

// ------------------------------------------------------------------------------------------- BatchNorm, vectorised (C % 4 == 0)
// The element-indexed kernels above pay a 64-bit division per element, scalar 4-byte accesses and (backward) two fp64
// divisions per element: ~2.5 TB/s on streams that are pure HBM traffic.  Here a thread owns ONE quad of channels for the
// whole launch (q = tid % (C/4), row lane = tid / (C/4)), so every per-channel constant lives in registers, rows are walked
// with a fixed stride and all accesses are 16-byte (8-byte for bf16 / 4-byte for the arg-max bytes).
struct BnLane {
    int q, sub, rpb;      // channel quad, row lane within the block, rows per block iteration
    bool on;
};
__device__ __forceinline__ BnLane bn_lane(int C) {
    BnLane l;
    const int c4 = C >> 2;
    l.rpb = 256 / c4;
    l.q = threadIdx.x % c4;
    l.sub = threadIdx.x / c4;
    l.on = l.sub < l.rpb;
    return l;
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// block-level fold of the per-thread (sum, sum-of-products) partials of one channel quad, then ONE fp64 atomic per channel
// per workgroup (every row lane adding its own partial put 256/(C/4) x more same-address atomics on the sums than the scalar
// kernel had and made the vector kernels slower than it)
__device__ __forceinline__ void bn_fold_sums(const BnLane& l, int C, const double (&s)[4], const double (&q)[4], double* __restrict__ sums) {
    __shared__ double red[2048];                       // [rpb][2C]: (256 / (C/4)) * 2C = 2048 doubles for any C
    if (l.on) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            red[l.sub * 2 * C + 4 * l.q + i] = s[i];
            red[l.sub * 2 * C + C + 4 * l.q + i] = q[i];
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * C; j += blockDim.x) {
        double t = 0.0;
        for (int r = 0; r < l.rpb; ++r) t += red[r * 2 * C + j];
        atomicAdd(sums + j, t);
    }
}

__global__ __launch_bounds__(256) void bn_stats_vec_kernel(const float* __restrict__ x, long rows, int C, int ld, double* __restrict__ sums) {
    const BnLane l = bn_lane(C);
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (l.on)
        for (long r = (long)blockIdx.x * l.rpb + l.sub; r < rows; r += (long)gridDim.x * l.rpb) {
            const f32x4 v = ld4(x + r * ld + 4 * l.q);
#pragma unroll
            for (int i = 0; i < 4; ++i) { const double d = v[i]; s[i] += d; q[i] += d * d; }
        }
    bn_fold_sums(l, C, s, q, sums);
}

__global__ __launch_bounds__(256) void bn_relu_vec_kernel(const float* __restrict__ x, long rows, int C, int ld, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y,
                                                          bf16_t* __restrict__ y_hi, bf16_t* __restrict__ y_lo, int ldo) {
    const BnLane l = bn_lane(C);
    if (!l.on) return;
    const int c = 4 * l.q;
    const f32x4 m = ld4(mean + c), rs = ld4(rstd + c), g = ld4(gamma + c), be = ld4(beta + c);
    for (long r = (long)blockIdx.x * l.rpb + l.sub; r < rows; r += (long)gridDim.x * l.rpb) {
        const f32x4 v = ld4(x + r * ld + c);
        f32x4 o;
        union { u32x2 u; bf16_t h[4]; } hi, lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[i] = fmaxf((v[i] - m[i]) * rs[i] * g[i] + be[i], 0.f);         // same expression order as the scalar kernel
            split_bf16(o[i], hi.h[i], lo.h[i]);
        }
        if (y) *reinterpret_cast<f32x4*>(y + r * ldo + c) = o;
        if (y_hi) {
            *reinterpret_cast<u32x2*>(y_hi + r * ldo + c) = hi.u;
            if (y_lo) *reinterpret_cast<u32x2*>(y_lo + r * ldo + c) = lo.u;
        }
    }
}

__global__ __launch_bounds__(256) void bn_relu_max_vec_kernel(const float* __restrict__ x, long groups, int K, int C,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ out, unsigned char* __restrict__ arg) {
    const BnLane l = bn_lane(C);
    if (!l.on) return;
    const int c = 4 * l.q;
    const f32x4 m = ld4(mean + c), rs = ld4(rstd + c), g = ld4(gamma + c), be = ld4(beta + c);
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = rs[i] * g[i]; b[i] = be[i] - m[i] * a[i]; }
    for (long s = (long)blockIdx.x * l.rpb + l.sub; s < groups; s += (long)gridDim.x * l.rpb) {
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bk[4] = {0, 0, 0, 0};
        for (int k = 0; k < K; ++k) {
            const f32x4 v = ld4(x + (s * K + k) * (long)C + c);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float t = fmaxf(v[i] * a[i] + b[i], 0.f);
                if (t > best[i]) { best[i] = t; bk[i] = k; }
            }
        }
        *reinterpret_cast<f32x4*>(out + s * C + c) = best;
        *reinterpret_cast<unsigned*>(arg + s * C + c) = (unsigned)bk[0] | ((unsigned)bk[1] << 8) | ((unsigned)bk[2] << 16) | ((unsigned)bk[3] << 24);
    }
}

// upstream gradient quad from the fp32 tensor, or (dyb != nullptr) from its bf16 form -- the dgrad GEMM that produces it can write
// bf16 directly: 0.4 GB instead of 0.8 GB written and half the bytes on each of the two reads of this backward (cfg-4 level 0)
__device__ __forceinline__ f32x4 ld_grad4(const float* dy, const bf16_t* dyb, long off) {
    if (dyb) {
        const u32x2 w = *reinterpret_cast<const u32x2*>(dyb + off);
        return f32x4{__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xffff0000u), __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xffff0000u)};
    }
    return ld4(dy + off);
}

template <int U>
__global__ __launch_bounds__(256) void bn_bwd_stats_vec_kernel(const float* __restrict__ x, int ld, const float* __restrict__ dy, int lddy,
                                                               const unsigned char* __restrict__ arg, int K, long rows, int C,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               double* __restrict__ sums, const bf16_t* __restrict__ dyb) {
    const BnLane l = bn_lane(C);
    const int c = 4 * l.q;
    const f32x4 m = ld4(mean + c), rs = ld4(rstd + c), g = ld4(gamma + c), be = ld4(beta + c);
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = rs[i] * g[i]; b[i] = be[i] - m[i] * a[i]; }
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    const long n = arg ? rows / K : rows;          // max mode iterates groups
    const long stride = (long)gridDim.x * l.rpb;
    for (long r0 = (long)blockIdx.x * l.rpb + l.sub; l.on && r0 < n; r0 += U * stride) {
        f32x4 d[U], xv[U];                         // U rows per trip, all loads first: the pass runs on 1024 workgroups (atomics), i.e. 16 waves per CU
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long r = min(r0 + u * stride, n - 1);
            d[u] = ld_grad4(dy, dyb, r * lddy + c);
            if (arg) {
                const unsigned ab = *reinterpret_cast<const unsigned*>(arg + r * C + c);
#pragma unroll
                for (int i = 0; i < 4; ++i) xv[u][i] = x[(r * K + ((ab >> (8 * i)) & 255u)) * ld + c + i];
            } else {
                xv[u] = ld4(x + r * ld + c);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r0 + u * stride >= n) break;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float gg = (xv[u][i] * a[i] + b[i] > 0.f) ? d[u][i] : 0.f;
                s[i] += gg;
                q[i] += (double)gg * (double)((xv[u][i] - m[i]) * rs[i]);
            }
        }
    }
    bn_fold_sums(l, C, s, q, sums);
}

template <int U>
__global__ __launch_bounds__(256) void bn_bwd_apply_vec_kernel(const float* __restrict__ x, int ld, const float* __restrict__ dy, int lddy,
                                                               const unsigned char* __restrict__ arg, int K, long rows, int C,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const double* __restrict__ sums, bf16_t* __restrict__ dx, int lddx,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               const bf16_t* __restrict__ dyb, double inv_rows) {
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            atomic_add_f32(dgamma + c, (float)sums[C + c]);
            atomic_add_f32(dbeta + c, (float)sums[c]);
        }
    const BnLane l = bn_lane(C);
    if (!l.on) return;
    const int c = 4 * l.q;
    const f32x4 m = ld4(mean + c), rs = ld4(rstd + c), g = ld4(gamma + c), be = ld4(beta + c);
    f32x4 a, b, s1, s2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = rs[i] * g[i];
        b[i] = be[i] - m[i] * a[i];
        s1[i] = (float)(sums[c + i] * inv_rows);        // (not "/ rows": eight fp64 divisions per thread in front of a ~13-trip row loop)
        s2[i] = (float)(sums[C + c + i] * inv_rows);
    }
    const unsigned uK = (unsigned)(K > 0 ? K : 1);
    const int kshift = (uK & (uK - 1)) == 0 ? __builtin_ctz(uK) : -1;        // k = 16 neighbours: a shift, not a 64-bit division per row
    // U rows per trip, every load of the trip in flight before the first use: one row per trip leaves a wave with 1.5 KB outstanding,
    // and the pass ran at 2.9 TB/s -- latency x bytes in flight, not HBM
    const long stride = (long)gridDim.x * l.rpb;
    // The exact dx sums to zero over the rows of every channel; its bf16 copy does not -- the rounding residues of 2 M rows are a random walk,
    // and whatever multiplies dx by a column of non-zero mean (a bias, post-ReLU features) picks up mean x (that walk) where the exact
    // product cancels: 4 - 7 % of the small gradients of the classification model's input layers (profiles/r04_point_grad_noise.txt).  So the
    // residue is carried: every element is rounded with the thread's running residue added, and what the threads of a workgroup are left
    // with goes onto the last element of the quad's first row lane.  The column sum of the stored bf16 then misses the exact one by at most
    // half an ulp per WORKGROUP instead of per element.
    f32x4 carry = {0.f, 0.f, 0.f, 0.f};
    u32x2* last_p = nullptr;
    union { u32x2 w; bf16_t h[4]; } last_o;
    last_o.w = u32x2{0u, 0u};
    for (long r0 = (long)blockIdx.x * l.rpb + l.sub; r0 < rows; r0 += U * stride) {
        f32x4 xv[U], d[U];
        unsigned ab[U], kk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long r = min(r0 + u * stride, rows - 1);
            xv[u] = ld4(x + r * ld + c);
            if (arg) {
                const long sg = kshift >= 0 ? (r >> kshift) : r / uK;
                kk[u] = (unsigned)(r - sg * uK);
                ab[u] = *reinterpret_cast<const unsigned*>(arg + sg * C + c);
                d[u] = ld_grad4(dy, dyb, sg * lddy + c);
            } else {
                d[u] = ld_grad4(dy, dyb, r * lddy + c);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long r = r0 + u * stride;
            if (r >= rows) break;
            union { u32x2 w; bf16_t h[4]; } o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool on = xv[u][i] * a[i] + b[i] > 0.f && (!arg || kk[u] == ((ab[u] >> (8 * i)) & 255u));
                const float gq = on ? d[u][i] : 0.f;
                const float xh = (xv[u][i] - m[i]) * rs[i];
                const float v = a[i] * (gq - s1[i] - xh * s2[i]) + carry[i];
                o.h[i] = f2bf(v);
                carry[i] = v - bf2f(o.h[i]);
            }
            last_p = reinterpret_cast<u32x2*>(dx + r * lddx + c);
            *last_p = o.w;
            last_o.w = o.w;
        }
    }
    __shared__ f32x4 left[256];
    left[threadIdx.x] = carry;
    __syncthreads();
    if (l.sub == 0 && last_p != nullptr) {
        const int c4 = C >> 2;
        f32x4 e = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < l.rpb; ++j) e += left[j * c4 + l.q];
#pragma unroll
        for (int i = 0; i < 4; ++i) last_o.h[i] = f2bf(bf2f(last_o.h[i]) + e[i]);
        *last_p = last_o.w;
    }
}

// ------------------------------------------------------------------------------------------- 3-NN interpolation
// out[b, n] = sum_j w[b, n, j] * f1[b, idx[b, n, j]] + f2[b, n]     (TransitionUp, models/3DViT/model.py:67-72)
__global__ void interp3_kernel(const float* __restrict__ f1, int S, const float* __restrict__ f2, const int* __restrict__ idx,
                               const float* __restrict__ w, int N, int C, long rows, float* __restrict__ out) {
    const long total = rows * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        const int c = (int)(i % C);
        const long b = r / N;
        float v = f2[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) v += w[r * 3 + j] * f1[(b * S + idx[r * 3 + j]) * (long)C + c];
        out[i] = v;
    }
}
// channel-quad version (C % 4 == 0, rows * C / 4 < 2^32): 16-byte accesses and one 32-bit division per four channels instead of a
// 64-bit division / modulo per element (the scalar kernels spent most of their instructions on index arithmetic)
__global__ __launch_bounds__(256) void interp3_vec_kernel(const float* __restrict__ f1, int S, const float* __restrict__ f2,
                                                          const int* __restrict__ idx, const float* __restrict__ w, int N, int C,
                                                          unsigned rows, float* __restrict__ out) {
    const unsigned qpr = (unsigned)C / 4, total = rows * qpr;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned r = i / qpr, c = 4 * (i - r * qpr), b = r / (unsigned)N;
        f32x4 v = ld4(f2 + (long)r * C + c);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float wj = w[r * 3 + j];
            const f32x4 f = ld4(f1 + ((long)b * S + idx[r * 3 + j]) * C + c);
            v[0] += wj * f[0]; v[1] += wj * f[1]; v[2] += wj * f[2]; v[3] += wj * f[3];
        }
        *reinterpret_cast<f32x4*>(out + (long)r * C + c) = v;
    }
}
// backward: df1[b, idx] += w * dout (df2 = dout is the caller's alias)
__global__ void interp3_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ idx, const float* __restrict__ w, int S,
                                   int N, int C, long rows, float* __restrict__ df1) {
    const long total = rows * C;
    if (total < S3D_U32_LOOP_MAX) {                                  // 32-bit index arithmetic (a 64-bit division is ~3x the instructions)
        const unsigned uC = (unsigned)C, uN = (unsigned)N;
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {
            const unsigned r = i / uC, c = i - r * uC, b = r / uN;
            const float g = dout[i];
#pragma unroll
            for (int j = 0; j < 3; ++j) atomic_add_f32(df1 + ((long)b * S + idx[r * 3 + j]) * (long)C + c, w[r * 3 + j] * g);
        }
        return;
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        const int c = (int)(i % C);
        const long b = r / N;
        const float g = dout[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) atomic_add_f32(df1 + (b * S + idx[r * 3 + j]) * (long)C + c, w[r * 3 + j] * g);
    }
}

// ------------------------------------------------------------------------------------------- grouped first convolution
// sample_and_group builds [B, S, k, 3 + C] rows ([xyz_rel | feats[idx]]) and the level's first 1x1 convolution runs over all
// B*S*k of them.  The convolution is linear and the feature part of a row is a COPY of a point's features, so
//     conv0(row (b, s, j)) = Pf[b, idx] + xyz_rel . Wx^T + bias,      Pf = feats . Wf^T  (one GEMM row per point, not per (s, j))
// k = 16 times fewer GEMM rows, no grouped operand in memory; the 3-column xyz part is done here in fp32 exactly as the
// reference does it (xyz_rel first, then the products).  Thread layout = the BatchNorm vector kernels' (one channel quad per
// thread for the whole launch, row lanes stride the rows): Pf rows are gathered with 16-byte loads (the per-cloud Pf block is
// L2-resident), x rows are written with 16-byte stores.
// Backward: dPf[b, n] = sum of dx over the rows whose neighbour is n -- gathered through the transposed neighbour lists
// (neighbor_csr_kernel), not scattered with atomics: a first version with R x C fp32 atomics (quad-per-thread layout, 16-byte
// strided addresses) took 2.6 ms for cfg-4's first level -- more than the GEMMs it replaced.  dWx / dbias are reduced per
// workgroup, then one atomic per entry; the per-point GEMMs (dWf = dPf^T feats, dfeats += dPf Wf) follow on the MFMA path.
__global__ __launch_bounds__(256) void group_proj_fwd_kernel(const S3dGroupProjArgs p, unsigned rows) {
    const BnLane l = bn_lane(p.C);
    const int c = 4 * l.q;
    double sm[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};          // BatchNorm statistics of the rows this thread writes
    f32x4 wx, wy, wz;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float* w = p.W + (long)(c + i) * p.ldw;
        wx[i] = w[0]; wy[i] = w[1]; wz[i] = w[2];
    }
    const f32x4 bb = ld4(p.bias + c);
    const unsigned SK = (unsigned)p.S * (unsigned)p.K;
    // U rows per trip: the neighbour indices of all U first, then the U gathers they address (a row is idx -> point -> Pf row: two dependent
    // loads; one row per trip left a wave with one 1 KB gather in flight and the launch at 2.4 TB/s of its output alone)
    constexpr int U = 4;
    const unsigned stride = gridDim.x * l.rpb;
    if (l.on)
        for (unsigned r0 = blockIdx.x * l.rpb + l.sub; r0 < rows; r0 += U * stride) {
            long pt[U];
            unsigned rr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                rr[u] = min(r0 + u * stride, rows - 1);
                pt[u] = (long)(rr[u] / SK) * p.N + p.idx[rr[u]];
            }
            f32x4 v[U];
            float rx[U], ry[U], rz[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float* pp = p.xyz + pt[u] * 3;
                const float* cc = p.new_xyz + (long)(rr[u] / (unsigned)p.K) * 3;
                rx[u] = pp[0] - cc[0]; ry[u] = pp[1] - cc[1]; rz[u] = pp[2] - cc[2];
                v[u] = ld4(p.Pf + pt[u] * p.ldp + c);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (r0 + u * stride >= rows) break;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[u][i] = v[u][i] + ((rx[u] * wx[i] + ry[u] * wy[i]) + rz[u] * wz[i]) + bb[i];
                    const double d = v[u][i];
                    sm[i] += d; sq[i] += d * d;
                }
                *reinterpret_cast<f32x4*>(p.x + (long)rr[u] * p.ldx + c) = v[u];
            }
        }
    if (p.sums) bn_fold_sums(l, p.C, sm, sq, p.sums);              // uniform branch; every thread reaches the barrier inside
}

// Transpose of the neighbour graph, one workgroup per cloud: LDS histogram of the in-degrees, block scan, cursor fill, then
// every point sorts its (short) list so that the backward's summation order is deterministic.
__global__ __launch_bounds__(256) void neighbor_csr_kernel(const int* __restrict__ idx, int N, int E, int* __restrict__ inv_off,
                                                           int* __restrict__ inv_rows) {
    extern __shared__ int csr_lds[];                    // cnt [N + 1] | part [256]
    int* cnt = csr_lds;
    int* part = csr_lds + N + 1;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int* id = idx + (long)b * E;
    int* rows = inv_rows + (long)b * E;
    for (int i = tid; i <= N; i += 256) cnt[i] = 0;
    __syncthreads();
    for (int e = tid; e < E; e += 256) atomicAdd(&cnt[id[e]], 1);
    __syncthreads();
    const int per = (N + 255) / 256, lo = tid * per, hi = min(lo + per, N);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += cnt[i];
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {                 // inclusive Hillis-Steele scan of the 256 partial sums
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - sum;                          // exclusive prefix of this thread's first bin
    for (int i = lo; i < hi; ++i) { const int c = cnt[i]; cnt[i] = run; inv_off[(long)b * (N + 1) + i] = run; run += c; }
    if (tid == 255) inv_off[(long)b * (N + 1) + N] = E;
    __syncthreads();
    for (int e = tid; e < E; e += 256) rows[atomicAdd(&cnt[id[e]], 1)] = e;       // cnt is now the fill cursor
    __syncthreads();                                    // the lists are complete (and visible) within the workgroup
    for (int n = tid; n < N; n += 256) {                // cnt[n] = end of list n, start = end of list n - 1
        const int end = cnt[n], start = n ? cnt[n - 1] : 0;
        for (int i = start + 1; i < end; ++i) {
            const int v = rows[i];
            int j = i - 1;
            while (j >= start && rows[j] > v) { rows[j + 1] = rows[j]; --j; }
            rows[j + 1] = v;
        }
    }
}

// Backward, point-centric: C/2 threads per point (a bf16 pair each) walk the rows that reference the point (inv lists), sum their
// dx in fp32 and write dPf once -- no atomics, fixed order.  dWx / dbias: per-thread partials over its points, folded per workgroup.
__global__ __launch_bounds__(256) void group_proj_bwd_kernel(const S3dGroupProjArgs p) {
    __shared__ float red[2048];                         // [point lanes][C][4] : (256 / (C/2)) * C * 4 <= 2048 floats
    const int c2 = p.C >> 1;                            // threads per point (C <= 512 here; wider: loop below)
    const int tpp = min(c2, 256), ppb = 256 / tpp;      // threads per point per pass, points per workgroup
    const int lane = threadIdx.x % tpp, sub = threadIdx.x / tpp;
    const bool on = sub < ppb;
    const long npts = (long)p.B * p.N;
    const int E = p.S * p.K;
    const int kshift = (p.K & (p.K - 1)) == 0 ? __builtin_ctz(p.K) : -1;       // entry -> centroid: e / K
    for (int cbase = 0; cbase < c2; cbase += 256) {     // one pass unless C > 512
        const int cp = cbase + lane;                    // channel pair
        const bool con = on && cp < c2;
        float gw[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, gb[2] = {0.f, 0.f};
        if (con)
            for (long pt = (long)blockIdx.x * ppb + sub; pt < npts; pt += (long)gridDim.x * ppb) {
                const int b = (int)(pt / p.N), n = (int)(pt - (long)b * p.N);
                const int* off = p.inv_off + (long)b * (p.N + 1) + n;
                const int start = off[0], end = off[1];
                const int* rows = p.inv_rows + (long)b * E;
                const float* pp = p.xyz + pt * 3;
                const float px = pp[0], py = pp[1], pz = pp[2];
                float a0 = 0.f, a1 = 0.f;
                auto take = [&](int e, unsigned u) {
                    const float* cc = p.new_xyz + ((long)b * p.S + (kshift >= 0 ? e >> kshift : e / p.K)) * 3;
                    const float g0 = __uint_as_float(u << 16), g1 = __uint_as_float(u & 0xffff0000u);
                    const float rx = px - cc[0], ry = py - cc[1], rz = pz - cc[2];
                    a0 += g0; a1 += g1;
                    gw[0][0] += g0 * rx; gw[0][1] += g0 * ry; gw[0][2] += g0 * rz;
                    gw[1][0] += g1 * rx; gw[1][1] += g1 * ry; gw[1][2] += g1 * rz;
                };
                auto dxw = [&](int e) { return *reinterpret_cast<const unsigned*>(p.dx + ((long)b * E + e) * p.lddx + 2 * cp); };
                int i = start;
                for (; i + 4 <= end; i += 4) {                 // four independent index -> row loads in flight
                    const int e0 = rows[i], e1 = rows[i + 1], e2 = rows[i + 2], e3 = rows[i + 3];
                    const unsigned u0 = dxw(e0), u1 = dxw(e1), u2 = dxw(e2), u3 = dxw(e3);
                    take(e0, u0); take(e1, u1); take(e2, u2); take(e3, u3);
                }
                for (; i < end; ++i) { const int e = rows[i]; take(e, dxw(e)); }
                gb[0] += a0; gb[1] += a1;
                *reinterpret_cast<float2*>(p.dPf + pt * p.ldp + 2 * cp) = make_float2(a0, a1);
            }
        const int cw = min(c2 - cbase, 256) * 2;        // channels covered by this pass
        if (con) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float* o = red + ((long)sub * cw + 2 * lane + i) * 4;
                o[0] = gw[i][0]; o[1] = gw[i][1]; o[2] = gw[i][2]; o[3] = gb[i];
            }
        }
        __syncthreads();
        for (int j = threadIdx.x; j < 4 * cw; j += blockDim.x) {
            float t = 0.f;
            for (int r = 0; r < ppb; ++r) t += red[(long)r * cw * 4 + j];
            const int ch = 2 * cbase + (j >> 2), d = j & 3;
            if (d < 3) atomic_add_f32(p.dW + (long)ch * p.ldw + d, t);
            else atomic_add_f32(p.dbias + ch, t);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------- misc row ops
// mean over the N points of each cloud: out[b][c] = mean_n x[b, n, c]   (x.mean(1), models/3DViT/model.py:325)
// One workgroup per cloud: 256 / C row lanes walk the points with coalesced channel-contiguous reads, LDS fold at the end
// (the first version summed a column per thread with 64 threads per cloud: 245 us for 25 MB).
__global__ __launch_bounds__(256) void mean_points_kernel(const float* __restrict__ x, int N, int C, float* __restrict__ out) {
    __shared__ float red[256];
    const int b = blockIdx.x;
    const int per = C <= 256 ? 256 / C : 1;
    for (int c0 = 0; c0 < C; c0 += 256) {                       // C > 256: one column chunk at a time
        const int cw = min(C - c0, 256);
        const int c = threadIdx.x % cw, sub = threadIdx.x / cw;
        float s = 0.f;
        if (sub < per)
            for (int n = sub; n < N; n += per) s += x[((long)b * N + n) * C + c0 + c];
        red[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x < cw) {
            float t = 0.f;
            for (int r = 0; r < per; ++r) t += red[r * cw + threadIdx.x];
            out[(long)b * C + c0 + threadIdx.x] = t / N;
        }
        __syncthreads();
    }
}
// y[r][c] = scale * x[r / N][c]   (backward of the mean: broadcast dfeat / N)
__global__ void bcast_rows_kernel(const float* __restrict__ x, int N, int C, long rows, float scale, float* __restrict__ y) {
    const long total = rows * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        y[i] = scale * x[(i / C / N) * C + (i % C)];
}
// out[r][0..ld) bf16 planes <- concat / copy of fp32 rows with column padding: generic "pack rows" for GEMM operands
__global__ void pack_rows_kernel(const float* __restrict__ x, int C, int ldx, long rows, bf16_t* __restrict__ hi,
                                 bf16_t* __restrict__ lo, int ldo) {
    const long total = rows * ldo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / ldo;
        const int c = (int)(i % ldo);
        bf16_t h, l;
        split_bf16(c < C ? x[r * ldx + c] : 0.f, h, l);
        hi[i] = h;
        if (lo) lo[i] = l;
    }
}
// quad version: C, ldx, ldo multiples of 4, rows * ldo / 4 < 2^32
__global__ __launch_bounds__(256) void pack_rows_vec_kernel(const float* __restrict__ x, int C, int ldx, unsigned rows, bf16_t* __restrict__ hi,
                                                            bf16_t* __restrict__ lo, int ldo) {
    const unsigned qpr = (unsigned)ldo / 4, total = rows * qpr;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned r = i / qpr, c = 4 * (i - r * qpr);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((int)c < C) v = ld4(x + (long)r * ldx + c);
        union { u32x2 u; uint32_t w[2]; } h, l;
        split_bf16x2(v[0], v[1], h.w[0], l.w[0]);
        split_bf16x2(v[2], v[3], h.w[1], l.w[1]);
        *reinterpret_cast<u32x2*>(hi + (long)r * ldo + c) = h.u;
        if (lo) *reinterpret_cast<u32x2*>(lo + (long)r * ldo + c) = l.u;
    }
}
// a += b (fp32)
__global__ void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) a[i] += b[i];
}
// SGD with momentum over a flat arena (+ split-plane refresh, gradient zeroing): torch.optim.SGD(lr, momentum)
__global__ void sgd_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ buf, bf16_t* __restrict__ hi,
                           bf16_t* __restrict__ lo, long n, float lr, float momentum, float grad_scale, const int* __restrict__ first,
                           const float* __restrict__ hyper) {
    const bool is_first = (*first == 0);
    if (hyper) { lr = hyper[0]; momentum = hyper[1]; grad_scale = hyper[2]; }      // device-resident (graph-replay safe)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gg = g[i] * grad_scale;
        const float bb = is_first ? gg : momentum * buf[i] + gg;
        buf[i] = bb;
        const float pp = p[i] - lr * bb;
        p[i] = pp;
        g[i] = 0.f;
        bf16_t h, l;
        split_bf16(pp, h, l);
        if (hi) hi[i] = h;
        if (lo) lo[i] = l;
    }
}
__global__ void bump_kernel(int* c) { *c += 1; }

// workgroups of the BatchNorm statistics passes (each ends in 2C fp64 atomics): S3D_BN_STATS_BLOCKS
static long bn_stats_blocks() {
    static const long v = s3d_tune_int("S3D_BN_STATS_BLOCKS");              // tuning builds only; 1024 measured best
    return v > 0 ? v : 1024;
}
inline unsigned grid_for(long n, int per = 256, long cap = 8192) {
    long b = (n + per - 1) / per;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

// ---------------------------------------------------------------------------------------------- launchers
int s3d_launch_fps(const float* xyz, long xyz_ld, const long long* start, int B, int N, int npoint, int* out_idx,
                   float* new_xyz, hipStream_t s) {
    S3D_REQUIRE(N <= 256 * FPS_MAX_PPT, "fps: N=%d exceeds %d points per cloud", N, 256 * FPS_MAX_PPT);
    const int lds = (N + 4) * 16;
    if (N <= 256) hipLaunchKernelGGL((fps_kernel<1>), dim3(B), dim3(256), lds, s, xyz, xyz_ld, start, N, npoint, out_idx, new_xyz);
    else if (N <= 512) hipLaunchKernelGGL((fps_kernel<2>), dim3(B), dim3(256), lds, s, xyz, xyz_ld, start, N, npoint, out_idx, new_xyz);
    else if (N <= 1024) hipLaunchKernelGGL((fps_kernel<4>), dim3(B), dim3(256), lds, s, xyz, xyz_ld, start, N, npoint, out_idx, new_xyz);
    else hipLaunchKernelGGL((fps_kernel<8>), dim3(B), dim3(256), lds, s, xyz, xyz_ld, start, N, npoint, out_idx, new_xyz);
    S3D_CHECK_LAUNCH("fps");
    return 0;
}
int s3d_launch_knn(const float* query, const float* ref, int B, int S, int N, int K, int* out_idx, float* out_w, hipStream_t s) {
    S3D_REQUIRE(N <= 64 * KNN_MAX_PPL, "knn: N=%d exceeds %d reference points", N, 64 * KNN_MAX_PPL);
    S3D_REQUIRE(K == 16 || K == 3, "knn: K=%d not built (16, 3)", K);
    const long total = (long)B * S;
    dim3 grid((unsigned)((total + 3) / 4));
#define S3D_KNN(KK, PPL) hipLaunchKernelGGL((knn_kernel<KK, PPL>), grid, dim3(256), 0, s, query, ref, S, N, total, out_idx, out_w)
#define S3D_KNN_N(KK)                      \
    if (N <= 256) S3D_KNN(KK, 4);          \
    else if (N <= 512) S3D_KNN(KK, 8);     \
    else if (N <= 1024) S3D_KNN(KK, 16);   \
    else S3D_KNN(KK, 32);
    if (K == 16) { S3D_KNN_N(16) } else { S3D_KNN_N(3) }
#undef S3D_KNN_N
#undef S3D_KNN
    S3D_CHECK_LAUNCH("knn");
    return 0;
}
int s3d_launch_group_gather(const float* xyz, const float* new_xyz, const float* feats, const int* idx, int B, int N, int S,
                            int K, int C, bf16_t* a_hi, bf16_t* a_lo, int lda, hipStream_t s) {
    const long rows = (long)B * S * K;
    hipLaunchKernelGGL(group_gather_kernel, dim3(grid_for(rows * lda)), dim3(256), 0, s, xyz, new_xyz, feats, idx, N, S, K, C,
                       rows, a_hi, a_lo, lda);
    S3D_CHECK_LAUNCH("group_gather");
    return 0;
}
int s3d_launch_group_scatter(const float* dA, int ldd, const int* idx, int B, int N, int S, int K, int C, float* dfeats, hipStream_t s) {
    const long rows = (long)B * S * K;
    hipLaunchKernelGGL(group_scatter_kernel, dim3(grid_for(rows * C)), dim3(256), 0, s, dA, ldd, idx, N, S, K, C, rows, dfeats);
    S3D_CHECK_LAUNCH("group_scatter");
    return 0;
}
// vector kernels: 4 | C, C/4 <= 256 lanes, 16-byte aligned rows
static bool bn_vec_ok(const S3dBnArgs& a) {
    static const bool off = s3d_tune_int("S3D_BN_SCALAR") >= 0;
    return !off && a.C % 4 == 0 && a.ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
}
static bool al(const void* p, unsigned bytes) { return (reinterpret_cast<uintptr_t>(p) & (bytes - 1)) == 0; }

static int group_proj_check(const S3dGroupProjArgs& a, const char* what) {
    S3D_REQUIRE(a.B > 0 && a.N > 0 && a.S > 0 && a.K > 0, "%s: empty problem", what);
    S3D_REQUIRE(a.C > 0 && a.C % 4 == 0 && a.C <= 1024, "%s: C=%d must be a multiple of 4 up to 1024", what, a.C);
    S3D_REQUIRE((long)a.B * a.S * a.K < (1L << 31), "%s: more than 2^31 grouped rows", what);
    S3D_REQUIRE(a.xyz && a.new_xyz && a.idx && a.W && a.ldw >= 3 && a.ldp % 4 == 0, "%s: null geometry / weight pointer or bad pitch", what);
    return 0;
}
int s3d_launch_group_project_fwd(const S3dGroupProjArgs& a, hipStream_t s) {
    if (int e = group_proj_check(a, "group_project_fwd")) return e;
    S3D_REQUIRE(a.Pf && a.x && a.bias && a.ldx % 4 == 0 && al(a.Pf, 16) && al(a.x, 16) && al(a.bias, 16),
                "group_project_fwd: Pf / x / bias must be non-null, 16-byte aligned, pitches multiples of 4");
    const long rows = (long)a.B * a.S * a.K;
    hipLaunchKernelGGL(group_proj_fwd_kernel, dim3(grid_for(rows, 256 / (a.C / 4), 8192)), dim3(256), 0, s, a, (unsigned)rows);
    S3D_CHECK_LAUNCH("group_project_fwd");
    return 0;
}
int s3d_launch_group_project_bwd(const S3dGroupProjArgs& a, hipStream_t s) {
    if (int e = group_proj_check(a, "group_project_bwd")) return e;
    S3D_REQUIRE(a.dx && a.dPf && a.dW && a.dbias && a.inv_off && a.inv_rows && a.lddx % 2 == 0 && al(a.dx, 4) && al(a.dPf, 8) && a.ldp % 2 == 0,
                "group_project_bwd: dx / dPf / dW / dbias / inv_off / inv_rows must be non-null (s3d_neighbor_csr builds the lists)");
    const long npts = (long)a.B * a.N;
    const int tpp = a.C / 2 < 256 ? a.C / 2 : 256;
    hipLaunchKernelGGL(group_proj_bwd_kernel, dim3(grid_for(npts, 256 / tpp, 2048)), dim3(256), 0, s, a);
    S3D_CHECK_LAUNCH("group_project_bwd");
    return 0;
}
int s3d_launch_neighbor_csr(const int* idx, int B, int N, int S, int K, int* inv_off, int* inv_rows, hipStream_t s) {
    S3D_REQUIRE(B > 0 && N > 0 && S > 0 && K > 0 && N <= 8192, "neighbor_csr: B=%d N=%d S=%d K=%d (N <= 8192)", B, N, S, K);
    const size_t lds = (size_t)(N + 1 + 256) * sizeof(int);
    hipLaunchKernelGGL(neighbor_csr_kernel, dim3(B), dim3(256), lds, s, idx, N, S * K, inv_off, inv_rows);
    S3D_CHECK_LAUNCH("neighbor_csr");
    return 0;
}
// Channel limits: the scalar statistics kernels map 256 / C rows onto a workgroup (C <= 256); the vector kernels give every
// lane one channel quad (C <= 1024, 4 | C, aligned rows) -- deit_small / deit_base widths of the 1-level model need those.
int s3d_launch_bn_fwd(const S3dBnArgs& a, hipStream_t s) {
    S3D_REQUIRE(a.C > 0 && (a.C <= 256 || (a.C <= 1024 && bn_vec_ok(a))),
                "batchnorm: C=%d must be in 1..256 (or a multiple of 4 up to 1024 with 16-byte aligned rows)", a.C);
    const unsigned cblocks = (unsigned)((a.C + 255) / 256);
    if (a.eval_mode) {
        S3D_REQUIRE(a.run_mean && a.run_var, "batchnorm(eval): running statistics required");
        hipLaunchKernelGGL(bn_eval_prepare_kernel, dim3(cblocks), dim3(256), 0, s, a.run_mean, a.run_var, a.C, a.eps, a.mean, a.rstd);
    } else if (a.have_sums) {
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(cblocks), dim3(256), 0, s, a.sums, a.rows, a.C, a.eps, a.momentum, a.mean, a.rstd,
                           a.run_mean, a.run_var, a.momentum_dev);
    } else {
        if (!a.sums_zeroed) (void)hipMemsetAsync(a.sums, 0, 2 * a.C * sizeof(double), s);
        const int per = a.C <= 256 ? 256 / a.C : 1;
        if (bn_vec_ok(a))
            hipLaunchKernelGGL(bn_stats_vec_kernel, dim3(grid_for(a.rows, 256 / (a.C / 4), bn_stats_blocks())), dim3(256), 0, s, a.x, a.rows, a.C, a.ldx, a.sums);
        else
            hipLaunchKernelGGL(bn_stats_kernel, dim3(grid_for(a.rows, per, 1024)), dim3(256), 0, s, a.x, a.rows, a.C, a.ldx, a.sums);
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(cblocks), dim3(256), 0, s, a.sums, a.rows, a.C, a.eps, a.momentum, a.mean, a.rstd,
                           a.run_mean, a.run_var, a.momentum_dev);
    }
    if (a.K > 0) {
        S3D_REQUIRE(a.rows % a.K == 0 && a.ldx == a.C, "batchnorm(max): rows must be groups*K and x compact");
        const long groups = a.rows / a.K;
        if (bn_vec_ok(a) && a.K <= 255 && al(a.y, 16) && al(a.arg, 4))
            hipLaunchKernelGGL(bn_relu_max_vec_kernel, dim3(grid_for(groups, 256 / (a.C / 4), 8192)), dim3(256), 0, s, a.x, groups, a.K,
                               a.C, a.mean, a.rstd, a.gamma, a.beta, a.y, a.arg);
        else
            hipLaunchKernelGGL(bn_relu_max_kernel, dim3(grid_for(groups * a.C)), dim3(256), 0, s, a.x, groups, a.K, a.C, a.mean, a.rstd,
                               a.gamma, a.beta, a.y, a.arg);
    } else {
        if (bn_vec_ok(a) && a.ldo % 4 == 0 && al(a.y, 16) && al(a.y_hi, 8) && al(a.y_lo, 8))
            hipLaunchKernelGGL(bn_relu_vec_kernel, dim3(grid_for(a.rows, 256 / (a.C / 4), 8192)), dim3(256), 0, s, a.x, a.rows, a.C, a.ldx,
                               a.mean, a.rstd, a.gamma, a.beta, a.y, a.y_hi, a.y_lo, a.ldo);
        else
            hipLaunchKernelGGL(bn_relu_kernel, dim3(grid_for(a.rows * a.C)), dim3(256), 0, s, a.x, a.rows, a.C, a.ldx, a.mean, a.rstd,
                               a.gamma, a.beta, a.y, a.y_hi, a.y_lo, a.ldo);
    }
    S3D_CHECK_LAUNCH_V("batchnorm_fwd", (bn_vec_ok(a) ? 1000 : 0) + (a.eval_mode ? 200 : a.have_sums ? 100 : 0) + (a.K > 0 ? 10 : 0));
    return 0;
}
int s3d_launch_bn_bwd(const S3dBnArgs& a, hipStream_t s) {
    const unsigned char* arg = a.K > 0 ? a.arg : nullptr;
    const bool vec = bn_vec_ok(a) && a.lddy % 4 == 0 && a.lddx % 4 == 0 && a.K <= 255 && al(a.dy, 16) && al(a.dy_bf, 8) && al(a.dx, 8) && al(arg, 4);
    S3D_REQUIRE(a.dy != nullptr || a.dy_bf != nullptr, "batchnorm bwd: dy (fp32) or dy_bf (bf16) required");
    S3D_REQUIRE(a.dy_bf == nullptr || vec, "batchnorm bwd: a bf16 upstream gradient needs the vector kernels (C %% 4 == 0, aligned rows)");
    S3D_REQUIRE(a.C > 0 && (a.C <= 256 || (a.C <= 1024 && vec)),
                "batchnorm: C=%d must be in 1..256 (or a multiple of 4 up to 1024 with 16-byte aligned rows)", a.C);
    if (!a.sums_zeroed) (void)hipMemsetAsync(a.sums, 0, 2 * a.C * sizeof(double), s);
    const int per = a.C <= 256 ? 256 / a.C : 1;
    const long n = a.K > 0 ? a.rows / a.K : a.rows;
    if (vec) {
        const int rpb = 256 / (a.C / 4);
        const long sblocks = bn_stats_blocks();
        static const int stats_unroll = s3d_tune_int("S3D_BN_BSTATS_UNROLL");
#define S3D_BN_BSTATS(U)                                                                                                                   \
    hipLaunchKernelGGL(bn_bwd_stats_vec_kernel<U>, dim3(grid_for(n, rpb, sblocks)), dim3(256), 0, s, a.x, a.ldx, a.dy, a.lddy, arg, a.K, a.rows, \
                       a.C, a.mean, a.rstd, a.gamma, a.beta, a.sums, a.dy_bf)
        if (stats_unroll == 1) S3D_BN_BSTATS(1);
        else if (stats_unroll == 2) S3D_BN_BSTATS(2);
        else if (stats_unroll == 8) S3D_BN_BSTATS(8);
        else S3D_BN_BSTATS(4);
#undef S3D_BN_BSTATS
        static const int apply_blocks = s3d_tune_int("S3D_BN_APPLY_BLOCKS"), apply_unroll = s3d_tune_int("S3D_BN_APPLY_UNROLL");
        const dim3 grid(grid_for(a.rows, rpb, apply_blocks > 0 ? apply_blocks : 8192));
        const double inv_rows = 1.0 / (double)a.rows;
#define S3D_BN_APPLY(U)                                                                                                                  \
    hipLaunchKernelGGL(bn_bwd_apply_vec_kernel<U>, grid, dim3(256), 0, s, a.x, a.ldx, a.dy, a.lddy, arg, a.K, a.rows, a.C, a.mean, a.rstd, \
                       a.gamma, a.beta, a.sums, a.dx, a.lddx, a.dgamma, a.dbeta, a.dy_bf, inv_rows)
        if (apply_unroll == 4) S3D_BN_APPLY(4);
        else if (apply_unroll == 2) S3D_BN_APPLY(2);
        else if (apply_unroll == 8) S3D_BN_APPLY(8);
        else S3D_BN_APPLY(1);
#undef S3D_BN_APPLY
        S3D_CHECK_LAUNCH_V("batchnorm_bwd", 100 + (a.K > 0 ? 10 : 0) + (a.dy_bf ? 1 : 0));
        return 0;
    }
    hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(grid_for(n, per, 1024)), dim3(256), 0, s, a.x, a.ldx, a.dy, a.lddy, arg, a.K,
                       a.rows, a.C, a.mean, a.rstd, a.gamma, a.beta, a.sums);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(a.rows * a.C)), dim3(256), 0, s, a.x, a.ldx, a.dy, a.lddy, arg, a.K,
                       a.rows, a.C, a.mean, a.rstd, a.gamma, a.beta, a.sums, a.dx, a.lddx, a.dgamma, a.dbeta);
    S3D_CHECK_LAUNCH("batchnorm_bwd");
    return 0;
}
int s3d_launch_interp3(const float* f1, int S, const float* f2, const int* idx, const float* w, int B, int N, int C, float* out,
                       hipStream_t s) {
    const long rows = (long)B * N;
    if (C % 4 == 0 && rows * (C / 4) < S3D_U32_LOOP_MAX && al(f1, 16) && al(f2, 16) && al(out, 16)) {
        hipLaunchKernelGGL(interp3_vec_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, s, f1, S, f2, idx, w, N, C, (unsigned)rows, out);
        S3D_CHECK_LAUNCH("interp3");
        return 0;
    }
    hipLaunchKernelGGL(interp3_kernel, dim3(grid_for(rows * C)), dim3(256), 0, s, f1, S, f2, idx, w, N, C, rows, out);
    S3D_CHECK_LAUNCH("interp3");
    return 0;
}
int s3d_launch_interp3_bwd(const float* dout, const int* idx, const float* w, int B, int S, int N, int C, float* df1, hipStream_t s) {
    const long rows = (long)B * N;
    // (a channel-quad version of this kernel is 2x slower: four atomics per lane at 16-byte strides instead of one coalesced row)
    hipLaunchKernelGGL(interp3_bwd_kernel, dim3(grid_for(rows * C)), dim3(256), 0, s, dout, idx, w, S, N, C, rows, df1);
    S3D_CHECK_LAUNCH("interp3_bwd");
    return 0;
}
int s3d_launch_mean_points(const float* x, int B, int N, int C, float* out, hipStream_t s) {
    hipLaunchKernelGGL(mean_points_kernel, dim3(B), dim3(256), 0, s, x, N, C, out);
    S3D_CHECK_LAUNCH("mean_points");
    return 0;
}
int s3d_launch_bcast_rows(const float* x, int N, int C, long rows, float scale, float* y, hipStream_t s) {
    hipLaunchKernelGGL(bcast_rows_kernel, dim3(grid_for(rows * C)), dim3(256), 0, s, x, N, C, rows, scale, y);
    S3D_CHECK_LAUNCH("bcast_rows");
    return 0;
}
int s3d_launch_pack_rows(const float* x, int C, int ldx, long rows, bf16_t* hi, bf16_t* lo, int ldo, hipStream_t s) {
    if (C % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && rows * (ldo / 4) < S3D_U32_LOOP_MAX && al(x, 16) && al(hi, 8) && al(lo, 8)) {
        hipLaunchKernelGGL(pack_rows_vec_kernel, dim3(grid_for(rows * (ldo / 4))), dim3(256), 0, s, x, C, ldx, (unsigned)rows, hi, lo, ldo);
        S3D_CHECK_LAUNCH("pack_rows");
        return 0;
    }
    hipLaunchKernelGGL(pack_rows_kernel, dim3(grid_for(rows * ldo)), dim3(256), 0, s, x, C, ldx, rows, hi, lo, ldo);
    S3D_CHECK_LAUNCH("pack_rows");
    return 0;
}
int s3d_launch_add_inplace(float* a, const float* b, long n, hipStream_t s) {
    hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for(n)), dim3(256), 0, s, a, b, n);
    S3D_CHECK_LAUNCH("add_inplace");
    return 0;
}
int s3d_launch_sgd(float* p, float* g, float* buf, bf16_t* hi, bf16_t* lo, long n, float lr, float momentum, float grad_scale,
                   int* step_counter, const float* hyper, hipStream_t s) {
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, s, p, g, buf, hi, lo, n, lr, momentum, grad_scale,
                       step_counter, hyper);
    hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, s, step_counter);
    S3D_CHECK_LAUNCH("sgd");
    return 0;
}
