// Fused launches of the timm Block forward for the small-batch voxel configurations (cfg-1 / cfg-2: M = B * 26 = 1664 token rows).
//
//   blk_attn_kernel   norm1 -> qkv head slice -> softmax(q k^T) v      one workgroup per (pair of samples, head)
//   blk_mlp1_kernel   norm2 -> fc1 hidden slice -> GELU                one workgroup per (64-row band, 192-wide hidden slice)
//
// (timm==0.3.2 Block / Attention / Mlp forward as restated in oracle/timm_shim/timm/models/vision_transformer.py:41-78, invoked at
// models/vit_3d_2d_pretrain.py:466-469.)  Together with the attn.proj and mlp.fc2 GEMM launches a block forward is four launches
// instead of seven: the LayerNorm launches, the attention launch and the round trips of xn1 / qkv / xn2 through memory as split-bf16
// GEMM operands are gone.
//
// Both kernels are the same "skinny GEMM": 64 activation rows x 192 weight rows x K = D, one workgroup of four waves per CU, where
//   * the A operand never exists in memory as a GEMM operand: every thread loads 1/4 of one residual-stream row (fp32), the row
//     statistics are a 4-lane DPP reduction, and the normalised row goes straight into LDS as the two bf16 planes of the WHOLE
//     [64][D] operand (96 KB at D = 384).  The LayerNorm is recomputed by every workgroup that shares the rows (H = 6 resp.
//     hidden / 192 = 8 times) -- the round-2 consumer-side fusion into the generic GEMM redid it 18 - 24 times, once per tile column;
//   * the B operand (a 192-row slice of the weight planes, 147 KB per plane at D = 384) is used by ONE workgroup for 64 rows only,
//     so it does not go through LDS at all: every wave loads the fragments of its own 48 weight rows straight from global memory
//     (L2) into MFMA operand registers, PS 64-wide k-slabs (12 KB per wave each) ahead of their use -- ~100 KB in flight per CU in
//     registers, which no LDS ring next to the A operand could hold.  That only works on a weight copy stored in fragment order
//     (s3d_pack_weights; refreshed by the optimizer kernel): straight from the row-major planes the 16 lanes of a quarter wave
//     read 16 different weight rows, every wave-load touches 16 - 64 cache lines, and the texture-address path, not bandwidth,
//     becomes the bound (measured: 20.4 / 22.1 us per launch, slower than the LDS-DMA ring);
//   * with both operands private to a wave or read-only, the main loop has NO barrier and no counted waits: 12 k-steps of 8
//     ds_read_b128 + 36 MFMAs (split-bf16 product = hi*lo + lo*hi + hi*hi on 16x16x32) per wave, waves free-running.
//     (First version, measured: weights through a four-stage LDS-DMA ring with a barrier per k-step, eight waves -- 17.6 / 19.3 us
//     per launch, i.e. the ~40 GB/s a single lock-stepped workgroup pulls per CU; profiles/r03_fused_v1_kernel_stats.txt.)
//   * workgroups that share a weight slice are mapped to the same XCD, so the slice is fetched into one L2 and re-read there.
//
// Why there is no fc2 / proj inside these launches: their reductions run over the hidden slices / heads, i.e. across workgroups.
// Accumulating the partial tiles into the residual stream with fp32 atomics was measured first (tools/probes/atomic_resid_probe.hip,
// profiles/r03_atomic_resid_probe.txt): 1.0 - 1.3 TB/s whatever the contention or access shape, i.e. 15 - 19 us for the 20 MB of
// partials of ONE mlp.fc2 -- more than the whole fc2 GEMM launch (14.5 us).  Every scope of global_atomic_add_f32 lowers to the same
// memory-side instruction on gfx950 (per-XCD L2s are not coherent), so there is no cheaper on-chip variant.
#include "fused_block.h"

#include "attn_frag.h"
#include "gemm.h"
#include "dma_tile.h"

#include <type_traits>

// ---- in-kernel timeline (make TL=1 only; tools/fused_timeline_probe.py): wave 0 / lane 0 of every workgroup stamps s_memtime at the
// phase boundaries into [workgroup][16] of a caller-provided buffer
#ifdef S3D_TIMELINE
__device__ unsigned long long* g_fb_tl = nullptr;
extern "C" int s3d_debug_fused_timeline_set(void* buf) {
    unsigned long long* b = reinterpret_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_fb_tl), &b, sizeof(b)) == hipSuccess ? 0 : 1;
}
__device__ int g_fb_tl_only = 0;          // 0: stamp blk_attn_kernel, 1: blk_mlp1_kernel
extern "C" int s3d_debug_fused_timeline_only(int which) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_fb_tl_only), &which, sizeof(which)) == hipSuccess ? 0 : 1;
}
#define FB_STAMP(i) do { if (g_fb_tl && threadIdx.x == 0 && g_fb_tl_only == FB_KERNEL_ID) g_fb_tl[(long)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FB_STAMP(i) do {} while (0)
#endif

namespace {

constexpr int FB_THREADS = 256;          // 4 waves, each on all 64 rows x 48 of the 192 weight rows
constexpr int FB_ROWS = 64;              // activation rows per workgroup
constexpr int FB_WROWS = 192;            // weight rows (output columns) per workgroup
constexpr int FB_APLANE = FB_ROWS * 128;             // 8 192 B: one plane of an A slab ([64 rows][64 k], 128-byte rows)
constexpr int FB_ASLAB = 2 * FB_APLANE;              // hi + lo

constexpr int fb_max(int a, int b) { return a > b ? a : b; }
constexpr int fb_lds_bytes(int D, int staging) { return fb_max((D / 64) * FB_ASLAB, staging); }

// sum over the 16 lanes of a DPP row (four butterfly steps), result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
    int x = __float_as_int(v);
#define S3D_ROW_STEP(ctrl) x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(x, x, ctrl, 0xf, 0xf, false)));
    S3D_ROW_STEP(0xB1)    // quad_perm:[1,0,3,2]
    S3D_ROW_STEP(0x4E)    // quad_perm:[2,3,0,1]
    S3D_ROW_STEP(0x141)   // row_half_mirror
    S3D_ROW_STEP(0x140)   // row_mirror
#undef S3D_ROW_STEP
    return __int_as_float(x);
}

// LayerNorm + split + streamed-weight GEMM of a 64 x 192 output tile; leaves the accumulators of this wave's 64 x 48 sub-tile in
// `acc` (swapped operand roles: lane l owns row 16 i + (l & 15), columns 48 wave + 16 j + 4 (l >> 4) .. + 3).  On return other waves
// may still be reading the A operand: callers put a barrier in front of any LDS reuse.
//   rows      functor: rows.grow(r) = global row of tile row r (clamped to a valid one), rows.ok(r) = tile row r exists
//   x         residual stream [.][D];  mean_out / rstd_out (nullable) [.] receive the statistics of the valid rows
//   w_hi/lo   PACKED weight planes of the [.][D] matrix; wrow0..2 = first weight row of the three 64-row blocks of this slice
//   xn_hi/lo  (nullable) normalised planes [.][D]: this workgroup stores the 64 columns of k-slab kp_store
// Row loads: 16 lanes per row, every quarter wave reads 256 consecutive bytes (the first version gave each row 4 lanes with 64
// bytes each: 16-byte pieces at a 64-byte stride, 64 segments per wave-load -- 9 - 10 k cycles for the 96 KB of a tile, in-kernel
// timeline profiles/r03_fused_timeline.txt).  Plain scalars / a by-value functor, not a parameter struct: a select between two struct
// members becomes a select between their ADDRESSES and drags the struct into scratch memory (the round-1 lesson of common.h).
template <int D, int FB_KERNEL_ID, typename Rows>
__device__ __forceinline__ void ln_gemm_64x192(const Rows rows, const float* x, const float* gamma, const float* beta, const float eps,
                                               float* mean_out, float* rstd_out, const bf16_t* w_hi, const bf16_t* w_lo, const long wrow0,
                                               const long wrow1, const long wrow2, bf16_t* xn_hi, bf16_t* xn_lo, const int kp_store,
                                               unsigned char* smem, f32x4 (&acc)[4][3]) {
    constexpr int KP = D / 64;                                             // 64-wide k-slabs
    // slabs of weight fragments in flight ahead of their use (48 registers each): two while the residual rows occupy 96 + 48 registers,
    // three from the end of the LayerNorm phase on (the main loop runs at the rate the fragments arrive: 30 B/clk/CU with two)
    constexpr int PS0 = KP < 2 ? KP : 2, PS = KP < 3 ? KP : 3;
    static_assert(D % 64 == 0, "model dimension");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int l16 = lane & 15;                                             // LayerNorm phase: tile row 16 pass + 4 wave + g, columns 64 j + 4 l16 ..

    FB_STAMP(0);
    float xr[4][KP][4];
    long grow[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        grow[ps] = rows.grow(16 * ps + 4 * wave + g);
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(x + grow[ps] * D + 64 * j + 4 * l16);
            xr[ps][j][0] = t[0]; xr[ps][j][1] = t[1]; xr[ps][j][2] = t[2]; xr[ps][j][3] = t[3];
        }
    }
    f32x4 gv[KP], bv[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) {
        gv[j] = *reinterpret_cast<const f32x4*>(gamma + 64 * j + 4 * l16);
        bv[j] = *reinterpret_cast<const f32x4*>(beta + 64 * j + 4 * l16);
    }
    // ---- weight fragments from the PACKED planes (s3d_pack_weights): the 16 x 32 block (rows 16 nb .., k 32 ks ..) of the weight matrix
    // is 1 KB in MFMA operand order -- lane l = (n = l & 15, g = l >> 4) finds W[16 nb + n][32 ks + 8 g .. + 7] at byte 16 l -- and
    // the blocks of one nb follow each other in k: a wave-load is 1 KB of consecutive bytes, a slab (two k-steps) 2 KB.
    const bf16_t* wp[2][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int cb = wave * 3 + j;                                       // 16-row block of the workgroup's 192 weight rows
        const long nb = ((cb >> 2) == 0 ? wrow0 : (cb >> 2) == 1 ? wrow1 : wrow2) / 16 + (cb & 3);
        const long off = nb * (D / 32) * 512 + lane * 8;
        wp[0][j] = w_hi + off; wp[1][j] = w_lo + off;
    }
    bf16x8 bw[KP][2][3][2];                                                // [slab][plane][n-block][k-step]; live: PS + 1 slabs
    auto fetch = [&](auto kp_tag) {
        constexpr int kp = decltype(kp_tag)::value;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) bw[kp][pl][j][q] = *reinterpret_cast<const bf16x8*>(wp[pl][j] + (2 * kp + q) * 512);
    };
    if constexpr (PS0 > 0) fetch(std::integral_constant<int, 0>{});
    if constexpr (PS0 > 1) fetch(std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);

    // ---- the whole A operand: slab j = [hi plane [64][64]][lo plane], 128-byte rows, 16-byte chunk c at slot c ^ dma_swz64(row):
    // k-step q reads chunks 4 q + (lane >> 4) exactly like the DMA GEMM tiles (conflict-free, dma_tile.h)
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int r = 16 * ps + 4 * wave + g;
        float s1 = 0.f;
#pragma unroll
        for (int j = 0; j < KP; ++j) s1 += (xr[ps][j][0] + xr[ps][j][1]) + (xr[ps][j][2] + xr[ps][j][3]);
        constexpr float inv_d = 1.0f / (float)D;
        const float mean = row16_sum(s1) * inv_d;
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < KP; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float d = xr[ps][j][i] - mean; s2 += d * d; }
        const float rstd = rsqrtf(row16_sum(s2) * inv_d + eps);
        const bool ok = rows.ok(r);
        if (mean_out != nullptr && l16 == 0 && ok) { mean_out[grow[ps]] = mean; rstd_out[grow[ps]] = rstd; }
        unsigned char* at = smem + r * 128 + (((l16 >> 1) ^ dma_swz64(r)) << 4) + 8 * (l16 & 1);
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            float y[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = (xr[ps][j][i] - mean) * (rstd * gv[j][i]) + bv[j][i];
            u32x2 hi, lo;
            uint32_t h2, l2;
            split_bf16x2(y[0], y[1], h2, l2); hi[0] = h2; lo[0] = l2;
            split_bf16x2(y[2], y[3], h2, l2); hi[1] = h2; lo[1] = l2;
            *reinterpret_cast<u32x2*>(at + j * FB_ASLAB) = hi;
            *reinterpret_cast<u32x2*>(at + j * FB_ASLAB + FB_APLANE) = lo;
            if (j == kp_store && xn_hi != nullptr && ok) {                 // block-uniform on kp_store / the pointer
                *reinterpret_cast<u32x2*>(xn_hi + grow[ps] * D + 64 * j + 4 * l16) = hi;
                *reinterpret_cast<u32x2*>(xn_lo + grow[ps] * D + 64 * j + 4 * l16) = lo;
            }
        }
    }
    FB_STAMP(1);
    if constexpr (PS > PS0) {
        __builtin_amdgcn_sched_barrier(0);
        fetch(std::integral_constant<int, PS0>{});
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                                       // the A operand is complete
    FB_STAMP(2);

    // ---- main loop: no barrier, no LDS writes; weight fragments PS slabs ahead
    auto slab = [&](auto kp_tag) {
        constexpr int kp = decltype(kp_tag)::value;
        if constexpr (kp + PS < KP) {
            fetch(std::integral_constant<int, kp + PS>{});
            __builtin_amdgcn_sched_barrier(0);                             // the loads stay here, a whole slab of MFMAs ahead of their use
        }
        const unsigned char* sA = smem + kp * FB_ASLAB;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bf16x8 a_hi[4], a_lo[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = i * 16 + n;
                const unsigned char* pa = sA + r * 128 + (((4 * q + g) ^ dma_swz64(r)) << 4);
                a_hi[i] = *reinterpret_cast<const bf16x8*>(pa);
                a_lo[i] = *reinterpret_cast<const bf16x8*>(pa + FB_APLANE);
            }
            // term-outer: the three MFMAs of an output block are 12 instructions apart (back to back they wait for each other's result:
            // 22 cycles per MFMA in the first timeline instead of the ~16 the matrix pipe issues at)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[kp][0][j][q], a_lo[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[kp][1][j][q], a_hi[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[kp][0][j][q], a_hi[i], acc[i][j], 0, 0, 0);
        }
    };
    slab(std::integral_constant<int, 0>{});
    if constexpr (KP > 1) slab(std::integral_constant<int, 1>{});
    if constexpr (KP > 2) slab(std::integral_constant<int, 2>{});
    if constexpr (KP > 3) slab(std::integral_constant<int, 3>{});
    if constexpr (KP > 4) slab(std::integral_constant<int, 4>{});
    if constexpr (KP > 5) slab(std::integral_constant<int, 5>{});
    static_assert(KP <= 6, "unrolled for D <= 384");
    FB_STAMP(3);                                                           // main loop done (this wave)
}

// contiguous runs of items per XCD (the dispatcher places workgroup b on XCD b % 8): workgroups that stream the same weight slice
// share an L2.  Pure speed; any placement is correct.
__device__ __forceinline__ int xcd_item(int bid, int nitem) {
    const int q = nitem >> 3, r = nitem & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// norm1 -> qkv -> attention for TWO samples and one head (hd = 64, N <= 32 tokens): rows 0..31 of the tile are sample 2 pr, rows
// 32..63 sample 2 pr + 1.  Writes what the backward reads: xn1 planes (its head's 64 columns), mean1 / rstd1 (head 0), the bf16 q | k | v
// head slices, the attention output planes and the log-sum-exp.
constexpr int QKV_PITCH = 72;            // bf16 elements per staged q / k / v row (144 B: 16-byte aligned, 8-byte transposed reads)
constexpr int QKV_TILE = 32 * QKV_PITCH * 2;                               // bytes per staged [32 tokens][64] plane
constexpr int QKV_STAGING = 12 * QKV_TILE;                                 // [sample][q,k,v][hi,lo]

template <int D>
__global__ __launch_bounds__(FB_THREADS) void blk_attn_kernel(const FusedAttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int npair = (p.Bb + 1) >> 1;
    const int item = xcd_item(blockIdx.x, npair * p.H);
    const int h = item / npair, pr = item % npair;                         // head-major: an XCD sees one or two heads' weights
    struct Rows {                                                           // tile row r = sample (r >> 5) of the pair, token r & 31
        int b0, Bb, N;
        __device__ __forceinline__ long grow(int r) const { return (long)min(b0 + (r >> 5), Bb - 1) * N + min(r & 31, N - 1); }   // clamped duplicates are never stored
        __device__ __forceinline__ bool ok(int r) const { return b0 + (r >> 5) < Bb && (r & 31) < N; }
    };
    f32x4 acc[4][3];
    constexpr int FB_KERNEL_ID = 0;
    ln_gemm_64x192<D, 0>(Rows{2 * pr, p.Bb, p.N}, p.x, p.gamma, p.beta, p.eps, h == 0 ? p.mean : nullptr, p.rstd, p.w_hi, p.w_lo, 64 * h,
                         D + 64 * h, 2 * D + 64 * h, p.xn_hi, p.xn_lo, h, smem, acc);      // xn1: this head's 64 columns

    // ---- q | k | v (+ bias) -> LDS tiles [sample][q,k,v][hi,lo][32 tokens][QKV_PITCH] over the (consumed) A operand
    constexpr int TILE = QKV_TILE;
    __syncthreads();                                                       // every wave is done reading the A operand
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int nn = wave * 48 + j * 16 + (lane >> 4) * 4;               // 4 consecutive columns of q | k | v
        const int which = nn >> 6, d = nn & 63;
        const f32x4 bq = *reinterpret_cast<const f32x4*>(p.bias + which * D + 64 * h + d);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sm = i >> 1, t = (i & 1) * 16 + (lane & 15);         // sample of the pair, token
            u32x2 hi, lo;
            uint32_t h2, l2;
            split_bf16x2(acc[i][j][0] + bq[0], acc[i][j][1] + bq[1], h2, l2); hi[0] = h2; lo[0] = l2;
            split_bf16x2(acc[i][j][2] + bq[2], acc[i][j][3] + bq[3], h2, l2); hi[1] = h2; lo[1] = l2;
            unsigned char* q = smem + ((sm * 3 + which) * 2) * TILE + (t * QKV_PITCH + d) * 2;
            *reinterpret_cast<u32x2*>(q) = hi;
            *reinterpret_cast<u32x2*>(q + TILE) = lo;
        }
    }
    __syncthreads();
    FB_STAMP(4);
    // ---- q | k | v hi planes -> memory (operands of the attention backward and of the qkv wgrad), 16 bytes per thread and trip
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int e = tid + FB_THREADS * i;                                // 6 tiles x 32 tokens x 8 chunks
        const int tile = e >> 8, t = (e >> 3) & 31, ch = e & 7;
        const int sb = tile / 3, which = tile % 3, bb = 2 * pr + sb;
        if (bb < p.Bb && t < p.N) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + (tile * 2) * TILE + (t * QKV_PITCH + 8 * ch) * 2);
            *reinterpret_cast<u32x4*>(p.qkv_hi + ((long)bb * p.N + t) * (3 * D) + which * D + 64 * h + 8 * ch) = v;
        }
    }
    FB_STAMP(5);
    // ---- attention: wave w < 2 owns sample w.  S^T = K Q^T on 32x32x16 MFMAs (every lane owns one query column), one key tile
    if (wave >= 2) return;
    const int bb = 2 * pr + wave;
    if (bb >= p.Bb) return;
    const int h2 = lane >> 5, l31 = lane & 31;
    const bf16_t* tq = reinterpret_cast<const bf16_t*>(smem + ((wave * 3 + 0) * 2) * TILE);
    const bf16_t* tk = reinterpret_cast<const bf16_t*>(smem + ((wave * 3 + 1) * 2) * TILE);
    const bf16_t* tv = reinterpret_cast<const bf16_t*>(smem + ((wave * 3 + 2) * 2) * TILE);
    constexpr int TEL = TILE / 2;                                          // elements per tile: the lo plane follows the hi plane
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int off = l31 * QKV_PITCH + h2 * 8 + 16 * s;
        const bf16x8 qh = *reinterpret_cast<const bf16x8*>(tq + off), ql = *reinterpret_cast<const bf16x8*>(tq + TEL + off);
        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(tk + off), kl = *reinterpret_cast<const bf16x8*>(tk + TEL + off);
        sacc = MFMA32(kl, qh, sacc);
        sacc = MFMA32(kh, ql, sacc);
        sacc = MFMA32(kh, qh, sacc);
    }
    float sv[16], mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        sv[r] = acc_row(r, h2) < p.N ? sacc[r] * p.scale : -INFINITY;
        mx = fmaxf(mx, sv[r]);
    }
    mx = half_max(mx);                                                     // finite: key 0 is always valid
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sv[r] = fast_exp(sv[r] - mx); l += sv[r]; }
    l = half_sum(l);
    U128 ph[2], pl[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; j += 2) split_bf16x2(sv[8 * s2 + j], sv[8 * s2 + j + 1], ph[s2].w[j / 2], pl[s2].w[j / 2]);
    f32x16 o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
        bf16x8 vh[2], vl[2];
        gather_frag_2x2<64, QKV_PITCH>(tv, d * 32 + l31, tv + TEL, d * 32 + l31, h2, vh, vl);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            o[d] = MFMA32(vl[s2], ph[s2].v, o[d]);
            o[d] = MFMA32(vh[s2], pl[s2].v, o[d]);
            o[d] = MFMA32(vh[s2], ph[s2].v, o[d]);
        }
    }
    if (l31 < p.N) {
        const float inv = 1.0f / l;
        const long orow = ((long)bb * p.N + l31) * D + 64 * h;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int c = 0; c < 4; ++c) {                                  // rows 8c + 4 h2 + {0..3} of the d-block: 4 consecutive d
                u32x2 hi, lo;
                uint32_t a, b2;
                split_bf16x2(o[d][4 * c] * inv, o[d][4 * c + 1] * inv, a, b2); hi[0] = a; lo[0] = b2;
                split_bf16x2(o[d][4 * c + 2] * inv, o[d][4 * c + 3] * inv, a, b2); hi[1] = a; lo[1] = b2;
                const long off = orow + d * 32 + 8 * c + 4 * h2;
                *reinterpret_cast<u32x2*>(p.att_hi + off) = hi;
                *reinterpret_cast<u32x2*>(p.att_lo + off) = lo;
            }
        if (h2 == 0) {
            const long li = p.lse_packed ? ((long)(bb >> 1) * p.H + h) * (2 * p.N) + (bb & 1) * p.N + l31 : ((long)bb * p.H + h) * p.N + l31;
            p.lse[li] = mx + logf(l);
        }
    }
    FB_STAMP(6);
#ifdef S3D_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FB_STAMP(7);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// norm2 -> fc1 slice -> GELU for a band of 64 token rows and 192 hidden units.  Writes xn2 planes (slice s its 64 columns 64 s ..),
// mean2 / rstd2 (slice 0), the bf16 pre-activation and the split planes of gelu(pre) -- the A operand of the mlp.fc2 launch.
constexpr int H_PITCH = 200;             // bf16 elements per staged output row (400 B: 16-byte aligned, rows 4 banks apart)
constexpr int H_TILE = FB_ROWS * H_PITCH * 2;                              // bytes per staged [64][192] bf16 array
constexpr int H_STAGING = 3 * H_TILE;

template <int D>
__global__ __launch_bounds__(FB_THREADS) void blk_mlp1_kernel(const FusedMlpArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int RB = p.band_rows;                                            // token rows per band (<= 64)
    const int nband = (int)((p.M + RB - 1) / RB);
    const int item = xcd_item(blockIdx.x, nband * p.nslice);
    const int js = item / nband, band = item % nband;                      // slice-major: the bands of a slice share an XCD
    const long m0 = (long)band * RB;
    struct Rows {
        long m0, M; int RB;
        __device__ __forceinline__ long grow(int r) const { return min(m0 + min(r, RB - 1), M - 1); }
        __device__ __forceinline__ bool ok(int r) const { return r < RB && m0 + r < M; }
    };
    f32x4 acc[4][3];
    constexpr int FB_KERNEL_ID = 1;
    const long w0 = (long)FB_WROWS * js;
    const bool st = js < D / 64;                                           // slices 0 .. D / 64 - 1 store one 64-column slab of xn2 each
    ln_gemm_64x192<D, 1>(Rows{m0, p.M, RB}, p.x, p.gamma, p.beta, p.eps, js == 0 ? p.mean : nullptr, p.rstd, p.w_hi, p.w_lo, w0, w0 + 64,
                         w0 + 128, st ? p.xn_hi : nullptr, p.xn_lo, js, smem, acc);

    // ---- epilogue: pre = acc + b1 -> bf16; gelu(pre) -> split planes; staged over the (consumed) A operand for 16-byte row stores
    constexpr int TILE = H_TILE;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int nn = wave * 48 + j * 16 + (lane >> 4) * 4;
        const f32x4 bq = *reinterpret_cast<const f32x4*>(p.bias + FB_WROWS * js + nn);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = i * 16 + (lane & 15);
            const float p0 = acc[i][j][0] + bq[0], p1 = acc[i][j][1] + bq[1], p2 = acc[i][j][2] + bq[2], p3 = acc[i][j][3] + bq[3];
            u32x2 pre, hi, lo;
            uint32_t a, b2;
            pre[0] = f2bf2(p0, p1); pre[1] = f2bf2(p2, p3);
            split_bf16x2(gelu_erf(p0), gelu_erf(p1), a, b2); hi[0] = a; lo[0] = b2;
            split_bf16x2(gelu_erf(p2), gelu_erf(p3), a, b2); hi[1] = a; lo[1] = b2;
            unsigned char* q = smem + (r * H_PITCH + nn) * 2;
            *reinterpret_cast<u32x2*>(q) = pre;
            *reinterpret_cast<u32x2*>(q + TILE) = hi;
            *reinterpret_cast<u32x2*>(q + 2 * TILE) = lo;
        }
    }
    __syncthreads();
    FB_STAMP(4);
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        const int e = tid + FB_THREADS * i;                                // 3 arrays x 64 rows x 24 chunks of 16 bytes
        const int arr = e / (FB_ROWS * 24), rc = e % (FB_ROWS * 24), r = rc / 24, ch = rc % 24;
        if (r < RB && m0 + r < p.M) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + arr * TILE + (r * H_PITCH + 8 * ch) * 2);
            bf16_t* const out = arr == 0 ? p.hpre : arr == 1 ? p.hact_hi : p.hact_lo;
            *reinterpret_cast<u32x4*>(out + (m0 + r) * p.hidden + FB_WROWS * js + 8 * ch) = v;
        }
    }
    FB_STAMP(6);
#ifdef S3D_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FB_STAMP(7);
#endif
}

// row-major planes [rows][K] -> fragment order: block (nb, ks) = rows 16 nb .., k 32 ks .. at element offset (nb * K / 32 + ks) * 512,
// inside it lane l = (n, g) holds 8 consecutive k of row n at element 8 l.  One thread moves one 16-byte piece of each plane.
__global__ __launch_bounds__(256) void pack_weights_kernel(const bf16_t* __restrict__ src_hi, const bf16_t* __restrict__ src_lo,
                                                           bf16_t* __restrict__ dst_hi, bf16_t* __restrict__ dst_lo, int rows, int K) {
    const long pieces = (long)rows * (K / 8);
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < pieces; e += (long)gridDim.x * 256) {
        const int l = (int)(e & 63);
        const long blk = e >> 6;
        const int kb = K / 32;
        const long nb = blk / kb;
        const int ks = (int)(blk % kb);
        const long src = (16 * nb + (l & 15)) * K + 32 * ks + 8 * (l >> 4);
        *reinterpret_cast<u32x4*>(dst_hi + 8 * e) = *reinterpret_cast<const u32x4*>(src_hi + src);
        if (src_lo) *reinterpret_cast<u32x4*>(dst_lo + 8 * e) = *reinterpret_cast<const u32x4*>(src_lo + src);
    }
}

template <typename K>
int set_lds(K kern, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 0 : 1;
}

template <int D>
int launch_attn(const FusedAttnArgs& a, hipStream_t s) {
    constexpr int LDS = fb_lds_bytes(D, QKV_STAGING);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static const int once = set_lds(blk_attn_kernel<D>, LDS);
    (void)once;
    const int grid = ((a.Bb + 1) / 2) * a.H;
    constexpr long long KEY = 700000000000LL + D;                          // bench.py: 7 = fused norm1 + qkv + attention
    if (s3d_prof_skipped(KEY)) return 0;
    const double M = (double)a.Bb * a.N;
    s3d_prof_begin(KEY, 2.0 * M * 3 * D * D + 4.0 * a.Bb * a.N * a.N * D, s);       // qkv GEMM + (q k^T, p v) of every head
    hipLaunchKernelGGL((blk_attn_kernel<D>), dim3(grid), dim3(FB_THREADS), LDS, s, a);
    s3d_prof_end(s);
    S3D_CHECK_LAUNCH("blk_attn");
    return 0;
}
template <int D>
int launch_mlp1(const FusedMlpArgs& a, hipStream_t s) {
    constexpr int LDS = fb_lds_bytes(D, H_STAGING);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static const int once = set_lds(blk_mlp1_kernel<D>, LDS);
    (void)once;
    const int grid = (int)((a.M + a.band_rows - 1) / a.band_rows) * a.nslice;
    constexpr long long KEY = 800000000000LL + D;                          // bench.py: 8 = fused norm2 + fc1 + GELU
    if (s3d_prof_skipped(KEY)) return 0;
    s3d_prof_begin(KEY, 2.0 * (double)a.M * a.hidden * D, s);
    hipLaunchKernelGGL((blk_mlp1_kernel<D>), dim3(grid), dim3(FB_THREADS), LDS, s, a);
    s3d_prof_end(s);
    S3D_CHECK_LAUNCH("blk_mlp1");
    return 0;
}

}  // namespace

bool s3d_fused_attn_ok(int Bb, int N, int D, int H) {
    return (D == 192 || D == 384) && H * 64 == D && N >= 1 && N <= 32 && Bb >= 1;
}
bool s3d_fused_mlp1_ok(long M, int D, int hidden) {
    // small-batch shapes only: from ~8 k rows on the 128-row GEMM tiles re-read the weights less often than 64-row bands do
    return (D == 192 || D == 384) && hidden % FB_WROWS == 0 && hidden / FB_WROWS >= D / 64 && M >= 1 && M <= 8192;
}

int s3d_launch_pack_weights(const bf16_t* src_hi, const bf16_t* src_lo, bf16_t* dst_hi, bf16_t* dst_lo, int rows, int K, hipStream_t s) {
    S3D_REQUIRE(src_hi && dst_hi && (src_lo == nullptr) == (dst_lo == nullptr), "pack_weights: null plane");
    S3D_REQUIRE(rows > 0 && K > 0 && rows % 16 == 0 && K % 32 == 0, "pack_weights: rows = %d must be a multiple of 16 and K = %d of 32", rows, K);
    const long pieces = (long)rows * (K / 8);
    const int grid = (int)((pieces + 255) / 256 < 2048 ? (pieces + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid), dim3(256), 0, s, src_hi, src_lo, dst_hi, dst_lo, rows, K);
    S3D_CHECK_LAUNCH("pack_weights");
    return 0;
}

int s3d_launch_fused_attn(const FusedAttnArgs& a, int D, hipStream_t s) {
    S3D_REQUIRE(s3d_fused_attn_ok(a.Bb, a.N, D, a.H), "fused attention block: unsupported shape Bb=%d N=%d D=%d H=%d", a.Bb, a.N, D, a.H);
    return D == 192 ? launch_attn<192>(a, s) : launch_attn<384>(a, s);
}
int s3d_launch_fused_mlp1(const FusedMlpArgs& a_in, int D, hipStream_t s) {
    FusedMlpArgs a = a_in;
    // Rows per band: 64 fill the MFMA tile; but when 64-row bands leave CUs without a workgroup, thinner bands on all 256 CUs shorten
    // every workgroup's row-proportional phases (loads, LayerNorm, GELU, stores).  cfg-2: 1664 rows x 8 slices = 208 workgroups of 64 rows
    // -> 256 of 52 (= two samples).
    a.band_rows = FB_ROWS;
    if (a.nslice > 0 && a.nslice <= 256) {
        const long full = (a.M + FB_ROWS - 1) / FB_ROWS, fit = 256 / a.nslice;
        if (full < fit) a.band_rows = (int)((a.M + fit - 1) / fit);
        if (a.band_rows < 16) a.band_rows = 16;
    }
    S3D_REQUIRE(s3d_fused_mlp1_ok(a.M, D, a.hidden) && a.nslice * FB_WROWS == a.hidden, "fused mlp block: unsupported shape M=%ld D=%d hidden=%d", a.M, D,
                a.hidden);
    return D == 192 ? launch_mlp1<192>(a, s) : launch_mlp1<384>(a, s);
}
