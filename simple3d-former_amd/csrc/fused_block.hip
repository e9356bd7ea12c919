// Fused launches of the timm Block forward for the small-batch voxel configurations (cfg-1 / cfg-2: M = B * 26 = 1664 token rows).
//
//   blk_attn_kernel   norm1 -> qkv head slice -> softmax(q k^T) v      one workgroup per (pair of samples, head)
//   blk_mlp1_kernel   norm2 -> fc1 hidden slice -> GELU                one workgroup per (band of <= 64 rows, 192-wide hidden slice)
//
// (timm==0.3.2 Block / Attention / Mlp forward as restated in oracle/timm_shim/timm/models/vision_transformer.py:41-78, invoked at
// models/vit_3d_2d_pretrain.py:466-469.)  Together with the attn.proj and mlp.fc2 GEMM launches a block forward is four launches
// instead of seven: the LayerNorm launches, the attention launch and the round trips of xn1 / qkv / xn2 through memory as split-bf16
// GEMM operands are gone.
//
// Both kernels are the same "skinny GEMM": 64 activation rows x 192 weight rows x K = D, eight waves (2 x 4) on 32 x 48 sub-tiles, where
//   * the A operand never exists in memory as a GEMM operand: every thread keeps 1/8 of one residual-stream row in registers (fp32), the
//     row statistics are an 8-lane DPP reduction, and each 64-wide k-slab is normalised, split into the two bf16 planes and written
//     into a double-buffered LDS tile just before the two k = 32 steps that consume it.  The LayerNorm is recomputed by every
//     workgroup that shares the rows (H = 6 resp. hidden / 192 = 8 times) -- the round-2 consumer-side fusion into the generic GEMM
//     redid it 18 - 24 times, once per tile column;
//   * the B operand (a 192-row slice of the row-major weight planes, 147 KB per plane at D = 384) streams through a ring of four 24 KB
//     LDS stages by LDS-DMA (global_load_lds_dwordx4), three stages in flight, one counted s_waitcnt + one barrier per k-step;
//     workgroups that share a weight slice are mapped to the same XCD, so the slice is fetched into one L2 and re-read there;
//   * split-bf16 product = three 16x16x32 MFMAs (hi*lo + lo*hi + hi*hi).
//
// What was measured on the way (round 3; profiles/r03_fused_*.txt, tools/fused_timeline_probe.py on the TL=1 build):
//   * this version: 17.6 us (attention half) / 19.3 us (MLP half) per launch against ~21.5 us each for the launches they replace;
//   * weights straight from the row-major planes into MFMA operand registers (no LDS, no barrier in the main loop): 20.4 / 22.1 us --
//     the 16 lanes of a quarter wave read 16 different weight rows, every wave-load touches 16 - 64 cache lines, the texture-address
//     path becomes the bound;
//   * the same on a second copy of the planes stored in MFMA fragment order (1 KB contiguous per wave-load), whole A operand resident
//     in LDS, four free-running waves: 16.2 / 18.8 us -- but keeping that copy current costs 46 us per step as a pack launch behind the
//     optimizer (80 us when the Adam kernel scatters the 16-byte pieces itself), more than the two launches gain over this version.
//     Its in-kernel timeline explains the plateau: with one workgroup per CU the phases run strictly one after the other --
//     first-touch latency + row loads + LayerNorm 13 k cycles, main loop 9.4 k (30 B/clk/CU of weights), GELU epilogue 6 k, stores 3 k;
//   * no fc2 / proj inside these launches: their reductions run across workgroups, and accumulating partial tiles into the residual
//     stream with fp32 atomics runs at 1.0 - 1.3 TB/s whatever the contention or access shape (tools/probes/atomic_resid_probe.hip,
//     profiles/r03_atomic_resid_probe.txt) -- 15 - 19 us for the 20 MB of partials of ONE mlp.fc2, more than the fc2 GEMM launch.
//     Every scope of global_atomic_add_f32 lowers to the same memory-side instruction on gfx950 (per-XCD L2s are not coherent).
#include "fused_block.h"

#include "attn_frag.h"
#include "dma_tile.h"
#include "gemm.h"

namespace {

constexpr int FB_THREADS = 512;          // 8 compute waves: 2 (rows) x 4 (weight rows)
#ifndef S3D_FB_PROD
#define S3D_FB_PROD 0
#endif
constexpr int FB_PROD = S3D_FB_PROD;     // weight-stream producer waves behind the compute waves (0: the compute waves issue the DMA themselves)
constexpr int FB_LAUNCH = FB_THREADS + 64 * FB_PROD;
constexpr int FB_ROWS = 64;              // activation rows per workgroup
constexpr int FB_WROWS = 192;            // weight rows (output columns) per workgroup
#ifndef S3D_FB_BK
#define S3D_FB_BK 32
#endif
constexpr int FB_BK = S3D_FB_BK;         // k per weight stage: 32 (64-byte row segments) or 64 (128-byte ones: whole cache lines)
#ifndef S3D_FB_NS
#define S3D_FB_NS 4
#endif
constexpr int FB_NS = S3D_FB_NS;         // weight stages in the ring (make EXP=1 EXTRA=-DS3D_FB_NS=5: measured, section 6)
constexpr int FB_ROWB = FB_BK * 2;                   // bytes per weight row and stage
constexpr int FB_PLANE = FB_WROWS * FB_ROWB;         // 12 288 B: one plane of a stage (k = 32)
constexpr int FB_STAGE = 2 * FB_PLANE;               // hi + lo
constexpr int FB_APLANE = FB_ROWS * 128;             // 8 192 B: one plane of an A slab ([64 rows][64 k], 128-byte rows)
constexpr int FB_ABUF = 2 * FB_APLANE;

constexpr int fb_lds_bytes(int D) { return FB_NS * FB_STAGE + 2 * FB_ABUF + 2 * D * 4; }

// in-kernel timeline (make TL=1 only; tools/fused_timeline_probe.py): thread 0 of every workgroup stamps s_memtime at the phase
// boundaries, s_memrealtime (100 MHz, chip-wide) at entry / exit.  Rows [0, 256) = blk_attn, [256, 512) = blk_mlp1, [512, 768) = blk_attn_bwd.
#ifdef S3D_TIMELINE
__device__ unsigned long long* g_fb_tl = nullptr;      // [768][FB_TL_SLOTS]
constexpr int FB_TL_SLOTS = 32;
#define FB_TL(w, i) do { if (g_fb_tl && threadIdx.x == 0) g_fb_tl[(long)(w) * FB_TL_SLOTS + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define FB_TL_REAL(w, i) do { if (g_fb_tl && threadIdx.x == 0) g_fb_tl[(long)(w) * FB_TL_SLOTS + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FB_TL(w, i) do {} while (0)
#define FB_TL_REAL(w, i) do {} while (0)
#endif

// sum over the 8 consecutive lanes that share a row: two quad butterflies + row_half_mirror (lane i <-> 7 - i within 8)
__device__ __forceinline__ float oct_sum(float v) {
    int x = __float_as_int(v);
#define S3D_OCT_STEP(ctrl) x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(x, x, ctrl, 0xf, 0xf, false)));
    S3D_OCT_STEP(0xB1)    // quad_perm:[1,0,3,2]
    S3D_OCT_STEP(0x4E)    // quad_perm:[2,3,0,1]
    S3D_OCT_STEP(0x141)   // row_half_mirror
#undef S3D_OCT_STEP
    return __int_as_float(x);
}

// LayerNorm + split + streamed-weight GEMM of a 64 x 192 output tile; leaves the accumulators of this wave's 32 x 48 sub-tile in
// `acc` (swapped operand roles: lane l owns row 32 (wave >> 2) + 16 i + (l & 15), columns 48 (wave & 3) + 16 j + 4 (l >> 4) .. + 3) and
// the row statistics of the thread's row (tid >> 3) in mean / rstd.  On return every DMA piece has landed and been consumed, but
// waves may still be reading the LAST stage / A slab: callers put a barrier in front of any LDS reuse.
//   xrow      this thread's share of its residual row: columns 8 * c8 + 64 * kp + 0..7 (already clamped to a valid row)
//   w_hi/lo   weight planes [.][D]; wrow0..2 = first weight row of the three 64-row blocks of this workgroup's slice
//   xn_hi/lo  where this thread stores its 8 normalised values of k-slab kp_store (nullptr: not this thread's job)
// (plain scalars, not a parameter struct: a select between two struct members becomes a select between their ADDRESSES and drags
// the struct into scratch memory -- the round-1 lesson of common.h, met again here)
template <int D>
__device__ __forceinline__ bool ln_gemm_64x192(const float* xrow, const float* gamma, const float* beta, const float eps, const bf16_t* w_hi,
                                               const bf16_t* w_lo, const long wrow0, const long wrow1, const long wrow2, bf16_t* xn_hi,
                                               bf16_t* xn_lo, const int kp_store, unsigned char* smem, f32x4 (&acc)[2][3], float& mean,
                                               float& rstd, const int tlw = 0) {
    constexpr int KP = D / 64, KT = D / FB_BK, NS = FB_NS;
    static_assert(D % 64 == 0 && KT >= NS - 1, "model dimension");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int row = tid >> 3, c8 = tid & 7;
    unsigned char* abuf = smem + NS * FB_STAGE;
    float* gb = reinterpret_cast<float*>(abuf + 2 * FB_ABUF);              // gamma [D] | beta [D]
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);

    // ---- weight stream: this lane's source of each of its wave's pieces (k-step 0).  FB_PROD > 0: only the producer waves (wave >= 8)
    //      stream; a wave that issues global_load_lds sits in the issue stage until the CU's memory pipeline has taken every piece
    //      (measured: ~1.1 k cycles per k-step for the 24 pieces of a stage, tools/fused_timeline_probe.py), and compute waves that do
    //      it themselves add that to every k-step.  Producer waves wait there instead while the compute waves read fragments and
    //      feed the matrix pipe; they hand a landed stage over at the k-step barrier (only the issuing wave can count its vmcnt).
    constexpr int NSTREAM = FB_PROD > 0 ? FB_PROD : FB_THREADS / 64;       // waves that issue the stream
    constexpr int PPW = FB_STAGE / 1024 / NSTREAM;                         // DMA pieces per streaming wave and stage
    static_assert((FB_STAGE / 1024) % NSTREAM == 0, "every streaming wave issues the same number of 1 KB pieces per stage");
    const int sw = FB_PROD > 0 ? wave - FB_THREADS / 64 : wave;            // index among the streaming waves (negative: compute wave)
    const bf16_t* gp[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        constexpr int RPP = 1024 / FB_ROWB, CPRW = FB_ROWB / 16, PP = FB_WROWS / RPP;   // rows per 1 KB piece, chunks per row, pieces per plane
        const int piece = max(sw, 0) * PPW + j;                            // [hi plane: PP pieces of RPP rows][lo plane]
        const int plane = piece / PP, rb = piece % PP;
        const int r = rb * RPP + lane / CPRW, c = lane % CPRW;
        const int rq = r >> 6;
        const long wrow = (rq == 0 ? wrow0 : rq == 1 ? wrow1 : wrow2) + (r & 63);
        gp[j] = (plane ? w_lo : w_hi) + wrow * D + ((c ^ (FB_BK == 64 ? dma_swz64(r) : dma_swz32(r))) << 3);
    }
    auto issue = [&](int s) {
        const unsigned dst = lds0 + (unsigned)((s % NS) * FB_STAGE + max(sw, 0) * PPW * 1024);
#pragma unroll
        for (int j = 0; j < PPW; ++j) glds16(gp[j] + s * FB_BK, dst + j * 1024);
    };
    if constexpr (FB_PROD > 0) {
        if (sw >= 0) {                                                     // ---- producer wave: stream, hand over, leave
            __syncthreads();                                               // (1) the compute waves' row loads are in the queue ahead of the stream
#pragma unroll
            for (int u = 0; u < NS - 1; ++u) issue(u);
            __syncthreads();                                               // (2) gamma / beta
#pragma unroll
            for (int st = 0; st < KT; ++st) {
                // stage st has landed once only this wave's pieces of the (at most NS - 2) younger stages are outstanding
                if (st + NS - 2 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PPW) : "memory");
                else if (NS > 3 && st + NS - 3 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS > 3 ? NS - 3 : 0) * PPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                                           // (3 + st) stage st handed over; the compute waves are done with st - 1
                if (st + NS - 1 < KT) issue(st + NS - 1);
            }
            return true;                                                   // barriers after this count the surviving waves only
        }
    } else {
        FB_TL(tlw, 2);
#pragma unroll
        for (int u = 0; u < NS - 1; ++u) issue(u);
        FB_TL(tlw, 3);                                                     // ring prologue issued
    }

    // ---- this thread's share of its row, gamma / beta -> LDS
    float xr[KP][8];
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(xrow + 64 * kp), b = *reinterpret_cast<const f32x4*>(xrow + 64 * kp + 4);
        xr[kp][0] = a[0]; xr[kp][1] = a[1]; xr[kp][2] = a[2]; xr[kp][3] = a[3];
        xr[kp][4] = b[0]; xr[kp][5] = b[1]; xr[kp][6] = b[2]; xr[kp][7] = b[3];
    }
    if constexpr (FB_PROD > 0) {
        // the row loads must be IN the queue before the producers start: pin them in front of the barrier (plain loads could sink past it)
        asm volatile("" ::: "memory");
        __syncthreads();                                                   // (1)
        FB_TL(tlw, 3);
    }
    for (int i = tid; i < 2 * D / 4; i += FB_THREADS)
        reinterpret_cast<f32x4*>(gb)[i] = *reinterpret_cast<const f32x4*>((i < D / 4 ? gamma : beta - D) + 4 * i);
    {
        float s = 0.f;
#pragma unroll
        for (int kp = 0; kp < KP; ++kp)
#pragma unroll
            for (int i = 0; i < 8; ++i) s += xr[kp][i];
        constexpr float inv_d = 1.0f / (float)D;
        mean = oct_sum(s) * inv_d;
        float q = 0.f;
#pragma unroll
        for (int kp = 0; kp < KP; ++kp)
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = xr[kp][i] - mean; q += d * d; }
        rstd = rsqrtf(oct_sum(q) * inv_d + eps);
    }
    FB_TL(tlw, 4);                                                         // rows arrived, statistics done

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                                       // gamma / beta are in LDS
    FB_TL(tlw, 5);

#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
        // ---- A slab kp: normalise, split, park in LDS buffer kp & 1 (its previous user, slab kp - 2, was retired two barriers ago)
        {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gb + 64 * kp + 8 * c8), g1 = *reinterpret_cast<const f32x4*>(gb + 64 * kp + 8 * c8 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(gb + D + 64 * kp + 8 * c8), b1 = *reinterpret_cast<const f32x4*>(gb + D + 64 * kp + 8 * c8 + 4);
            float y[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                y[i] = (xr[kp][i] - mean) * rstd * g0[i] + b0[i];
                y[4 + i] = (xr[kp][4 + i] - mean) * rstd * g1[i] + b1[i];
            }
            u32x4 hi, lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) { uint32_t h2, l2; split_bf16x2(y[2 * i], y[2 * i + 1], h2, l2); hi[i] = h2; lo[i] = l2; }
            unsigned char* at = abuf + (kp & 1) * FB_ABUF + row * 128 + ((c8 ^ dma_swz64(row)) << 4);
            *reinterpret_cast<u32x4*>(at) = hi;
            *reinterpret_cast<u32x4*>(at + FB_APLANE) = lo;
            if (kp == kp_store && xn_hi != nullptr) {                      // block-uniform on kp_store, per-thread on the pointer
                __builtin_nontemporal_store(hi, reinterpret_cast<u32x4*>(xn_hi));     // read next by the backward (wgrad operand)
                *reinterpret_cast<u32x4*>(xn_lo) = lo;
            }
        }
        constexpr int SPS = 64 / FB_BK;                                    // stages per 64-wide A slab: 2 (k = 32) or 1 (k = 64)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int s = SPS == 2 ? 2 * kp + half : kp;
            if (SPS == 2 || half == 0) {
                // stage s has landed once at most the NS - 2 younger stages' pieces of this wave are outstanding (loads retire in order;
                // stores in between only make the wait longer)
                if constexpr (FB_PROD == 0) {
                    if (s + NS - 1 <= KT && NS > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PPW) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();                                           // everyone's pieces + the A slab; stage s - 1 is free
                FB_TL(tlw, 6 + s);                                         // k-step s may start (6 .. 17 at D = 384)
                if constexpr (FB_PROD == 0) { if (s + NS - 1 < KT) issue(s + NS - 1); }
                if (s == 4) FB_TL(tlw, 26);                                // (timeline) ring refill of k-step 4 issued
            }
            const unsigned char* sW = smem + (s % NS) * FB_STAGE;
            const unsigned char* sA = abuf + (kp & 1) * FB_ABUF;
            const int kcA = half * 4 + (lane >> 4);
            bf16x8 a_hi[2], a_lo[2], b_hi[3], b_lo[3];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wm * 32 + i * 16 + (lane & 15);
                const unsigned char* q = sA + r * 128 + ((kcA ^ dma_swz64(r)) << 4);
                a_hi[i] = *reinterpret_cast<const bf16x8*>(q);
                a_lo[i] = *reinterpret_cast<const bf16x8*>(q + FB_APLANE);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int r = wn * 48 + j * 16 + (lane & 15);
                const unsigned char* q = FB_BK == 64 ? sW + r * 128 + ((kcA ^ dma_swz64(r)) << 4)
                                                     : sW + r * 64 + (((lane >> 4) ^ dma_swz32(r)) << 4);
                b_hi[j] = *reinterpret_cast<const bf16x8*>(q);
                b_lo[j] = *reinterpret_cast<const bf16x8*>(q + FB_PLANE);
            }
#ifdef S3D_TIMELINE
            if (s == 4) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); FB_TL(tlw, 27); }      // fragments of k-step 4 arrived
#endif
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_hi[j], a_lo[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_lo[j], a_hi[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_hi[j], a_hi[i], acc[i][j], 0, 0, 0);
            if (s == 4) FB_TL(tlw, 28);                                    // (timeline) MFMAs of k-step 4 issued
        }
    }
    FB_TL(tlw, 20);                                                        // last MFMAs issued
    return false;
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

template <int D>
__global__ __launch_bounds__(FB_LAUNCH) void blk_attn_kernel(const FusedAttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    FB_TL_REAL(blockIdx.x, 0); FB_TL(blockIdx.x, 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int npair = (p.Bb + 1) >> 1;
    const int item = xcd_item(blockIdx.x, npair * p.H);
    const int h = item / npair, pr = item % npair;                         // head-major: an XCD sees one or two heads' weights
    const int row = tid >> 3, c8 = tid & 7;
    const int sidx = row >> 5, tok = row & 31;
    const int b = 2 * pr + sidx;
    const bool row_ok = b < p.Bb && tok < p.N;
    const long grow = (long)min(b, p.Bb - 1) * p.N + min(tok, p.N - 1);   // clamped duplicate rows are computed and never stored

    f32x4 acc[2][3];
    float mean, rstd;
    const long xo = row_ok ? grow * D + 64 * h + 8 * c8 : -1;              // this head's 64 columns of xn1
    if (ln_gemm_64x192<D>(p.x + grow * D + 8 * c8, p.gamma, p.beta, p.eps, p.w_hi, p.w_lo, 64 * h, D + 64 * h, 2 * D + 64 * h,
                          xo >= 0 ? p.xn_hi + xo : nullptr, xo >= 0 ? p.xn_lo + xo : nullptr, h, smem, acc, mean, rstd, blockIdx.x))
        return;                                                            // weight-stream producer wave: done
    if (h == 0 && c8 == 0 && row_ok) { p.mean[grow] = mean; p.rstd[grow] = rstd; }

    // ---- q | k | v (+ bias) -> LDS tiles [sample][q,k,v][hi,lo][32 tokens][QKV_PITCH] in the (now idle) weight ring
    constexpr int TILE = QKV_TILE;
    static_assert(12 * TILE <= FB_NS * FB_STAGE, "staged q / k / v tiles fit the weight ring");
    const int wm = wave >> 2, wn = wave & 3;
    __syncthreads();                                                       // every wave is done with the last stage
    FB_TL(blockIdx.x, 21);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int n = wn * 48 + j * 16 + (lane >> 4) * 4;                  // 4 consecutive columns of q | k | v
        const int which = n >> 6, d = n & 63;
        const f32x4 bq = *reinterpret_cast<const f32x4*>(p.bias + which * D + 64 * h + d);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = i * 16 + (lane & 15);                            // token; the wave's 32 rows are sample wm
            u32x2 hi, lo;
            uint32_t h2, l2;
            split_bf16x2(acc[i][j][0] + bq[0], acc[i][j][1] + bq[1], h2, l2); hi[0] = h2; lo[0] = l2;
            split_bf16x2(acc[i][j][2] + bq[2], acc[i][j][3] + bq[3], h2, l2); hi[1] = h2; lo[1] = l2;
            unsigned char* q = smem + ((wm * 3 + which) * 2) * TILE + (t * QKV_PITCH + d) * 2;
            *reinterpret_cast<u32x2*>(q) = hi;
            *reinterpret_cast<u32x2*>(q + TILE) = lo;
        }
    }
    __syncthreads();
    // ---- q | k | v hi planes -> memory (operands of the attention backward and of the qkv wgrad), 16 bytes per thread and trip
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int e = tid + FB_THREADS * i;                                // 6 tiles x 32 tokens x 8 chunks
        const int tile = e >> 8, t = (e >> 3) & 31, ch = e & 7;
        const int sb = tile / 3, which = tile % 3, bb = 2 * pr + sb;
        if (bb < p.Bb && t < p.N) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + (tile * 2) * TILE + (t * QKV_PITCH + 8 * ch) * 2);
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p.qkv_hi + ((long)bb * p.N + t) * (3 * D) + which * D + 64 * h + 8 * ch));
        }
    }
    FB_TL(blockIdx.x, 22);                                                 // q | k | v staged and their stores issued
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
#ifdef S3D_TIMELINE
    FB_TL(blockIdx.x, 23);                                                 // attention done, stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FB_TL(blockIdx.x, 24); FB_TL_REAL(blockIdx.x, 25);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// norm2 -> fc1 slice -> GELU for a band of band_rows (<= 64) token rows and 192 hidden units.  Writes xn2 planes (slice s its 64
// columns 64 s ..), mean2 / rstd2 (slice 0), the bf16 pre-activation and the split planes of gelu(pre) -- the A operand of mlp.fc2.
constexpr int H_PITCH = 200;             // bf16 elements per staged output row (400 B: 16-byte aligned, rows 4 banks apart)
constexpr int H_TILE = FB_ROWS * H_PITCH * 2;                              // bytes per staged [64][192] bf16 array

template <int D>
__global__ __launch_bounds__(FB_LAUNCH) void blk_mlp1_kernel(const FusedMlpArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    FB_TL_REAL(256 + blockIdx.x, 0); FB_TL(256 + blockIdx.x, 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int RB = p.band_rows;
    const int nband = (int)((p.M + RB - 1) / RB);
    const int item = xcd_item(blockIdx.x, nband * p.nslice);
    const int js = item / nband, band = item % nband;                      // slice-major: the bands of a slice share an XCD
    const int row = tid >> 3, c8 = tid & 7;
    const long m0 = (long)band * RB;
    const bool row_ok = row < RB && m0 + row < p.M;
    const long grow = min(m0 + min(row, RB - 1), p.M - 1);

    f32x4 acc[2][3];
    float mean, rstd;
    const long xo = (row_ok && js < D / 64) ? grow * D + 64 * js + 8 * c8 : -1;   // slices 0 .. D / 64 - 1 store one 64-column slab of xn2 each
    const long w0 = (long)FB_WROWS * js;
    if (ln_gemm_64x192<D>(p.x + grow * D + 8 * c8, p.gamma, p.beta, p.eps, p.w_hi, p.w_lo, w0, w0 + 64, w0 + 128,
                          xo >= 0 ? p.xn_hi + xo : nullptr, xo >= 0 ? p.xn_lo + xo : nullptr, js, smem, acc, mean, rstd, 256 + blockIdx.x))
        return;                                                            // weight-stream producer wave: done
    if (js == 0 && c8 == 0 && row_ok) { p.mean[grow] = mean; p.rstd[grow] = rstd; }

    // ---- epilogue: pre = acc + b1 -> bf16; gelu(pre) -> split planes; staged through the idle weight ring for 16-byte row stores
    constexpr int TILE = H_TILE;
    static_assert(3 * TILE <= FB_NS * FB_STAGE, "staged outputs fit the weight ring");
    const int wm = wave >> 2, wn = wave & 3;
    __syncthreads();
    FB_TL(256 + blockIdx.x, 21);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int n = wn * 48 + j * 16 + (lane >> 4) * 4;
        const f32x4 bq = *reinterpret_cast<const f32x4*>(p.bias + FB_WROWS * js + n);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = wm * 32 + i * 16 + (lane & 15);
            const float p0 = acc[i][j][0] + bq[0], p1 = acc[i][j][1] + bq[1], p2 = acc[i][j][2] + bq[2], p3 = acc[i][j][3] + bq[3];
            u32x2 pre, hi, lo;
            uint32_t a, b2;
            pre[0] = f2bf2(p0, p1); pre[1] = f2bf2(p2, p3);
            split_bf16x2(gelu_erf(p0), gelu_erf(p1), a, b2); hi[0] = a; lo[0] = b2;
            split_bf16x2(gelu_erf(p2), gelu_erf(p3), a, b2); hi[1] = a; lo[1] = b2;
            unsigned char* q = smem + (r * H_PITCH + n) * 2;
            *reinterpret_cast<u32x2*>(q) = pre;
            *reinterpret_cast<u32x2*>(q + TILE) = hi;
            *reinterpret_cast<u32x2*>(q + 2 * TILE) = lo;
        }
    }
    __syncthreads();
    FB_TL(256 + blockIdx.x, 22);                                           // GELU epilogue staged
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int e = tid + FB_THREADS * i;                                // 3 arrays x 64 rows x 24 chunks of 16 bytes
        const int arr = e / (FB_ROWS * 24), rc = e % (FB_ROWS * 24), r = rc / 24, ch = rc % 24;
        if (r < RB && m0 + r < p.M) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + arr * TILE + (r * H_PITCH + 8 * ch) * 2);
            bf16_t* const out = arr == 0 ? p.hpre : arr == 1 ? p.hact_hi : p.hact_lo;
            u32x4* const dst = reinterpret_cast<u32x4*>(out + (m0 + r) * p.hidden + FB_WROWS * js + 8 * ch);
            if (arr == 0) __builtin_nontemporal_store(v, dst); else *dst = v;      // hpre: read next by the backward
        }
    }
#ifdef S3D_TIMELINE
    FB_TL(256 + blockIdx.x, 23);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FB_TL(256 + blockIdx.x, 24); FB_TL_REAL(256 + blockIdx.x, 25);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward of the attention half for TWO samples and one head (hd = 64, N <= 32 tokens): d(att) = d(x_mid) @ Wproj restricted to this
// head's 64 input columns (attn.proj dgrad), then the attention backward on it -- dQ by one wave, dK / dV by another, per sample --
// with d(att) handed over in LDS.  (timm Attention backward; unfused: s3d_gemm(0, 1, BF16_BIAS) + attn_bwd_small_kernel.)
//   * the weight slice Wproj[:, 64 h .. 64 h + 63] (D rows of 128 bytes, 48 KB at D = 384) is staged once per workgroup; the rows of each
//     32-row k-tile are stored in MFMA k-slot order (slot_key), so that the transposed LDS read delivers the B-side fragment whose k-slots
//     are 8 CONSECUTIVE output features -- and the A-side fragment (d(x_mid) of this lane's token) is a plain 16-byte row read from memory;
//   * 256 threads: wave (sample s, half) computes d(att)^T[32 features of its half][32 tokens] with 32x32x16 MFMAs, stores it as the bf16
//     [32 tokens][64] tile the attention backward stages anyway, then runs phase A (half 0: lane = query, dQ) or phase B (half 1: lane = key,
//     dK and dV) of attn_bwd_small_kernel; P is recomputed from the saved log-sum-exp, delta = rowsum(dO * O).
constexpr int BW_WPITCH = 72;            // bf16 elements per staged weight / tile row (144 B: 16-byte aligned, rows 4 banks apart)
constexpr int BW_TILE = 32 * BW_WPITCH;  // elements of a staged [32 tokens][64] tile
constexpr int BW_SAMPLE = 4 * BW_TILE + 128;      // Q | K | V | dO tiles + 64 floats (delta, lse)
constexpr int BW_THREADS = 512;          // eight waves stage; waves 0..3 compute (sample, half)
constexpr int bw_apitch(int D) { return D + 8; }
constexpr int BW_STAT = 2 * 3 * 64;      // floats: this head's slices of the row-statistics vectors u | c (q, k, v columns)
constexpr int bw_lds_bytes(int D) { return (D * BW_WPITCH + 64 * bw_apitch(D) + 2 * BW_SAMPLE) * 2 + BW_STAT * 4; }

// row statistics of four gradient values g (bf16 pairs as stored) against the saved pre-activation z (LDS, 4 bf16) and the u / c slices
__device__ __forceinline__ void bw_row_stats(const u32x2 g, const bf16_t* z, const float* u, const float* c, float& s1, float& s2) {
    const u32x2 zz = *reinterpret_cast<const u32x2*>(z);
    const f32x4 u4 = *reinterpret_cast<const f32x4*>(u), c4 = *reinterpret_cast<const f32x4*>(c);
    const float gv[4] = {__uint_as_float(g[0] << 16), __uint_as_float(g[0] & 0xffff0000u), __uint_as_float(g[1] << 16), __uint_as_float(g[1] & 0xffff0000u)};
    const float zv[4] = {__uint_as_float(zz[0] << 16), __uint_as_float(zz[0] & 0xffff0000u), __uint_as_float(zz[1] << 16), __uint_as_float(zz[1] & 0xffff0000u)};
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1 = fmaf(gv[e], u4[e], s1); s2 = fmaf(gv[e], zv[e] - c4[e], s2); }
}

template <int D>
__global__ __launch_bounds__(BW_THREADS) void blk_attn_bwd_kernel(const FusedAttnBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int HD = 64, NS = HD / 16, KT = D / 32, AP = bw_apitch(D), WP = BW_WPITCH, CPR = D / 8;
    FB_TL_REAL(512 + blockIdx.x, 0); FB_TL(512 + blockIdx.x, 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h2 = lane >> 5, l31 = lane & 31;
    const int npair = (p.Bb + 1) >> 1;
    const int item = xcd_item(blockIdx.x, npair * p.H);
    const int h = item / npair, pr = item % npair;                         // head-major: an XCD sees one or two heads' weight slices
    bf16_t* sW = reinterpret_cast<bf16_t*>(smem);                          // [D][WP], rows of a k-tile in k-slot order
    bf16_t* sA = sW + D * WP;                                              // d(x_mid) rows of the pair: [64][AP]
    bf16_t* sT = sA + 64 * AP;                                             // per sample: Q | K | V | dO tiles, then delta / lse
    float* sU = reinterpret_cast<float*>(sT + 2 * BW_SAMPLE);              // [3][64] u, then [3][64] c of this head (row statistics)
    const bool stats = p.st_s1 != nullptr;                                 // uniform

    // ---- stage everything with coalesced 16-byte loads (row fragments gathered straight from memory touch 32 rows per load
    //      instruction: measured, the texture-address path then costs more than the arithmetic of the whole kernel)
#pragma unroll
    for (int i = 0; i < D * 8 / BW_THREADS; ++i) {
        const int c = tid + BW_THREADS * i, o = c >> 3, ch = c & 7;
        const int ol5 = o & 31, s2 = ol5 >> 4, hh = (ol5 >> 3) & 1, j = ol5 & 7;
        const int r = (o & ~31) + slot_key(s2, hh, j);
        *reinterpret_cast<u32x4*>(sW + r * WP + 8 * ch) = *reinterpret_cast<const u32x4*>(p.w_hi + (long)o * D + 64 * h + 8 * ch);
    }
#pragma unroll
    for (int i = 0; i < 64 * CPR / BW_THREADS; ++i) {
        const int c = tid + BW_THREADS * i, row = c / CPR, ch = c % CPR;
        const int bb = min(2 * pr + (row >> 5), p.Bb - 1), tt = min(row & 31, p.N - 1);
        *reinterpret_cast<u32x4*>(sA + row * AP + 8 * ch) = *reinterpret_cast<const u32x4*>(p.dxm + ((long)bb * p.N + tt) * p.lddxm + 8 * ch);
    }
#pragma unroll
    for (int i = 0; i < 2 * 3 * 256 / BW_THREADS; ++i) {
        const int c = tid + BW_THREADS * i, sm = c / 768, which = (c % 768) >> 8, r = (c & 255) >> 3, ch = c & 7;
        const int bb = min(2 * pr + sm, p.Bb - 1), tt = min(r, p.N - 1);
        *reinterpret_cast<u32x4*>(sT + sm * BW_SAMPLE + which * BW_TILE + r * WP + 8 * ch) =
            *reinterpret_cast<const u32x4*>(p.qkv_hi + ((long)bb * p.N + tt) * (3 * D) + which * D + 64 * h + 8 * ch);
    }
    if (stats && tid >= 128 && tid < 128 + BW_STAT) {
        const int c = tid - 128, which = c / 192, part = (c % 192) >> 6, f = c & 63;
        sU[c] = (which ? p.st_c : p.st_u)[part * D + 64 * h + f];
    }
    if (tid < 64) {
        const int sm = tid >> 5, bb = min(2 * pr + sm, p.Bb - 1), tt = min(tid & 31, p.N - 1);
        const long li = p.lse_packed ? ((long)(bb >> 1) * p.H + h) * (2 * p.N) + (bb & 1) * p.N + tt : ((long)bb * p.H + h) * p.N + tt;
        reinterpret_cast<float*>(sT + sm * BW_SAMPLE + 4 * BW_TILE)[32 + (tid & 31)] = p.lse[li];
    }
    FB_TL(512 + blockIdx.x, 2);                                            // this thread's staged pieces are in LDS
    __syncthreads();
    FB_TL(512 + blockIdx.x, 3);

    const int smp = (wave >> 1) & 1, half = wave & 1;
    const bool worker = wave < 4;
    const int b = min(2 * pr + smp, p.Bb - 1);
    const bool active = worker && 2 * pr + smp < p.Bb;
    const bool tok_ok = l31 < p.N;
    const long tokrow = (long)b * p.N + min(l31, p.N - 1);
    const bf16_t* ldsQ = sT + smp * BW_SAMPLE;
    const bf16_t* ldsK = ldsQ + BW_TILE;
    const bf16_t* ldsV = ldsK + BW_TILE;
    bf16_t* ldsDO = sT + smp * BW_SAMPLE + 3 * BW_TILE;
    float* ldsR = reinterpret_cast<float*>(ldsDO + BW_TILE);               // [0..31] delta, [32..63] lse

    // ---- d(att)^T[feature 32 half + acc_row][token l31] = sum_o Wproj[o][64 h + feature] * d(x_mid)[token][o]
    if (worker) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const bf16_t* arow = sA + (smp * 32 + l31) * AP + h2 * 8;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            bf16x8 wf[2];
            gather_frag_s2<HD, WP>(sW + kt * 32 * WP, h2, 32 * half + l31, wf);
            U128 a0, a1;
            a0.u = *reinterpret_cast<const u32x4*>(arow + 32 * kt);
            a1.u = *reinterpret_cast<const u32x4*>(arow + 32 * kt + 16);
            acc = MFMA32(wf[0], a0.v, acc);
            acc = MFMA32(wf[1], a1.v, acc);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {                                      // rows 8 c + 4 h2 + {0..3} of the accumulator: 4 consecutive features
            u32x2 v;
            v[0] = f2bf2(acc[4 * c], acc[4 * c + 1]); v[1] = f2bf2(acc[4 * c + 2], acc[4 * c + 3]);
            *reinterpret_cast<u32x2*>(ldsDO + l31 * WP + 32 * half + 8 * c + 4 * h2) = v;
        }
    }
    FB_TL(512 + blockIdx.x, 4);                                            // proj dgrad done
    __syncthreads();                                                       // both halves of d(att) of both samples are in LDS
    FB_TL(512 + blockIdx.x, 5);

    // ---- row fragments of this lane's token; scores and dP of both phases
    bf16x8 qf[NS], kf[NS], vf[NS], dof[NS];
    f32x16 sacc, dpacc;
    float lse_q = 0.f;
    if (worker) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int off = l31 * WP + 16 * s + 8 * h2;
            U128 t;
            t.u = *reinterpret_cast<const u32x4*>(ldsQ + off); qf[s] = t.v;
            t.u = *reinterpret_cast<const u32x4*>(ldsK + off); kf[s] = t.v;
            t.u = *reinterpret_cast<const u32x4*>(ldsV + off); vf[s] = t.v;
            t.u = *reinterpret_cast<const u32x4*>(ldsDO + off); dof[s] = t.v;
        }
        lse_q = ldsR[32 + l31];
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
        if (half == 0) {        // phase A: lane = query.  S^T = K . Q^T, dP^T = V . dO^T
#pragma unroll
            for (int s = 0; s < NS; ++s) { sacc = MFMA32(kf[s], qf[s], sacc); dpacc = MFMA32(vf[s], dof[s], dpacc); }
        } else {                // phase B: lane = key.    S = Q . K^T,     dP = dO . V^T
#pragma unroll
            for (int s = 0; s < NS; ++s) { sacc = MFMA32(qf[s], kf[s], sacc); dpacc = MFMA32(dof[s], vf[s], dpacc); }
        }
    }
    // delta[q] = sum_k P[q][k] dP[q][k]  (= rowsum(dO * O) in exact arithmetic; with the P this backward recomputes from the bf16 q, k it
    // makes the rows of dS sum to zero exactly), by the phase-A wave, handed to the phase-B wave through LDS
    float pq[16], delta = 0.f;
    if (worker && half == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pq[r] = acc_row(r, h2) < p.N ? fast_exp(sacc[r] * p.scale - lse_q) : 0.f;
            delta += pq[r] * dpacc[r];
        }
        delta = half_sum(delta);
        if (h2 == 0) ldsR[l31] = delta;
    }
    __syncthreads();
    FB_TL(512 + blockIdx.x, 6);                                            // delta handed over
    if (!worker) return;

    const long orow = tokrow * (3 * D) + h * HD;
    if (half == 0) {
        // ---- phase A: dQ^T = K^T . dS^T
        U128 dsf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 8 * s2 + j;
                dsf[s2].h[j] = f2bf(pq[r] * (dpacc[r] - delta) * p.scale);
            }
        f32x16 dq[2];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;
        bf16x8 k0f[2], k1f[2];
        gather_frag_2x2<HD, WP>(ldsK, l31, ldsK, 32 + l31, h2, k0f, k1f);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            dq[0] = MFMA32(k0f[s2], dsf[s2].v, dq[0]);
            dq[1] = MFMA32(k1f[s2], dsf[s2].v, dq[1]);
        }
        float st1 = 0.f, st2 = 0.f;
        if (active && tok_ok) {
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    u32x2 v;
                    v[0] = f2bf2(dq[d][4 * c], dq[d][4 * c + 1]); v[1] = f2bf2(dq[d][4 * c + 2], dq[d][4 * c + 3]);
                    const int f = d * 32 + 8 * c + 4 * h2;
                    *reinterpret_cast<u32x2*>(p.dqkv + orow + f) = v;
                    if (stats) bw_row_stats(v, ldsQ + l31 * WP + f, sU + f, sU + 192 + f, st1, st2);
                }
        }
        if (stats) {
            st1 = half_sum(st1); st2 = half_sum(st2);
            if (active && tok_ok && h2 == 0) { atomic_add_f32(p.st_s1 + tokrow, st1); atomic_add_f32(p.st_s2 + tokrow, st2); }
        }
    } else {
        // ---- phase B: dV^T = dO^T . P, dK^T = Q^T . dS
        float st1 = 0.f, st2 = 0.f;
        U128 pf[2], dsf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 8 * s2 + j, q = acc_row(r, h2);
                const bool ok = tok_ok && q < p.N;
                const int qc = min(q, p.N - 1);
                const float pr_ = ok ? fast_exp(sacc[r] * p.scale - ldsR[32 + qc]) : 0.f;
                pf[s2].h[j] = f2bf(pr_);
                dsf[s2].h[j] = f2bf(pr_ * (dpacc[r] - ldsR[qc]) * p.scale);
            }
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            f32x16 dk, dv;
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
            bf16x8 fo[2], fq[2];
            gather_frag_2x2<HD, WP>(ldsDO, d * 32 + l31, ldsQ, d * 32 + l31, h2, fo, fq);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                dv = MFMA32(fo[s2], pf[s2].v, dv);
                dk = MFMA32(fq[s2], dsf[s2].v, dk);
            }
            if (active && tok_ok) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    u32x2 a, v;
                    a[0] = f2bf2(dk[4 * c], dk[4 * c + 1]); a[1] = f2bf2(dk[4 * c + 2], dk[4 * c + 3]);
                    v[0] = f2bf2(dv[4 * c], dv[4 * c + 1]); v[1] = f2bf2(dv[4 * c + 2], dv[4 * c + 3]);
                    const int f = d * 32 + 8 * c + 4 * h2;
                    const long off = orow + f;
                    *reinterpret_cast<u32x2*>(p.dqkv + off + D) = a;
                    *reinterpret_cast<u32x2*>(p.dqkv + off + 2 * D) = v;
                    if (stats) {
                        bw_row_stats(a, ldsK + l31 * WP + f, sU + 64 + f, sU + 192 + 64 + f, st1, st2);
                        bw_row_stats(v, ldsV + l31 * WP + f, sU + 128 + f, sU + 192 + 128 + f, st1, st2);
                    }
                }
            }
        }
        if (stats) {
            st1 = half_sum(st1); st2 = half_sum(st2);
            if (active && tok_ok && h2 == 0) { atomic_add_f32(p.st_s1 + tokrow, st1); atomic_add_f32(p.st_s2 + tokrow, st2); }
        }
    }
#ifdef S3D_TIMELINE
    FB_TL(512 + blockIdx.x, 7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FB_TL(512 + blockIdx.x, 8); FB_TL_REAL(512 + blockIdx.x, 9);
#endif
}

#ifdef S3D_TIMELINE
}  // namespace
extern "C" int s3d_debug_fused_timeline_set(void* buf) {
    unsigned long long* b = reinterpret_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_fb_tl), &b, sizeof(b)) == hipSuccess ? 0 : 1;
}
namespace {
#endif

template <typename K>
int set_lds(K kern, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 0 : 1;
}

template <int D>
int launch_attn(const FusedAttnArgs& a, hipStream_t s) {
    constexpr int LDS = fb_lds_bytes(D);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static const int once = set_lds(blk_attn_kernel<D>, LDS);
    (void)once;
    const int grid = ((a.Bb + 1) / 2) * a.H;
    constexpr long long KEY = 700000000000LL + D;                          // bench.py: 7 = fused norm1 + qkv + attention
    if (s3d_prof_skipped(KEY)) return 0;
    const double M = (double)a.Bb * a.N;
    s3d_prof_begin(KEY, 2.0 * M * 3 * D * D + 4.0 * a.Bb * a.N * a.N * D, s);       // qkv GEMM + (q k^T, p v) of every head
    hipLaunchKernelGGL((blk_attn_kernel<D>), dim3(grid), dim3(FB_LAUNCH), LDS, s, a);
    s3d_prof_end(s);
    S3D_CHECK_LAUNCH_V("blk_attn", D);
    return 0;
}
template <int D>
int launch_mlp1(const FusedMlpArgs& a, hipStream_t s) {
    constexpr int LDS = fb_lds_bytes(D);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static const int once = set_lds(blk_mlp1_kernel<D>, LDS);
    (void)once;
    const int grid = (int)((a.M + a.band_rows - 1) / a.band_rows) * a.nslice;
    constexpr long long KEY = 800000000000LL + D;                          // bench.py: 8 = fused norm2 + fc1 + GELU
    if (s3d_prof_skipped(KEY)) return 0;
    s3d_prof_begin(KEY, 2.0 * (double)a.M * a.hidden * D, s);
    hipLaunchKernelGGL((blk_mlp1_kernel<D>), dim3(grid), dim3(FB_LAUNCH), LDS, s, a);
    s3d_prof_end(s);
    S3D_CHECK_LAUNCH_V("blk_mlp1", D);
    return 0;
}

}  // namespace

namespace {
template <int D>
int launch_attn_bwd(const FusedAttnBwdArgs& a, hipStream_t s) {
    constexpr int LDS = bw_lds_bytes(D);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static const int once = set_lds(blk_attn_bwd_kernel<D>, LDS);
    (void)once;
    const int grid = ((a.Bb + 1) / 2) * a.H;
    constexpr long long KEY = 900000000000LL + D;                          // bench.py: 9 = fused proj dgrad + attention backward
    if (s3d_prof_skipped(KEY)) return 0;
    const double M = (double)a.Bb * a.N;
    s3d_prof_begin(KEY, 2.0 * M * D * D + 10.0 * a.Bb * a.N * a.N * D, s);          // proj dgrad + (s, dp, dq, dk, dv) of every head
    hipLaunchKernelGGL((blk_attn_bwd_kernel<D>), dim3(grid), dim3(BW_THREADS), LDS, s, a);
    s3d_prof_end(s);
    S3D_CHECK_LAUNCH_V("blk_attn_bwd", D + (a.st_s1 ? 100000 : 0));
    return 0;
}
}  // namespace

bool s3d_fused_attn_bwd_ok(int Bb, int N, int D, int H) { return s3d_fused_attn_ok(Bb, N, D, H); }
int s3d_launch_fused_attn_bwd(const FusedAttnBwdArgs& a, int D, hipStream_t s) {
    S3D_REQUIRE(s3d_fused_attn_bwd_ok(a.Bb, a.N, D, a.H), "fused attention backward: unsupported shape Bb=%d N=%d D=%d H=%d", a.Bb, a.N, D, a.H);
    S3D_REQUIRE(a.lddxm % 8 == 0, "fused attention backward: d(x_mid) row pitch must be a multiple of 8");
    return D == 192 ? launch_attn_bwd<192>(a, s) : launch_attn_bwd<384>(a, s);
}

bool s3d_fused_attn_ok(int Bb, int N, int D, int H) {
    // small-batch shapes only, like the MLP half: with 1e4 - 1e5 short sequences (group_embed pass 1 on a deit_tiny / deit_small backbone)
    // every pair of sequences would re-stream its head's 192-row weight slice and recompute norm1 H times, where the 128-row GEMM tiles
    // re-read the weights 4x less often; only cfg-1 / cfg-2 sizes were measured in favour of the fused launch
    return (D == 192 || D == 384) && H * 64 == D && N >= 1 && N <= 32 && Bb >= 1 && (long)Bb * N <= 8192;
}
bool s3d_fused_mlp1_ok(long M, int D, int hidden) {
    // small-batch shapes only: from ~8 k rows on the 128-row GEMM tiles re-read the weights less often than 64-row bands do
    return (D == 192 || D == 384) && hidden % FB_WROWS == 0 && hidden / FB_WROWS >= D / 64 && M >= 1 && M <= 8192;
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
