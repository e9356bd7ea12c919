// The whole MLP branch of a timm Block as ONE launch for D = 192 (deit_tiny: the point path's transformer, 16 - 33 k token rows):
//     x_out = x_mid + fc2(gelu(fc1(norm2(x_mid))))                      (timm Block.forward / Mlp.forward, 3DViT/model.py:318-320)
// instead of LayerNorm + fc1 GEMM (GELU epilogue) + fc2 GEMM (residual epilogue): 11.6 + 64.5 + 63.6 us per block at cfg-4, the hidden
// activations (101 MB per block as two planes) written and read back through HBM in between (profiles/r04_cfg4_kernel_stats.txt).
//
// Per workgroup a band of 4 x RPW token rows (RPW = 16 or 32 rows per wave); NO cross-workgroup reduction:
//   * norm2 of a wave's rows happens in registers, in MFMA B-operand layout (lane = row l & 15, k-group l >> 4: eight consecutive features
//     per k-step, row statistics by two cross-lane adds), and the split-bf16 planes of xn STAY in registers for the whole launch (96 VGPRs
//     at RPW = 32) -- LDS holds nothing but the weight stream.
//   * the hidden dimension is walked in chunks of 32 units.  fc1 is computed TRANSPOSED, h^T[unit][row] = W1[unit][:] . xn^T, so that a
//     lane ends up with hidden units of ONE row; the rows of W1 are fed to the MFMA in the order (i >> 2) * 8 + f * 4 + (i & 3) (fragment f,
//     MFMA row i), which makes a lane's 2 x 4 accumulator values eight CONSECUTIVE hidden units: exactly a B-operand k-group for
//     out^T[o][row] += W2[o][units] . h^T -- the GEMM -> GEMM seam needs no LDS round trip, no cross-lane move -- and one 16-byte store each
//     for the saved pre-activation / activation (the backward's gelu' and fc2 wgrad).
//   * W1 / W2 chunks (both planes, 48 KB per chunk) arrive through a three-stage LDS-DMA ring (global_load_lds_dwordx4, source-side slot
//     swizzle: 384-byte W1 rows alias like 128-byte ones -> dma_swz64 on the low slot bits; 64-byte W2 rows -> dma_swz32).
// Three MFMAs per product (hi hi + hi lo + lo hi): the forward's precision.
//
// MEASURED (profiles/r05_fused_mlp_d192.txt): 163 us per launch at cfg-4 (257 bands = two rounds of ~81 us on 256 CUs) against 11.4 + 64.5 +
// 63.6 us for the three launches it replaces -- the cfg-4 step gains 1.5 % (15.34 -> 15.12 ms), cfg-5 nothing.  An ablation says why it is
// not the 35 us the MFMA count promises: without MFMAs, weight stream and activation stores the launch still takes 70 % of its time -- a
// 16-row wave tile reads 48 KB of weight fragments from LDS per 72 MFMAs, and the GELU / split epilogue is ~300 wave64 VALU instructions per
// chunk (4 cycles each); with one or two waves per SIMD these phases run one after the other instead of overlapping.
#include "fused_mlp.h"
#include "dma_tile.h"
#include "gemm.h"

namespace {

constexpr int FM_D = 192, FM_H = 768, FM_CH = 32, FM_NCH = FM_H / FM_CH;        // model dim, hidden, units per chunk, chunks
constexpr int FM_W1P = FM_CH * FM_D * 2;                                        // bytes of one W1 chunk plane: 32 rows x 384 B = 12 KB
constexpr int FM_W2P = FM_D * FM_CH * 2;                                        // bytes of one W2 chunk plane: 192 rows x 64 B = 12 KB
constexpr int FM_STAGE = 2 * FM_W1P + 2 * FM_W2P;                               // 48 KB
constexpr int FM_NS = 3;
constexpr int FM_PIECES = FM_STAGE / 1024;                                      // 1 KB DMA pieces per stage: 48 = [W1 hi 12 | W1 lo 12 | W2 hi 12 | W2 lo 12]

struct FusedMlpFullArgs {
    const float* x; const float* gamma; const float* beta; float eps;
    const bf16_t *w1_hi, *w1_lo; const float* b1;
    const bf16_t *w2_hi, *w2_lo; const float* b2;
    float* x_out; bf16_t* xn_hi; bf16_t* xn_lo; float* mean; float* rstd;
    bf16_t* hpre; bf16_t* hact_hi;
    long M;
};

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
    U128 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) split_bf16x2(v[2 * e], v[2 * e + 1], h.w[e], l.w[e]);
    hi = h.v; lo = l.v;
}

// NW waves of RPW rows each; the first NDMA of them deal the weight stream's pieces among themselves (NW = 9, NDMA = 8: a ninth wave that
// only computes -- bands of 144 rows, so that cfg-4's 32 896 rows are 229 workgroups = ONE round on 256 CUs instead of 257 = two)
template <int RPW, int NW, int NDMA = NW>
__global__ __launch_bounds__(64 * NW) void blk_mlp_full_kernel(const FusedMlpFullArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RF = RPW / 16;                                        // 16-row fragments per wave
    constexpr int KS = FM_D / 32;                                       // k-steps of fc1: 6
    constexpr int OF = FM_D / 16;                                       // output fragments of fc2: 12
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    constexpr int FM_PPW = FM_PIECES / NDMA;                            // pieces per DMA wave: 12 (four waves) or 6 (eight)
    static_assert((NDMA == 4 || NDMA == 8) && NW >= NDMA, "four or eight waves carry the weight stream");
    const bool dma_wave = wave < NDMA;                                  // uniform per wave
    const long row0 = ((long)blockIdx.x * NW + wave) * RPW;
    const bool all_rows = row0 + RPW <= p.M;                            // uniform per wave: every row fragment issues its stores
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    float* sb1 = reinterpret_cast<float*>(smem + FM_NS * FM_STAGE);     // [768] fc1 bias

    // ---- weight stream: the stage's 48 pieces are dealt out to the waves in order; plane = piece / 12: W1 hi, W1 lo, W2 hi, W2 lo
    const bf16_t* gp[FM_PPW];
    long gstep[1];
    {
        const int wv = dma_wave ? wave : 0;                             // (a wave without pieces: valid pointers it never uses)
        const int plane = (wv * FM_PPW) / 12, p0 = (wv * FM_PPW) % 12;          // (a wave's pieces lie in one plane: 12 % FM_PPW == 0)
        if (plane < 2) {
            const bf16_t* base = plane == 0 ? p.w1_hi : p.w1_lo;
#pragma unroll
            for (int j = 0; j < FM_PPW; ++j) {
                const int t = (p0 + j) * 64 + lane;                     // 16-byte slot of the chunk plane: 32 rows x 24 slots
                const int r = t / 24, s = t % 24;                       // LDS row r = f * 16 + i holds hidden unit (i >> 2) * 8 + f * 4 + (i & 3)
                const int u = ((r & 15) >> 2) * 8 + (r >> 4) * 4 + (r & 3);
                const int src = (s & ~7) | ((s & 7) ^ dma_swz64(r));    // LDS slot (r, s) holds global 16-byte chunk `src` of that unit's row
                gp[j] = base + (long)u * FM_D + src * 8;
            }
            gstep[0] = (long)FM_CH * FM_D;                              // next chunk: 32 rows further
        } else {
            const bf16_t* base = plane == 2 ? p.w2_hi : p.w2_lo;
#pragma unroll
            for (int j = 0; j < FM_PPW; ++j) {
                const int r = (p0 + j) * 16 + (lane >> 2), s = lane & 3;        // 192 rows x 4 slots (32 units)
                gp[j] = base + (long)r * FM_H + ((s ^ dma_swz32(r)) << 3);
            }
            gstep[0] = FM_CH;                                           // next chunk: 32 columns further
        }
    }
    auto issue = [&](int c) {
        if (!dma_wave) return;
        const unsigned dst = lds0 + (unsigned)((c % FM_NS) * FM_STAGE + wave * FM_PPW * 1024);
#pragma unroll
        for (int j = 0; j < FM_PPW; ++j) glds16(gp[j] + (long)c * gstep[0], dst + j * 1024);
    };
#pragma unroll
    for (int u = 0; u < FM_NS - 1; ++u) issue(u);

    // ---- norm2 in registers (B-operand layout): lane = (row l15 of fragment rf, k-group g)
    bf16x8 xh[RF][KS], xl[RF][KS];
    for (int i = tid; i < FM_H; i += 64 * NW) sb1[i] = p.b1[i];
#pragma unroll
    for (int rf = 0; rf < RF; ++rf) {
        const long row = row0 + rf * 16 + l15;
        const long rr = row < p.M ? row : p.M - 1;
        float v[KS][8];
        float s1 = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float* src = p.x + rr * FM_D + ks * 32 + g * 8;
            const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[ks][e] = a[e]; v[ks][4 + e] = b[e]; s1 += a[e] + b[e]; }
        }
        s1 += __shfl_xor(s1, 16); s1 = half_sum(s1);
        const float mean = s1 * (1.0f / FM_D);
        float s2 = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[ks][e] - mean; s2 += d * d; }
        s2 += __shfl_xor(s2, 16); s2 = half_sum(s2);
        const float rstd = rsqrtf(s2 * (1.0f / FM_D) + p.eps);
        if (g == 0 && row < p.M) { p.mean[row] = mean; p.rstd[row] = rstd; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int col = ks * 32 + g * 8;
            const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + col), gb = *reinterpret_cast<const f32x4*>(p.gamma + col + 4);
            const f32x4 ba = *reinterpret_cast<const f32x4*>(p.beta + col), bb = *reinterpret_cast<const f32x4*>(p.beta + col + 4);
            float y[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = (v[ks][e] - mean) * rstd * ga[e] + ba[e];
                y[4 + e] = (v[ks][4 + e] - mean) * rstd * gb[e] + bb[e];
            }
            split8(y, xh[rf][ks], xl[rf][ks]);
            if (row < p.M) {
                U128 h; h.v = xh[rf][ks];
                *reinterpret_cast<u32x4*>(p.xn_hi + row * FM_D + col) = h.u;
                if (p.xn_lo) { U128 l; l.v = xl[rf][ks]; *reinterpret_cast<u32x4*>(p.xn_lo + row * FM_D + col) = l.u; }
            }
        }
    }

    f32x4 out[RF][OF];
#pragma unroll
    for (int rf = 0; rf < RF; ++rf)
#pragma unroll
        for (int of = 0; of < OF; ++of) out[rf][of] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < FM_NCH; ++c) {
        // stage c must have landed.  Younger than its pieces (vmcnt retires in issue order): the two 16-byte stores per row fragment of each
        // of the last NS - 1 chunks and the pieces of the NS - 2 stages behind it -- they may all stay in flight (waiting for them too
        // exposed a store acknowledgement + most of a DMA latency at every chunk: 182 us per launch at cfg-4)
        // A wave whose rows lie (partly) beyond M skips those stores altogether (execz branch), so for it only the DMA pieces are younger:
        // counting stores it never issued would let pieces of stage c stay in flight past the barrier (the tail band of cfg-4 /
        // cfg-5).  A smaller count is always safe.
        if (c + FM_NS - 1 <= FM_NCH) {
            if (all_rows) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((FM_NS - 2) * FM_PPW + (FM_NS - 1) * 2 * RF) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((FM_NS - 2) * FM_PPW) : "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (c + FM_NS - 1 < FM_NCH) issue(c + FM_NS - 1);
        const unsigned char* s1h = smem + (c % FM_NS) * FM_STAGE;
        const unsigned char *s1l = s1h + FM_W1P, *s2h = s1h + 2 * FM_W1P, *s2l = s2h + FM_W2P;
        // ---- fc1 (transposed): h^T[unit][row], units of fragment f in the order (i >> 2) * 8 + f * 4 + (i & 3)
        f32x4 hacc[RF][2];
#pragma unroll
        for (int rf = 0; rf < RF; ++rf) { hacc[rf][0] = f32x4{0.f, 0.f, 0.f, 0.f}; hacc[rf][1] = hacc[rf][0]; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int r = f * 16 + l15;                             // LDS row (the DMA stored the units in MFMA order)
                const int slot = ks * 4 + g;                            // 16-byte slot of the row: 8 consecutive k
                const int off = r * (FM_D * 2) + (((slot & ~7) | ((slot & 7) ^ dma_swz64(r))) << 4);
                const bf16x8 wh = *reinterpret_cast<const bf16x8*>(s1h + off), wl = *reinterpret_cast<const bf16x8*>(s1l + off);
#pragma unroll
                for (int rf = 0; rf < RF; ++rf) {
                    hacc[rf][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh[rf][ks], hacc[rf][f], 0, 0, 0);
                    hacc[rf][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[rf][ks], hacc[rf][f], 0, 0, 0);
                    hacc[rf][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[rf][ks], hacc[rf][f], 0, 0, 0);
                }
            }
        }
        // ---- bias + GELU; a lane now holds units c * 32 + g * 8 .. + 7 of row l15: the B operand of fc2, and one 16-byte store each
        bf16x8 hh[RF], hl[RF];
        {
            const f32x4 ba = *reinterpret_cast<const f32x4*>(sb1 + c * FM_CH + g * 8), bb = *reinterpret_cast<const f32x4*>(sb1 + c * FM_CH + g * 8 + 4);
#pragma unroll
            for (int rf = 0; rf < RF; ++rf) {
                float pre[8], act[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { pre[e] = hacc[rf][0][e] + ba[e]; pre[4 + e] = hacc[rf][1][e] + bb[e]; }
#pragma unroll
                for (int e = 0; e < 8; ++e) act[e] = gelu_erf(pre[e]);
                split8(act, hh[rf], hl[rf]);
                const long row = row0 + rf * 16 + l15;
                if (row < p.M) {
                    U128 q;
#pragma unroll
                    for (int e = 0; e < 4; ++e) q.w[e] = f2bf2(pre[2 * e], pre[2 * e + 1]);
                    *reinterpret_cast<u32x4*>(p.hpre + row * FM_H + c * FM_CH + g * 8) = q.u;
                    U128 a; a.v = hh[rf];
                    *reinterpret_cast<u32x4*>(p.hact_hi + row * FM_H + c * FM_CH + g * 8) = a.u;
                }
            }
        }
        // ---- fc2 (transposed): out^T[o][row] += W2[o][units of this chunk] . h^T
#pragma unroll
        for (int of = 0; of < OF; ++of) {
            const int o = of * 16 + l15;
            const int off = o * 64 + ((g ^ dma_swz32(o)) << 4);
            const bf16x8 wh = *reinterpret_cast<const bf16x8*>(s2h + off), wl = *reinterpret_cast<const bf16x8*>(s2l + off);
#pragma unroll
            for (int rf = 0; rf < RF; ++rf) {
                out[rf][of] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, hh[rf], out[rf][of], 0, 0, 0);
                out[rf][of] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, hl[rf], out[rf][of], 0, 0, 0);
                out[rf][of] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, hh[rf], out[rf][of], 0, 0, 0);
            }
        }
    }
    // ---- x_out = x_mid + out + b2: lane = row l15, outputs of * 16 + g * 4 .. + 3
#pragma unroll
    for (int rf = 0; rf < RF; ++rf) {
        const long row = row0 + rf * 16 + l15;
        if (row >= p.M) continue;
#pragma unroll
        for (int of = 0; of < OF; ++of) {
            const int o = of * 16 + g * 4;
            const f32x4 r = *reinterpret_cast<const f32x4*>(p.x + row * FM_D + o), b = *reinterpret_cast<const f32x4*>(p.b2 + o);
            *reinterpret_cast<f32x4*>(p.x_out + row * FM_D + o) = (out[rf][of] + b) + r;
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

// From 16 k rows on: cfg-4 (32 896 rows, 229 bands of nine waves) -5.4 % of the step against LayerNorm + fc1 + fc2 as three launches, cfg-5
// (16 416 rows, 129 bands of eight waves) -1.4 % (profiles/r05_fused_mlp_d192.txt).
bool s3d_fused_mlp_full_ok(long M, int D, int hidden) {
    static const int min_rows = s3d_tune_int("S3D_FUSED_MLP_MIN_ROWS");
    return D == FM_D && hidden == FM_H && M >= (min_rows > 0 ? min_rows : 16384);
}

int s3d_launch_fused_mlp_full(const FusedMlpArgs& a, const bf16_t* w2_hi, const bf16_t* w2_lo, const float* b2, float* x_out, int D, hipStream_t s) {
    S3D_REQUIRE(s3d_fused_mlp_full_ok(a.M, D, a.hidden), "fused MLP: D = 192, hidden = 768, >= 16384 rows (got D=%d hidden=%d M=%ld)", D, a.hidden, a.M);
    S3D_REQUIRE(a.x && a.gamma && a.beta && a.w_hi && a.w_lo && a.bias && w2_hi && w2_lo && b2 && x_out && a.xn_hi && a.mean && a.rstd && a.hpre && a.hact_hi,
                "fused MLP: null pointer");
    FusedMlpFullArgs f;
    f.x = a.x; f.gamma = a.gamma; f.beta = a.beta; f.eps = a.eps; f.w1_hi = a.w_hi; f.w1_lo = a.w_lo; f.b1 = a.bias;
    f.w2_hi = w2_hi; f.w2_lo = w2_lo; f.b2 = b2; f.x_out = x_out; f.xn_hi = a.xn_hi; f.xn_lo = a.xn_lo; f.mean = a.mean; f.rstd = a.rstd;
    f.hpre = a.hpre; f.hact_hi = a.hact_hi; f.M = a.M;
    constexpr int LDS = FM_NS * FM_STAGE + FM_H * 4;
    constexpr long long KEY = 1500000000000LL + 192;                    // bench.py: 15 = fused norm2 + fc1 + GELU + fc2 + residual
    if (s3d_prof_skipped(KEY)) return 0;
    // Eight waves of 16 rows (bands of 128 rows, two waves per SIMD).  Measured (profiles/r05_fused_mlp_d192.txt): four waves of 32 rows -- every
    // weight fragment read serving two row fragments, one wave per SIMD -- take 211 us per launch at cfg-4 against 163, four waves of 16 rows 238.
    s3d_prof_begin(KEY, 2.0 * 2.0 * (double)a.M * FM_D * FM_H, s);
    // ... and a ninth, compute-only wave (bands of 144 rows) where that saves a round of workgroups: a workgroup is paced by its own weight
    // stream and fragment reads, so the 257th band of cfg-4 cost as much as the 256 before it (145 us per launch, two rounds)
    const long wg8 = (a.M + 127) / 128, wg9 = (a.M + 143) / 144;
    const bool nine = (wg9 + 255) / 256 < (wg8 + 255) / 256;
    static const int nw_env = s3d_tune_int("S3D_FUSED_MLP_NW");           // tuning builds: 5 / 6 / 7 (four DMA waves), 8, 9
    const int nw = nw_env > 0 ? nw_env : (nine ? 9 : 8);
#define S3D_FM_LAUNCH(NW_, ND_)                                                                                                  \
    do {                                                                                                                         \
        static bool set = false;                                                                                                 \
        set_lds_once(blk_mlp_full_kernel<16, NW_, ND_>, LDS, set);                                                               \
        hipLaunchKernelGGL((blk_mlp_full_kernel<16, NW_, ND_>), dim3((unsigned)((a.M + 16 * NW_ - 1) / (16 * NW_))), dim3(64 * NW_), LDS, s, f); \
    } while (0)
    switch (nw) {
#ifdef S3D_EXPERIMENTAL_TILES
        case 5: S3D_FM_LAUNCH(5, 4); break;
        case 6: S3D_FM_LAUNCH(6, 4); break;
        case 7: S3D_FM_LAUNCH(7, 4); break;
#endif
        case 9: S3D_FM_LAUNCH(9, 8); break;
        default: S3D_FM_LAUNCH(8, 8); break;
    }
#undef S3D_FM_LAUNCH
    s3d_prof_end(s);
    S3D_CHECK_LAUNCH_V("blk_mlp_full", nw * 16 + 16);
    return 0;
}
