// Multi-head attention for gfx950 (MI355X), flash-style, one wave per (batch, head, 32-row tile),
// v_mfma_f32_32x32x16_bf16.  Works for any token count N and head dims 64 / 192 / 256:
//   cfg-1/2  N=26  hd=64      cfg-3  N=15 / 197, hd=256 and the seq-first encoder layer (N = B*P*P, hd=192)
//   cfg-4/5  N=257 / 513, hd=64
//
// Layout trick ("swapped" product): the score tile is computed transposed, S^T[key][query] = K . Q^T, so that in
// the MFMA C/D layout (col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) every lane owns ONE query column:
// row max / row sum / online rescale are lane-local (16 registers + one cross-half shuffle).  The probabilities
// are then already in B-operand layout for O^T[d][query] = V^T . P^T, whose k-slots are the 8 keys a lane holds
// per 16-key step; the matching A operand (V^T) is gathered from an LDS-staged [32 keys][hd] tile.
//
// Forward runs in split-bf16 (hi + lo planes, three MFMAs per product) so logits stay within 1e-3 of the fp32
// reference; backward runs in plain bf16 on the hi planes.
#include "attention.h"
#include "adam_fill.h"
#include "attn_frag.h"
#include "dma_tile.h"

#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

namespace {

// Fragment loads are unconditional: callers clamp token rows to N-1 (finite data) instead of predicating the load --
// a `cond ? *p : 0` load makes hipcc branch and wait per load, serialising the latencies.  Out-of-range keys are
// masked in the score tile and out-of-range queries are never stored, so clamped duplicates are harmless.
__device__ __forceinline__ bf16x8 ld_frag(const bf16_t* p) {
    U128 u;
    u.u = *reinterpret_cast<const u32x4*>(p);
    return u.v;
}

// cooperative (one wave) copy of a [32 rows][HD] bf16 tile into this wave's LDS region (row pitch HD*2 bytes)
// block-diagonal mask of two packed sequences (S3dAttnArgs::seg): query and key must lie in the same segment.  A template
// flag, not a run-time test: the mask costs the few-microsecond cfg-2 kernels 0.7 % of the step when it is compiled in.
template <bool SEG>
__device__ __forceinline__ bool seg_ok(const AttnArgs& p, int q, int k) {
    if constexpr (!SEG) return true;
    return (q >= p.seg) == (k >= p.seg);
}

template <int HD, int PITCH = HD>
__device__ __forceinline__ void stage_tile(bf16_t* lds, const bf16_t* g, long ld, long row0_off, long st_ld, int t0,
                                           int N, int lane) {
    constexpr int CPR = HD / 8;                 // 16-byte chunks per row
#pragma unroll
    for (int i = 0; i < (32 * CPR + 63) / 64; ++i) {
        const int c = min(lane + 64 * i, 32 * CPR - 1);
        const int r = c / CPR, cc = c % CPR;
        const int t = min(t0 + r, N - 1);
        *reinterpret_cast<u32x4*>(lds + r * PITCH + cc * 8) =
            *reinterpret_cast<const u32x4*>(g + row0_off + (long)t * st_ld + cc * 8);
    }
}

// ------------------------------------------------------------------------------------------- forward
// DROP: attention-weight dropout compiled in (only the group encoder layer uses it; the timm blocks never do, and merely carrying
// the masked path changed the register allocation of the 26-token kernel: 5.7 -> 8.3 us per launch)
constexpr int fwd_pitch(int HD) { return HD + 8; }

template <int HD, bool SPLIT, bool SEG = false, bool DROP = false>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NS = HD / 16, NDB = (HD + 31) / 32, NPL = SPLIT ? 2 : 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h2 = lane >> 5, l31 = lane & 31;
    // tile rows are padded by 16 bytes: 16-byte row-fragment reads of 32 different rows and the transposed reads stay conflict-free
    constexpr int PITCH = fwd_pitch(HD);
    bf16_t* ldsV = reinterpret_cast<bf16_t*>(smem) + wave * (NPL * 32 * PITCH);

    const int QT = (p.N + 31) / 32;
    const long W = (long)p.Bb * p.H * QT;
    long item = (long)blockIdx.x * (blockDim.x >> 6) + wave;
    const bool active = item < W;
    if (!active) item = W - 1;
    const int qt = (int)(item % QT);
    const int bh = (int)(item / QT);
    const int h = bh % p.H, b = bh / p.H;
    const long st_ld = p.st * p.ld;
    const long base = (long)b * p.sb * p.ld + h * HD;           // + t*st_ld + which*D + d
    const int q0 = qt * 32, qrow = q0 + l31;
    const bool qok = qrow < p.N;
    const int qrow_c = min(qrow, p.N - 1);

    // Q and K row fragments pass through this wave's tile space (coalesced loads, then 16-byte LDS reads) instead of being gathered from
    // memory with every lane on its own row (32 rows per load instruction: the slowest thing the texture-address path does, DESIGN.md
    // section 6); the LDS operations of one wave stay in order, so the V tile can follow into the same space.
    bf16x8 qh[NS], ql[SPLIT ? NS : 1];
    const int fo = l31 * PITCH + h2 * 8;
    {
        stage_tile<HD, PITCH>(ldsV, p.qkv_hi, p.ld, base, st_ld, q0, p.N, lane);
        if constexpr (SPLIT) stage_tile<HD, PITCH>(ldsV + 32 * PITCH, p.qkv_lo, p.ld, base, st_ld, q0, p.N, lane);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            U128 t;
            t.u = *reinterpret_cast<const u32x4*>(ldsV + fo + 16 * s); qh[s] = t.v;
            if constexpr (SPLIT) { t.u = *reinterpret_cast<const u32x4*>(ldsV + 32 * PITCH + fo + 16 * s); ql[s] = t.v; }
        }
        __builtin_amdgcn_wave_barrier();
    }
    f32x16 o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_i = -INFINITY, l_i = 0.f;

    const int KT = (p.N + 31) / 32;
    for (int kt = 0; kt < KT; ++kt) {
        const int k0 = kt * 32;
        __syncthreads();                                          // previous tile's LDS reads are done
        stage_tile<HD, PITCH>(ldsV, p.qkv_hi, p.ld, base + p.D, st_ld, k0, p.N, lane);                       // K first ...
        if constexpr (SPLIT) stage_tile<HD, PITCH>(ldsV + 32 * PITCH, p.qkv_lo, p.ld, base + p.D, st_ld, k0, p.N, lane);
        __builtin_amdgcn_wave_barrier();
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            U128 kh, kl;
            kh.u = *reinterpret_cast<const u32x4*>(ldsV + fo + 16 * s);
            if constexpr (SPLIT) {
                kl.u = *reinterpret_cast<const u32x4*>(ldsV + 32 * PITCH + fo + 16 * s);
                sacc = MFMA32(kl.v, qh[s], sacc);
                sacc = MFMA32(kh.v, ql[s], sacc);
            }
            sacc = MFMA32(kh.v, qh[s], sacc);
        }
        __builtin_amdgcn_wave_barrier();
        stage_tile<HD, PITCH>(ldsV, p.qkv_hi, p.ld, base + 2 * p.D, st_ld, k0, p.N, lane);                   // ... then V into the same space
        if constexpr (SPLIT) stage_tile<HD, PITCH>(ldsV + 32 * PITCH, p.qkv_lo, p.ld, base + 2 * p.D, st_ld, k0, p.N, lane);
        float sv[16], mloc = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = (k0 + acc_row(r, h2)) < p.N && seg_ok<SEG>(p, qrow_c, k0 + acc_row(r, h2));
            sv[r] = ok ? sacc[r] * p.scale : -INFINITY;
            mloc = fmaxf(mloc, sv[r]);
        }
        mloc = half_max(mloc);
        const float mnew = fmaxf(m_i, mloc);                      // finite: every key tile holds >= 1 valid key
        const float mold = m_i;
        const float alpha = fast_exp(m_i - mnew);                     // exp(-inf) = 0 on the first tile
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sv[r] = fast_exp(sv[r] - mnew);                           // invalid keys: exp(-inf) = 0
            lsum += sv[r];
        }
        lsum = half_sum(lsum);
        l_i = l_i * alpha + lsum;
        m_i = mnew;
        if constexpr (DROP) {                                     // dropout on the attention weights (normaliser undropped)
            const unsigned long long key = drop_key(p.drop_seed, p.drop_site);
            const unsigned long long rowbase = ((unsigned long long)bh * p.N + min(qrow, p.N - 1)) * p.N;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sv[r] = drop_keep_attn(key, rowbase, (uint32_t)(k0 + acc_row(r, h2)), p.drop_thr) ? sv[r] * p.drop_scale : 0.f;
        }
        // rescale the running output only when some row's maximum moved (wave-uniform test): after the first few key tiles of a long
        // sequence it almost never does, and alpha == 1 exactly for every row then (6 x 16 multiplies per tile at hd = 192)
        if (__builtin_amdgcn_ballot_w64(mnew != mold) != 0ull) {
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        }

        U128 ph[2], pl[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; j += 2) split_bf16x2(sv[8 * s2 + j], sv[8 * s2 + j + 1], ph[s2].w[j / 2], pl[s2].w[j / 2]);
        __syncthreads();                                          // V tile is staged
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
            bf16x8 vh[2], vl[2];
            if constexpr (SPLIT) gather_frag_2x2<HD, PITCH>(ldsV, d * 32 + l31, ldsV + 32 * PITCH, d * 32 + l31, h2, vh, vl);
            else gather_frag_s2<HD, PITCH>(ldsV, h2, d * 32 + l31, vh);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                if constexpr (SPLIT) {
                    o[d] = MFMA32(vl[s2], ph[s2].v, o[d]);
                    o[d] = MFMA32(vh[s2], pl[s2].v, o[d]);
                }
                o[d] = MFMA32(vh[s2], ph[s2].v, o[d]);
            }
        }
    }

    if (active && qok) {
        const float inv = 1.0f / l_i;
        const long orow = ((long)b * p.sb + (long)qrow * p.st) * p.ldo + h * HD;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int c = 0; c < 4; ++c) {                          // rows 8c + 4*h2 + {0..3}: 4 consecutive d
                union { uint2 u; bf16_t h[4]; } hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) split_bf16(o[d][4 * c + i] * inv, hi.h[i], lo.h[i]);
                const int dcol = d * 32 + 8 * c + 4 * h2;
                if (HD % 32 != 0 && dcol >= HD) continue;
                const long off = orow + dcol;
                *reinterpret_cast<uint2*>(p.out_hi + off) = hi.u;
                if (p.out_lo) *reinterpret_cast<uint2*>(p.out_lo + off) = lo.u;
            }
        if (h2 == 0 && p.lse) p.lse[(long)bh * p.N + qrow] = m_i + logf(l_i);
    }
}

// ------------------------------------------------------------------------------------------- forward, ONE tile (N <= 32) at hd = 256, split
// The first pass of group_embed (cfg-3): 15-token sequences packed two to a 32-row tile (SEG), 18 816 (pair, head) items per launch, 92 KB of
// Q / K / V planes each -- a pure stream.  attn_fwd_kernel stages Q, K and V one after the other through the wave's only tile space and sits out
// three memory round trips per item (0.55 ms per launch = 4.2 TB/s).  Here a wave requests Q AND K (four planes, 64 x 16 bytes per lane) before
// it waits for anything -- the 256 staging registers are free at that point: no fragments, no output tile yet -- and V as soon as the Q fragments
// have been read back out of LDS, so that V's round trip runs under the score MFMAs and the softmax.  One exposed round trip per item.
template <bool SEG>
__global__ __launch_bounds__(256) void attn_fwd_tile256_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int HD = 256, NS = HD / 16, NDB = HD / 32, PITCH = fwd_pitch(HD), NCH = 16;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int h2 = lane >> 5, l31 = lane & 31;
    bf16_t* ldsT = reinterpret_cast<bf16_t*>(smem) + wave * (2 * 32 * PITCH);       // [hi plane][lo plane] of the tile in hand
    const long W = (long)p.Bb * p.H;
    long item = (long)blockIdx.x * (blockDim.x >> 6) + wave;
    const bool active = item < W;
    if (!active) item = W - 1;
    const int bh = (int)item, h = bh % p.H, b = bh / p.H;
    const long st_ld = p.st * p.ld;
    const long base = (long)b * p.sb * p.ld + h * HD;
    const int qrow = l31;
    const bool qok = qrow < p.N;
    const int qrow_c = min(qrow, p.N - 1);
    // chunk i of a plane: tile row 2 i + lane / 32 (clamped to the last token), 16-byte column lane % 32.  Row pairs inside the sequence: a
    // wave-uniform base advanced by two rows per i + ONE lane offset; the clamped pairs at the end (rows >= N) build theirs per lane
    const unsigned loff = (unsigned)(((long)h2 * st_ld + l31 * 8) * 2);
    const int npair = p.N >> 1;                                        // pairs (2 i, 2 i + 1) with both rows < N
    const long pair_bytes = 4 * st_ld;
    const char* const gh = reinterpret_cast<const char*>(p.qkv_hi + base), *const gl = reinterpret_cast<const char*>(p.qkv_lo + base);
    const long kofs = (long)p.D * 2, vofs = (long)p.D * 4;            // bytes from a row's Q to its K / V
    auto ld = [&](const char* g, int i) -> u32x4 {
        if (i < npair) return *reinterpret_cast<const u32x4*>(g + i * pair_bytes + loff);      // wave-uniform choice
        return *reinterpret_cast<const u32x4*>(g + (unsigned)(((long)min(2 * i + h2, p.N - 1) * st_ld + l31 * 8) * 2));
    };
    bf16_t* const sdst = ldsT + h2 * PITCH + l31 * 8;
    auto put = [&](const u32x4 (&rh)[NCH], const u32x4 (&rl)[NCH]) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            *reinterpret_cast<u32x4*>(sdst + 2 * i * PITCH) = rh[i];
            *reinterpret_cast<u32x4*>(sdst + 32 * PITCH + 2 * i * PITCH) = rl[i];
        }
    };
    u32x4 ah[NCH], al[NCH], bh_[NCH], bl[NCH];                         // a: Q, later V; b: K
#pragma unroll
    for (int i = 0; i < NCH; ++i) { ah[i] = ld(gh, i); al[i] = ld(gl, i); }
#pragma unroll
    for (int i = 0; i < NCH; ++i) { bh_[i] = ld(gh + kofs, i); bl[i] = ld(gl + kofs, i); }
    put(ah, al);
    __builtin_amdgcn_wave_barrier();
    bf16x8 qh[NS], ql[NS];
    const int fo = l31 * PITCH + h2 * 8;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        U128 t;
        t.u = *reinterpret_cast<const u32x4*>(ldsT + fo + 16 * s); qh[s] = t.v;
        t.u = *reinterpret_cast<const u32x4*>(ldsT + 32 * PITCH + fo + 16 * s); ql[s] = t.v;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < NCH; ++i) { ah[i] = ld(gh + vofs, i); al[i] = ld(gl + vofs, i); }   // V on its way
    put(bh_, bl);                                                      // K
    __builtin_amdgcn_wave_barrier();
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        U128 kh, kl;
        kh.u = *reinterpret_cast<const u32x4*>(ldsT + fo + 16 * s);
        kl.u = *reinterpret_cast<const u32x4*>(ldsT + 32 * PITCH + fo + 16 * s);
        sacc = MFMA32(kl.v, qh[s], sacc);
        sacc = MFMA32(kh.v, ql[s], sacc);
        sacc = MFMA32(kh.v, qh[s], sacc);
    }
    float sv[16], mloc = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const bool ok = acc_row(r, h2) < p.N && seg_ok<SEG>(p, qrow_c, acc_row(r, h2));
        sv[r] = ok ? sacc[r] * p.scale : -INFINITY;
        mloc = fmaxf(mloc, sv[r]);
    }
    const float m_i = half_max(mloc);                                  // finite: a row always sees its own segment
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sv[r] = fast_exp(sv[r] - m_i); lsum += sv[r]; }
    const float l_i = half_sum(lsum);
    U128 ph[2], pl[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; j += 2) split_bf16x2(sv[8 * s2 + j], sv[8 * s2 + j + 1], ph[s2].w[j / 2], pl[s2].w[j / 2]);
    __builtin_amdgcn_wave_barrier();                                   // the K fragments have been read
    put(ah, al);                                                       // V
    __builtin_amdgcn_wave_barrier();
    const float inv = 1.0f / l_i;
    f32x16 o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d) {
        bf16x8 vh[2], vl[2];
        gather_frag_2x2<HD, PITCH>(ldsT, d * 32 + l31, ldsT + 32 * PITCH, d * 32 + l31, h2, vh, vl);
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            o[d] = MFMA32(vl[s2], ph[s2].v, o[d]);
            o[d] = MFMA32(vh[s2], pl[s2].v, o[d]);
            o[d] = MFMA32(vh[s2], ph[s2].v, o[d]);
        }
    }
    // The output leaves through the tile space: a lane owns ONE query row and four consecutive d per register group -- stored from there, an
    // instruction writes 32 rows x 16 bytes (64 partial lines; measured: 140 of 552 us per launch at the cfg-3 geometry).  Transposed through LDS
    // it writes two whole 512-byte row segments.
    __builtin_amdgcn_wave_barrier();                                   // the V fragments have been read
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            union { uint2 u; bf16_t h[4]; } hi, lo;
#pragma unroll
            for (int k = 0; k < 4; ++k) split_bf16(o[d][4 * c + k] * inv, hi.h[k], lo.h[k]);
            const int at = l31 * PITCH + d * 32 + 8 * c + 4 * h2;
            *reinterpret_cast<uint2*>(ldsT + at) = hi.u;
            *reinterpret_cast<uint2*>(ldsT + 32 * PITCH + at) = lo.u;
        }
    __builtin_amdgcn_wave_barrier();
    if (active) {
        const long orow0 = (long)b * p.sb * p.ldo + h * HD + l31 * 8;
#pragma unroll
        for (int i2 = 0; i2 < NCH; ++i2) {
            const int r = 2 * i2 + h2;
            if (r < p.N) {
                const long off = orow0 + (long)r * p.st * p.ldo;
                *reinterpret_cast<u32x4*>(p.out_hi + off) = *reinterpret_cast<const u32x4*>(sdst + 2 * i2 * PITCH);
                if (p.out_lo) *reinterpret_cast<u32x4*>(p.out_lo + off) = *reinterpret_cast<const u32x4*>(sdst + 32 * PITCH + 2 * i2 * PITCH);
            }
        }
    }
    if (active && qok && h2 == 0 && p.lse) p.lse[(long)bh * p.N + qrow] = m_i + logf(l_i);
}

// ------------------------------------------------------------------------------------------- backward: dQ (+ delta)
template <int HD, bool SEG = false>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NS = HD / 16, NDB = (HD + 31) / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h2 = lane >> 5, l31 = lane & 31;
    bf16_t* ldsK = reinterpret_cast<bf16_t*>(smem) + wave * (32 * HD);

    const int QT = (p.N + 31) / 32;
    const long W = (long)p.Bb * p.H * QT;
    long item = (long)blockIdx.x * (blockDim.x >> 6) + wave;
    const bool active = item < W;
    if (!active) item = W - 1;
    const int qt = (int)(item % QT);
    const int bh = (int)(item / QT);
    const int h = bh % p.H, b = bh / p.H;
    const long st_ld = p.st * p.ld;
    const long base = (long)b * p.sb * p.ld + h * HD;
    const int q0 = qt * 32, qrow = q0 + l31;
    const bool qok = qrow < p.N;
    const int qrow_c = min(qrow, p.N - 1);
    const long tokrow = (long)b * p.sb + (long)qrow_c * p.st;

    bf16x8 qf[NS], dof[NS];
    float delta = 0.f;
    {
        const long off = base + (long)qrow_c * st_ld + h2 * 8;
        const long doff = tokrow * p.lddo + h * HD + h2 * 8;
        const long ooff = tokrow * p.ldo + h * HD + h2 * 8;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            qf[s] = ld_frag(p.qkv_hi + off + 16 * s);
            dof[s] = ld_frag(p.dout + doff + 16 * s);
            U128 a, ol, d;
            a.v = ld_frag(p.out_hi + ooff + 16 * s);
            ol.v = ld_frag((p.out_lo ? p.out_lo : p.out_hi) + ooff + 16 * s);
            d.v = dof[s];
            const float lo_on = p.out_lo ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) delta += bf2f(d.h[j]) * (bf2f(a.h[j]) + lo_on * bf2f(ol.h[j]));
        }
        delta = half_sum(delta);
    }
    const float lse_q = p.lse[(long)bh * p.N + qrow_c];
    if (active && qok && h2 == 0) p.delta[(long)bh * p.N + qrow] = delta;

    f32x16 dq[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;

    const int KT = (p.N + 31) / 32;
    for (int kt = 0; kt < KT; ++kt) {
        const int k0 = kt * 32;
        __syncthreads();
        stage_tile<HD>(ldsK, p.qkv_hi, p.ld, base + p.D, st_ld, k0, p.N, lane);
        const int krow = min(k0 + l31, p.N - 1);
        const long koff = base + p.D + (long)krow * st_ld + h2 * 8;
        const long voff = base + 2 * p.D + (long)krow * st_ld + h2 * 8;
        f32x16 sacc, dpacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            sacc = MFMA32(ld_frag(p.qkv_hi + koff + 16 * s), qf[s], sacc);      // S^T  = K . Q^T
            dpacc = MFMA32(ld_frag(p.qkv_hi + voff + 16 * s), dof[s], dpacc);   // dP^T = V . dO^T
        }
        U128 dsf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 8 * s2 + j;
                const bool ok = (k0 + acc_row(r, h2)) < p.N && seg_ok<SEG>(p, qrow_c, k0 + acc_row(r, h2));
                const float pr = ok ? fast_exp(sacc[r] * p.scale - lse_q) : 0.f;
                float dpn = dpacc[r];
                if (p.drop_thr)
                    dpn = drop_keep_attn(drop_key(p.drop_seed, p.drop_site), ((unsigned long long)bh * p.N + qrow_c) * p.N,
                                         (uint32_t)(k0 + acc_row(r, h2)), p.drop_thr) ? dpn * p.drop_scale : 0.f;
                dsf[s2].h[j] = f2bf(pr * (dpn - delta) * p.scale);
            }
        __syncthreads();
#pragma unroll
        for (int d = 0; d < NDB; d += 2) {
            bf16x8 k0f[2], k1f[2];
            if (d + 1 < NDB) gather_frag_2x2<HD>(ldsK, d * 32 + l31, ldsK, (d + 1) * 32 + l31, h2, k0f, k1f);
            else gather_frag_s2<HD>(ldsK, h2, d * 32 + l31, k0f);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                dq[d] = MFMA32(k0f[s2], dsf[s2].v, dq[d]);                                    // dQ^T = K^T . dS^T
                if (d + 1 < NDB) dq[d + 1] = MFMA32(k1f[s2], dsf[s2].v, dq[d + 1]);
            }
        }
    }
    if (active && qok) {
        const long orow = tokrow * p.lddq + h * HD;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                union { uint2 u; bf16_t h[4]; } v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v.h[i] = f2bf(dq[d][4 * c + i]);
                const int dcol = d * 32 + 8 * c + 4 * h2;
                if (HD % 32 != 0 && dcol >= HD) continue;
                *reinterpret_cast<uint2*>(p.dqkv + orow + dcol) = v.u;
            }
    }
}

// ------------------------------------------------------------------------------------------- backward: dK, dV
// DSPLIT > 1 splits the output d-range of dK/dV over blockIdx.y (register budget at hd = 256).
template <int HD, int DSPLIT, bool SEG = false>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NS = HD / 16, NDB = ((HD + 31) / 32) / DSPLIT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h2 = lane >> 5, l31 = lane & 31;
    constexpr int WAVE_LDS = 2 * 32 * HD * 2 + 256;                   // Q | dO tiles (bf16) + lse / delta of the tile (fp32)
    bf16_t* ldsQ = reinterpret_cast<bf16_t*>(smem + wave * WAVE_LDS);
    bf16_t* ldsDO = ldsQ + 32 * HD;
    float* ldsR = reinterpret_cast<float*>(ldsDO + 32 * HD);          // [0..31] lse, [32..63] delta
    const int dblk0 = blockIdx.y * NDB;

    const int KT = (p.N + 31) / 32;
    const long W = (long)p.Bb * p.H * KT;
    long item = (long)blockIdx.x * (blockDim.x >> 6) + wave;
    const bool active = item < W;
    if (!active) item = W - 1;
    const int kt = (int)(item % KT);
    const int bh = (int)(item / KT);
    const int h = bh % p.H, b = bh / p.H;
    const long st_ld = p.st * p.ld;
    const long base = (long)b * p.sb * p.ld + h * HD;
    const int k0 = kt * 32, krow = k0 + l31;
    const bool kok = krow < p.N;
    const int krow_c = min(krow, p.N - 1);

    bf16x8 kf[NS], vf[NS];
    {
        const long koff = base + p.D + (long)krow_c * st_ld + h2 * 8;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            kf[s] = ld_frag(p.qkv_hi + koff + 16 * s);
            vf[s] = ld_frag(p.qkv_hi + koff + p.D + 16 * s);
        }
    }
    f32x16 dk[NDB], dv[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[d][r] = 0.f; dv[d][r] = 0.f; }

    const long dobase = (long)b * p.sb * p.lddo + h * HD;
    const long st_lddo = p.st * p.lddo;
    const int QT = (p.N + 31) / 32;
    for (int qt = 0; qt < QT; ++qt) {
        const int q0 = qt * 32;
        __syncthreads();
        stage_tile<HD>(ldsQ, p.qkv_hi, p.ld, base, st_ld, q0, p.N, lane);
        stage_tile<HD>(ldsDO, p.dout, p.lddo, dobase, st_lddo, q0, p.N, lane);
        const int qrow = min(q0 + l31, p.N - 1);
        // one coalesced load per query instead of 32 broadcast loads per lane inside the register loop below
        ldsR[lane] = (h2 == 0 ? p.lse : p.delta)[(long)bh * p.N + qrow];
        const long qoff = base + (long)qrow * st_ld + h2 * 8;
        const long dooff = dobase + (long)qrow * st_lddo + h2 * 8;
        f32x16 sacc, dpacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            sacc = MFMA32(ld_frag(p.qkv_hi + qoff + 16 * s), kf[s], sacc);     // S  = Q . K^T   (col = key)
            dpacc = MFMA32(ld_frag(p.dout + dooff + 16 * s), vf[s], dpacc);    // dP = dO . V^T
        }
        __syncthreads();                                          // tiles + lse / delta are staged
        U128 pf[2], dsf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 8 * s2 + j;
                const int q = q0 + acc_row(r, h2);
                const bool ok = kok && (q < p.N) && seg_ok<SEG>(p, q, krow_c);
                const int qc = min(q, p.N - 1);
                const float lse_r = ldsR[acc_row(r, h2)];
                const float del_r = ldsR[32 + acc_row(r, h2)];
                const float pr = ok ? fast_exp(sacc[r] * p.scale - lse_r) : 0.f;
                float dm = 1.f;
                if (p.drop_thr)
                    dm = drop_keep_attn(drop_key(p.drop_seed, p.drop_site), ((unsigned long long)bh * p.N + qc) * p.N, (uint32_t)krow_c, p.drop_thr)
                             ? p.drop_scale : 0.f;
                pf[s2].h[j] = f2bf(pr * dm);
                dsf[s2].h[j] = f2bf(pr * (dpacc[r] * dm - del_r) * p.scale);
            }
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
            const int col = (dblk0 + d) * 32 + l31;
            bf16x8 fo[2], fq[2];
            gather_frag_2x2<HD>(ldsDO, col, ldsQ, col, h2, fo, fq);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                dv[d] = MFMA32(fo[s2], pf[s2].v, dv[d]);      // dV^T = dO^T . P
                dk[d] = MFMA32(fq[s2], dsf[s2].v, dk[d]);     // dK^T = Q^T . dS
            }
        }
    }
    if (active && kok) {
        const long orow = ((long)b * p.sb + (long)krow * p.st) * p.lddq + h * HD;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                union { uint2 u; bf16_t h[4]; } a, v;
#pragma unroll
                for (int i = 0; i < 4; ++i) { a.h[i] = f2bf(dk[d][4 * c + i]); v.h[i] = f2bf(dv[d][4 * c + i]); }
                const int dcol = (dblk0 + d) * 32 + 8 * c + 4 * h2;
                if (HD % 32 != 0 && dcol >= HD) continue;
                const long off = orow + dcol;
                *reinterpret_cast<uint2*>(p.dqkv + off + p.D) = a.u;
                *reinterpret_cast<uint2*>(p.dqkv + off + 2 * p.D) = v.u;
            }
    }
}

// ------------------------------------------------------------------------------------------- backward: dK, dV, long sequences
// Same mathematics as attn_bwd_dkv_kernel, organised for long sequences (cfg-3's encoder layer: N = 12 544 per (batch, head)).
// There the per-wave kernel re-streams all of Q and dO from L2 / HBM once per 32-key tile, with both the LDS staging loads and
// the MFMA row fragments coming from global memory and nothing prefetched (PMC: 71 % of wave cycles waiting, MFMA 7 % busy).
// Here the four waves of a workgroup own four consecutive key tiles of one (batch, head) and SHARE one stream of query tiles:
// all 256 threads stage Q / dO / lse / delta of tile t+1 into the other LDS buffer (prefetched into registers before tile t's
// arithmetic), every operand of the MFMAs comes from LDS (row fragments by ds_read_b128 on a padded pitch, transposed fragments
// by transpose reads), and there is one barrier per tile: a quarter of the staging traffic, no global fragment loads, and the
// load latency hidden behind the previous tile.
// MASK: the instantiation that reads S3dAttnArgs::drop_mask (kept apart: its extra state took the hd = 64 kernel of the point path from
// 128 to 136 registers = three waves per SIMD instead of four, 59.6 -> 91.8 us)
// ABL: timing ablations (s3d_debug_knob 3; results wrong), 0 = the product
template <int HD, int DSPLIT, int NWV = 4, bool MASK = false, int ABL = 0>
__global__ __launch_bounds__(64 * NWV) void attn_bwd_dkv_coop_kernel(const AttnArgs p) {
    constexpr int NTHR = 64 * NWV;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NS = HD / 16, NDB = ((HD + 31) / 32) / DSPLIT, CPR = HD / 8;
    constexpr int PITCH = HD + 8;                                      // 16-byte pad: conflict-free ds_read_b128 of a column of rows
    constexpr int BUF_BYTES = 2 * 32 * PITCH * 2 + 256 + (MASK ? NWV * 128 : 0);   // Q | dO tiles + lse / delta [+ the waves' dropout-mask words]
    constexpr int NCH = (32 * CPR + NTHR - 1) / NTHR;                        // 16-byte chunks per thread per tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h2 = lane >> 5, l31 = lane & 31;
    const int dblk0 = blockIdx.y * NDB;

    const int KT = (p.N + 31) / 32, KTB = (KT + NWV - 1) / NWV;
    const int bh = blockIdx.x / KTB;
    int kt = (blockIdx.x % KTB) * NWV + wave;
    const bool active = kt < KT;
    kt = min(kt, KT - 1);
    const int h = bh % p.H, b = bh / p.H;
    const long st_ld = p.st * p.ld;
    const long base = (long)b * p.sb * p.ld + h * HD;
    const int k0 = kt * 32, krow = k0 + l31;
    const bool kok = krow < p.N;
    const float sc2 = p.scale * 1.4426950408889634f;                  // scores -> log2 units (lse is staged pre-scaled)
    const int krow_c = min(krow, p.N - 1);
    const unsigned long long dkey = p.drop_thr ? drop_key(p.drop_seed, p.drop_site) : 0ull;
    const DropRow dcol = drop_row(dkey, (unsigned long long)bh * p.N * p.N + (krow_c & ~1));  // hash index = (even key of the pair) + q * N

    bf16x8 kf[NS], vf[NS];
    {
        const long koff = base + p.D + (long)krow_c * st_ld + h2 * 8;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            kf[s] = ld_frag(p.qkv_hi + koff + 16 * s);
            vf[s] = ld_frag(p.qkv_hi + koff + p.D + 16 * s);
        }
    }
    f32x16 dk[NDB], dv[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[d][r] = 0.f; dv[d][r] = 0.f; }

    const long dobase = (long)b * p.sb * p.lddo + h * HD;
    const long st_lddo = p.st * p.lddo;
    const int QT = (p.N + 31) / 32;

    // staging registers, TWO sets (DEEP; set t & 1 holds query tile t): the loads of tile t + 2 are issued at the top of iteration t and stored at
    // the end of iteration t + 1.  One tile ahead was not enough at hd = 192: a q-tile iteration lasts ~2.2 us, the mask words (one 128-byte
    // line per wave and tile, 50 KB apart) and the first touch of a Q / dO tile miss L2, and a wave's loads return in order -- the staging
    // mechanism on cache-hot addresses measured 3.5 - 4.3 ms less per cfg-3 launch than on the real ones (profiles/r06_attn_backward_staging.txt)
    constexpr bool DEEP = MASK && HD >= 192 && NWV == 4;
    constexpr bool DEEPQ = DEEP && ABL == 8;                           // the Q / dO tiles too (48 staging registers: spills at hd = 192)
    constexpr int NSET = DEEP ? 2 : 1, NSETQ = DEEPQ ? 2 : 1;
    u32x4 rq[NSETQ][NCH], rd[NSETQ][NCH];
    float rr = 0.f;                                                    // lse / delta: one tile ahead (the bh's workgroups share the lines)
    [[maybe_unused]] uint32_t rm[NSET] = {};
    // S3dAttnArgs::drop_mask: the 32 words (one per query row) of tile (query tile, THIS wave's key tile), staged with the query tile
    [[maybe_unused]] const unsigned int* mwave = MASK ? p.drop_mask + ((long)bh * QT * KT + kt) * 32 + l31 : nullptr;
    // whole tiles: wave-uniform bases + one 32-bit byte offset per chunk and tensor (see attn_fwd_coop_pipe_kernel's gload); GA: hd >= 192
    constexpr bool GA = HD >= 192 && ABL != 7;
    [[maybe_unused]] unsigned qoff[GA ? NCH : 1], dooff[GA ? NCH : 1], qoff_last[GA ? NCH : 1], dooff_last[GA ? NCH : 1];
    if constexpr (GA) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = min(tid + NTHR * i, 32 * CPR - 1);
            const int r = c / CPR, cc = c % CPR, rl = min(r, p.N - 1 - (QT - 1) * 32);
            qoff[i] = (unsigned)(((long)r * st_ld + cc * 8) * 2);       dooff[i] = (unsigned)(((long)r * st_lddo + cc * 8) * 2);
            qoff_last[i] = (unsigned)(((long)rl * st_ld + cc * 8) * 2); dooff_last[i] = (unsigned)(((long)rl * st_lddo + cc * 8) * 2);
        }
    }
    const char* const gq = reinterpret_cast<const char*>(p.qkv_hi + base), *const gdo = reinterpret_cast<const char*>(p.dout + dobase);
    auto gload_qd = [&](auto set_tag, int q0) {
        constexpr int S = DEEPQ ? decltype(set_tag)::value : 0;
        if constexpr (GA) {
            const int qt_ = min(q0 >> 5, QT - 1);
            const bool last = qt_ >= QT - 1;                           // block-uniform
            const long qb = (long)qt_ * 64 * st_ld, db = (long)qt_ * 64 * st_lddo;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                rq[S][i] = *reinterpret_cast<const u32x4*>(gq + qb + (last ? qoff_last[i] : qoff[i]));
                rd[S][i] = *reinterpret_cast<const u32x4*>(gdo + db + (last ? dooff_last[i] : dooff[i]));
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = min(tid + NTHR * i, 32 * CPR - 1);
            const int r = c / CPR, cc = c % CPR;
            const long t = ABL == 7 ? r : min(q0 + r, p.N - 1);
            rq[S][i] = *reinterpret_cast<const u32x4*>(p.qkv_hi + base + t * st_ld + cc * 8);
            rd[S][i] = *reinterpret_cast<const u32x4*>(p.dout + dobase + t * st_lddo + cc * 8);
        }
    };
    auto gload_rr = [&](int q0) {
        if (tid < 64) rr = (tid < 32 ? p.lse : p.delta)[(long)bh * p.N + min(q0 + (tid & 31), p.N - 1)];
    };
    auto gload_aux = [&](auto set_tag, int q0) {                       // the wave's mask words of the tile
        constexpr int S = decltype(set_tag)::value;
        if constexpr (MASK) { if (h2 == 0) rm[S] = mwave[ABL == 6 ? 0 : (long)(q0 >> 5) * KT * 32]; }
    };
    auto lstore = [&](auto set_tag, int buf) {
        constexpr int S = decltype(set_tag)::value, SQ = DEEPQ ? S : 0;
        bf16_t* q = reinterpret_cast<bf16_t*>(smem + buf * BUF_BYTES);
        bf16_t* dO = q + 32 * PITCH;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + NTHR * i;
            if (32 * CPR % NTHR == 0 || c < 32 * CPR) {
                const int r = c / CPR, cc = c % CPR;
                *reinterpret_cast<u32x4*>(q + r * PITCH + cc * 8) = rq[SQ][i];
                *reinterpret_cast<u32x4*>(dO + r * PITCH + cc * 8) = rd[SQ][i];
            }
        }
        // lse is staged in log2 units: P = exp2(S * scale * log2 e - lse2)
        if (tid < 64) reinterpret_cast<float*>(dO + 32 * PITCH)[tid] = tid < 32 ? rr * 1.4426950408889634f : rr;
        if constexpr (MASK) { if (h2 == 0) reinterpret_cast<uint32_t*>(dO + 32 * PITCH)[64 + wave * 32 + l31] = rm[S]; }
    };
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, DEEP ? 1 : 0>;

    gload_qd(Set0{}, 0); gload_rr(0); gload_aux(Set0{}, 0);
    lstore(Set0{}, 0);
    if constexpr (DEEP) { if (QT > 1) { if constexpr (DEEPQ) gload_qd(Set1{}, 32); gload_aux(Set1{}, 32); } }
    __syncthreads();
    // iteration qt computes on buffer qt & 1; DEEP: loads tile qt + 2 into set qt & 1 (stored by the previous iteration), stores tile qt + 1
    // from set (qt + 1) & 1 into buffer (qt + 1) & 1 -- last read in iteration qt - 1, behind that iteration's barrier.  No compute happens
    // between the prologue's barrier and the loop (cf. the forward's prologue race, profiles/r06_attn_prologue_race.txt).
    auto body = [&](int qt, auto cur_tag, auto nxt_tag) {
        const int q0 = qt * 32;
        const bool more = qt + 1 < QT;                                // block-uniform
        // order: the Q / dO loads whose data the END of this iteration stores first, the deep ones (two iterations to land) behind them
        if constexpr (!DEEPQ) { if (more && ABL != 4) gload_qd(cur_tag, ABL == 5 ? 0 : q0 + 32); }   // (ABL 5: the staging mechanism on a cache-hot tile)
        if (more && ABL != 4) gload_rr(ABL == 5 ? 0 : q0 + 32);
        if constexpr (DEEP) {
            if (qt + 2 < QT && ABL != 4) { if constexpr (DEEPQ) gload_qd(cur_tag, q0 + 64); gload_aux(cur_tag, ABL == 5 ? 0 : q0 + 64); }
        } else { if (more && ABL != 4) gload_aux(cur_tag, ABL == 5 ? 0 : q0 + 32); }
        const bf16_t* ldsQ = reinterpret_cast<const bf16_t*>(smem + (qt & 1) * BUF_BYTES);
        const bf16_t* ldsDO = ldsQ + 32 * PITCH;
        const float* ldsR = reinterpret_cast<const float*>(ldsDO + 32 * PITCH);
        f32x16 sacc, dpacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
        if constexpr (HD >= 192) {   // fragments of k-step s + 1 are read before the MFMAs of step s (round 6, see attn_bwd_dq_coop_kernel)
            bf16x8 qf2[2], of2[2];
            qf2[0] = *reinterpret_cast<const bf16x8*>(ldsQ + l31 * PITCH + h2 * 8);
            of2[0] = *reinterpret_cast<const bf16x8*>(ldsDO + l31 * PITCH + h2 * 8);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s + 1 < NS) {
                    const int off = l31 * PITCH + 16 * (s + 1) + h2 * 8;
                    qf2[(s + 1) & 1] = *reinterpret_cast<const bf16x8*>(ldsQ + off);
                    of2[(s + 1) & 1] = *reinterpret_cast<const bf16x8*>(ldsDO + off);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ABL != 3) {
                sacc = MFMA32(qf2[s & 1], kf[s], sacc);                                       // S  = Q . K^T   (col = key)
                dpacc = MFMA32(of2[s & 1], vf[s], dpacc);                                     // dP = dO . V^T
                } else { asm volatile("" :: "v"(qf2[s & 1]), "v"(of2[s & 1])); }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int off = l31 * PITCH + 16 * s + h2 * 8;
                sacc = MFMA32(*reinterpret_cast<const bf16x8*>(ldsQ + off), kf[s], sacc);
                dpacc = MFMA32(*reinterpret_cast<const bf16x8*>(ldsDO + off), vf[s], dpacc);
            }
        }
        // lse / delta of the 16 query rows this lane's accumulator registers belong to: rows come in four runs of four consecutive
        // ones (acc_row) -> eight 16-byte LDS reads instead of 32 scalar ones; the two bf16 operands are converted in pairs
        f32x4 lse4[4], del4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            lse4[g] = *reinterpret_cast<const f32x4*>(ldsR + 8 * g + 4 * h2);
            del4[g] = *reinterpret_cast<const f32x4*>(ldsR + 32 + 8 * g + 4 * h2);
        }
        U128 pf[2], dsf[2];
        if constexpr (!MASK) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    float pv[2], dv2[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int r = 8 * s2 + j + e, qr = acc_row(r, h2);
                        const int q = q0 + qr;
                        // columns of keys >= N are never stored, so only the ragged last QUERY tile needs a select (clamped rows would count twice)
                        float pr = __builtin_amdgcn_exp2f(fmaf(sacc[r], sc2, -lse4[r >> 2][r & 3]));
                        if (q0 + 32 > p.N) pr = (q < p.N) ? pr : 0.f;
                        float dm = 1.f; // rows q >= N: pr = 0 kills both products, so the mask index needs no clamp (keeps q * N linear in r)
                        if (p.drop_thr) dm = drop_half(drop_hash_at(dcol, (uint32_t)q * (uint32_t)p.N), (uint32_t)krow_c & 1u, p.drop_thr) ? p.drop_scale : 0.f;
                        pv[e] = pr * dm;
                        dv2[e] = pr * (dpacc[r] * dm - del4[r >> 2][r & 3]) * p.scale;
                    }
                    pf[s2].w[j / 2] = f2bf2(pv[0], pv[1]);
                    dsf[s2].w[j / 2] = f2bf2(dv2[0], dv2[1]);
                }
        } else if constexpr (ABL == 1) {                               // (no softmax / dS arithmetic: the accumulators are only kept alive)
            asm volatile("" :: "v"(sacc), "v"(dpacc));
#pragma unroll
            for (int q = 0; q < 4; ++q) { pf[0].w[q] = pf[1].w[q] = 0x3c003c00u; dsf[0].w[q] = dsf[1].w[q] = 0x3c003c00u; }
        } else {                                                       // the mask from the bits the forward stored
            u32x4 mw4[4];                                              // the words of this lane's 16 query rows (four runs of four, like lse4)
#pragma unroll
            for (int g = 0; g < 4; ++g) mw4[g] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint32_t*>(ldsR) + 64 + wave * 32 + 8 * g + 4 * h2);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    float pv[2], dv2[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int r = 8 * s2 + j + e;
                        float pr = __builtin_amdgcn_exp2f(fmaf(sacc[r], sc2, -lse4[r >> 2][r & 3]));
                        if (q0 + 32 > p.N) pr = (q0 + acc_row(r, h2) < p.N) ? pr : 0.f;
                        const float dm = __uint_as_float((uint32_t)__builtin_amdgcn_sbfe((int)mw4[r >> 2][r & 3], l31, 1) & __float_as_uint(p.drop_scale));
                        pv[e] = pr * dm;
                        dv2[e] = pr * (dpacc[r] * dm - del4[r >> 2][r & 3]) * p.scale;
                    }
                    pf[s2].w[j / 2] = f2bf2(pv[0], pv[1]);
                    dsf[s2].w[j / 2] = f2bf2(dv2[0], dv2[1]);
                }
        }
        if constexpr (HD % 32 == 0 && NDB >= 2 && HD >= 192) {
            // the transposed dO / Q fragments of d-block d + 1 are in flight during block d's four MFMAs
            Frag2x2 fr[2];
            gather_issue_2x2<HD, PITCH>(ldsDO, dblk0 * 32 + l31, ldsQ, dblk0 * 32 + l31, h2, fr[0]);
#pragma unroll
            for (int d = 0; d < NDB; ++d) {
                bf16x8 fo[2], fq[2];
                gather_wait(fr[d & 1], fo, fq);
                if (d + 1 < NDB) gather_issue_2x2<HD, PITCH>(ldsDO, (dblk0 + d + 1) * 32 + l31, ldsQ, (dblk0 + d + 1) * 32 + l31, h2, fr[(d + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ABL == 2) { asm volatile("" :: "v"(fo[0]), "v"(fo[1]), "v"(fq[0]), "v"(fq[1]), "v"(pf[0].v), "v"(pf[1].v), "v"(dsf[0].v), "v"(dsf[1].v)); }
                else {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    dv[d] = MFMA32(fo[s2], pf[s2].v, dv[d]);      // dV^T = dO^T . P
                    dk[d] = MFMA32(fq[s2], dsf[s2].v, dk[d]);     // dK^T = Q^T . dS
                }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int d = 0; d < NDB; ++d) {
                const int col = (dblk0 + d) * 32 + l31;
                bf16x8 fo[2], fq[2];
                gather_frag_2x2<HD, PITCH>(ldsDO, col, ldsQ, col, h2, fo, fq);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    dv[d] = MFMA32(fo[s2], pf[s2].v, dv[d]);      // dV^T = dO^T . P
                    dk[d] = MFMA32(fq[s2], dsf[s2].v, dk[d]);     // dK^T = Q^T . dS
                }
            }
        }
        if (more && ABL != 4) lstore(nxt_tag, (qt + 1) & 1);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);                            // (the two copies of the DEEP loop stay apart: merged they spill)
    };
    if constexpr (DEEP) {
        // always an even number of iterations: for an odd QT the last one is a phantom tile (q0 = 32 QT >= N: every probability is selected to
        // zero by the ragged-tile rule, nothing is loaded or stored for it) -- a third copy of the body for the tail made hipcc spill
        for (int qt = 0; qt < QT; qt += 2) { body(qt, Set0{}, Set1{}); body(qt + 1, Set1{}, Set0{}); }
    } else {
        for (int qt = 0; qt < QT; ++qt) body(qt, Set0{}, Set0{});
    }
    if (active && kok) {
        const long orow = ((long)b * p.sb + (long)krow * p.st) * p.lddq + h * HD;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                union { uint2 u; bf16_t h[4]; } a, v;
#pragma unroll
                for (int i = 0; i < 4; ++i) { a.h[i] = f2bf(dk[d][4 * c + i]); v.h[i] = f2bf(dv[d][4 * c + i]); }
                const int dcol = (dblk0 + d) * 32 + 8 * c + 4 * h2;
                if (HD % 32 != 0 && dcol >= HD) continue;
                const long off = orow + dcol;
                *reinterpret_cast<uint2*>(p.dqkv + off + p.D) = a.u;
                *reinterpret_cast<uint2*>(p.dqkv + off + 2 * p.D) = v.u;
            }
    }
}

// ------------------------------------------------------------------------------------------- long sequences: shared key stream
// Counterparts of attn_fwd_kernel / attn_bwd_dq_kernel organised like attn_bwd_dkv_coop_kernel: the four waves of a workgroup own
// four consecutive QUERY tiles of one (batch, head) and share one double-buffered stream of key tiles (K and V rows, hi and lo
// planes in the split forward), staged by all 256 threads with the next tile prefetched into registers.
template <int HD, int NT_, int NTHR = 256>
struct CoopStage {                                                     // NT_ tiles of [32 rows][HD] bf16 per buffer, padded pitch
    static constexpr int CPR = HD / 8, PITCH = HD + 8, TILE = 32 * PITCH;
    static constexpr int NCH = (32 * CPR + NTHR - 1) / NTHR;
    static constexpr int BUF_BYTES = NT_ * TILE * 2;
    u32x4 r[NT_][NCH];
    __device__ __forceinline__ void gload(const bf16_t* const (&src)[NT_], const long (&rowoff)[NT_], const long (&pitch)[NT_], int t0,
                                          int N, int tid) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = min(tid + NTHR * i, 32 * CPR - 1);
            const int row = c / CPR, cc = c % CPR;
            const long t = min(t0 + row, N - 1);
#pragma unroll
            for (int k = 0; k < NT_; ++k) r[k][i] = *reinterpret_cast<const u32x4*>(src[k] + rowoff[k] + t * pitch[k] + cc * 8);
        }
    }
    __device__ __forceinline__ void lstore(unsigned char* buf, int tid) const {
        bf16_t* b = reinterpret_cast<bf16_t*>(buf);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + NTHR * i;
            if (32 * CPR % NTHR == 0 || c < 32 * CPR) {
                const int row = c / CPR, cc = c % CPR;
#pragma unroll
                for (int k = 0; k < NT_; ++k) *reinterpret_cast<u32x4*>(b + k * TILE + row * PITCH + cc * 8) = r[k][i];
            }
        }
    }
};
// The same stream with the tile addresses split into a wave-uniform base per tensor (SGPR pair, advanced per tile) and ONE 32-bit byte offset per
// chunk and tensor that never changes (a second set for the ragged last tile): the loads take the `v_off, s[base]` form and the per-tile
// address arithmetic of gload() (min, two multiplies, 64-bit mad and adds per chunk) disappears -- see attn_fwd_coop_pipe_kernel's gload.
template <int HD, int NT_, int NTHR = 256>
struct CoopStageU : CoopStage<HD, NT_, NTHR> {
    using B = CoopStage<HD, NT_, NTHR>;
    unsigned co[NT_][B::NCH], co_last[NT_][B::NCH];
    const char* gb[NT_];
    long tile_bytes[NT_];
    int last_tile;
    __device__ __forceinline__ void init(const bf16_t* const (&src)[NT_], const long (&rowoff)[NT_], const long (&pitch)[NT_], int N, int tid) {
        last_tile = (N + 31) / 32 - 1;
#pragma unroll
        for (int k = 0; k < NT_; ++k) {
            gb[k] = reinterpret_cast<const char*>(src[k] + rowoff[k]);
            tile_bytes[k] = 64 * pitch[k];
#pragma unroll
            for (int i = 0; i < B::NCH; ++i) {
                const int c = min(tid + NTHR * i, 32 * B::CPR - 1);
                const int row = c / B::CPR, cc = c % B::CPR;
                co[k][i] = (unsigned)(((long)row * pitch[k] + cc * 8) * 2);
                co_last[k][i] = (unsigned)(((long)min(row, N - 1 - last_tile * 32) * pitch[k] + cc * 8) * 2);
            }
        }
    }
    __device__ __forceinline__ void gload_tile(int t) {                // tile index (tiles past the end read the last one)
        const int tc = min(t, last_tile);
        const bool last = tc >= last_tile;                             // block-uniform
#pragma unroll
        for (int k = 0; k < NT_; ++k) {
            const char* b = gb[k] + tc * tile_bytes[k];
#pragma unroll
            for (int i = 0; i < B::NCH; ++i) this->r[k][i] = *reinterpret_cast<const u32x4*>(b + (last ? co_last[k][i] : co[k][i]));
        }
    }
};

// NWV waves per workgroup (4 or 8): with eight, two waves share every SIMD and one's softmax / mask VALU phase runs beside the
// other's MFMAs (the LDS stream and its footprint stay the same; twice the query rows per workgroup)
// NBUF = 1 (round 6, hd = 192 split): ONE LDS buffer of the key stream (51 KB instead of 102) so that TWO workgroups share a CU -- two waves
// per SIMD from different workgroups, one's softmax / mask VALU phase beside the other's MFMAs (the tile in flight waits in registers as
// before; the price is a second barrier per key tile, paid while the other workgroup computes).
template <int HD, bool SPLIT, int NWV = 4, bool DROP = true, int NBUF = 2>
__global__ __launch_bounds__(64 * NWV) void attn_fwd_coop_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NS = HD / 16, NDB = (HD + 31) / 32, NPL = SPLIT ? 2 : 1;
    using ST = CoopStage<HD, 2 * NPL, 64 * NWV>;                       // K_hi [K_lo] V_hi [V_lo]
    constexpr int PITCH = ST::PITCH, TILE = ST::TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h2 = lane >> 5, l31 = lane & 31;

    const int QT = (p.N + 31) / 32, QTB = (QT + NWV - 1) / NWV;
    const int bh = blockIdx.x / QTB;
    int qt = (blockIdx.x % QTB) * NWV + wave;
    const bool active = qt < QT;
    qt = min(qt, QT - 1);
    const int h = bh % p.H, b = bh / p.H;
    const long st_ld = p.st * p.ld;
    const long base = (long)b * p.sb * p.ld + h * HD;
    const int q0 = qt * 32, qrow = q0 + l31;
    const bool qok = qrow < p.N;
    const int qrow_c = min(qrow, p.N - 1);
    const unsigned long long dkey = (DROP && p.drop_thr) ? drop_key(p.drop_seed, p.drop_site) : 0ull;
    const DropRow drow = drop_row(dkey, ((unsigned long long)bh * p.N + qrow_c) * p.N);     // mask index = row base + key column

    bf16x8 qh[NS], ql[SPLIT ? NS : 1];
    {
        const long off = base + (long)qrow_c * st_ld + h2 * 8;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            qh[s] = ld_frag(p.qkv_hi + off + 16 * s);
            if constexpr (SPLIT) ql[s] = ld_frag(p.qkv_lo + off + 16 * s);
        }
    }
    f32x16 o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_i = -INFINITY, l_i = 0.f;
    const float sc2 = p.scale * 1.4426950408889634f;                  // scores -> log2 units

    ST st;
    const bf16_t* src[2 * NPL];
    long rowoff[2 * NPL], pitch[2 * NPL];
#pragma unroll
    for (int k = 0; k < 2 * NPL; ++k) {
        src[k] = (SPLIT && (k & 1)) ? p.qkv_lo : p.qkv_hi;
        rowoff[k] = base + ((SPLIT ? (k >> 1) : k) + 1) * (long)p.D;   // K then V
        pitch[k] = st_ld;
    }
    const int KT = (p.N + 31) / 32;
    st.gload(src, rowoff, pitch, 0, p.N, tid);
    st.lstore(smem, tid);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int k0 = kt * 32;
        const bool more = kt + 1 < KT;
        if (more) st.gload(src, rowoff, pitch, k0 + 32, p.N, tid);
        const bf16_t* tb = reinterpret_cast<const bf16_t*>(smem + (NBUF == 2 ? (kt & 1) : 0) * ST::BUF_BYTES);
        const bf16_t* ldsKh = tb;
        const bf16_t* ldsKl = tb + TILE;                               // split only
        const bf16_t* ldsVh = tb + NPL * TILE;
        const bf16_t* ldsVl = ldsVh + TILE;                            // split only
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int off = l31 * PITCH + 16 * s + h2 * 8;
            const bf16x8 kh = *reinterpret_cast<const bf16x8*>(ldsKh + off);
            if constexpr (SPLIT) {
                const bf16x8 kl = *reinterpret_cast<const bf16x8*>(ldsKl + off);
                sacc = MFMA32(kl, qh[s], sacc);
                sacc = MFMA32(kh, ql[s], sacc);
            }
            sacc = MFMA32(kh, qh[s], sacc);
        }
        // The running maximum lives in log2 units (m_i = max(raw score) * scale * log2 e): one fma + v_exp per probability instead of
        // multiply / subtract / multiply / v_exp, and the key-validity select only on the last (ragged) key tile.
        float sv[16], mloc = -INFINITY;
        if (k0 + 32 > p.N) {                                      // wave-uniform
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (k0 + acc_row(r, h2) >= p.N) sacc[r] = -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sacc[r]);
        mloc = half_max(mloc) * sc2;                              // sc2 > 0
        const float mnew = fmaxf(m_i, mloc);
        const float mold = m_i;
        const float alpha = __builtin_amdgcn_exp2f(m_i - mnew);
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sv[r] = __builtin_amdgcn_exp2f(fmaf(sacc[r], sc2, -mnew));      // masked keys: exp2(-inf) = 0
            lsum += sv[r];
        }
        lsum = half_sum(lsum);
        l_i = l_i * alpha + lsum;
        m_i = mnew;
        if (DROP && p.drop_thr) {                                 // DROP = false: the masked path is compiled out (hd = 256 spilled with it)
            uint32_t wb = 0u;                                     // keep bits of this lane's 16 keys, bit = acc_row(r, 0)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {                        // keys acc_row(r, h2) (even) and + 1: one hash, 16 bits each
                const uint32_t z = drop_hash_at(drow, (uint32_t)(k0 + acc_row(r, h2)));
                const bool keep0 = drop_half(z, 0u, p.drop_thr), keep1 = drop_half(z, 1u, p.drop_thr);
                sv[r] = keep0 ? sv[r] : 0.f;                          // (the 1 / (1 - p) scale is applied once, with 1 / l, at the end)
                sv[r + 1] = keep1 ? sv[r + 1] : 0.f;
                wb |= (keep0 ? (1u << acc_row(r, 0)) : 0u) | (keep1 ? (1u << acc_row(r + 1, 0)) : 0u);
            }
            // S3dAttnArgs::drop_mask: the query's word of tile (qt, kt) = the two half-waves' bits (keys 4 h2 + ..), for the backward kernels
            if (p.drop_mask) {                                    // block-uniform
                wb <<= 4 * h2;
                const auto pr = __builtin_amdgcn_permlane32_swap(wb, wb, false, false);
                if (active && h2 == 0) p.drop_mask[(((long)bh * QT + qt) * KT + kt) * 32 + l31] = pr[0] | pr[1];
            }
        }
        // rescale the running output only when some row's maximum moved (wave-uniform test): after the first few key tiles of a long
        // sequence it almost never does, and alpha == 1 exactly for every row then (6 x 16 multiplies per tile at hd = 192)
        if (__builtin_amdgcn_ballot_w64(mnew != mold) != 0ull) {
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        }
        U128 ph[2], pl[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; j += 2) split_bf16x2(sv[8 * s2 + j], sv[8 * s2 + j + 1], ph[s2].w[j / 2], pl[s2].w[j / 2]);
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
            bf16x8 vh[2], vl[2];
            if constexpr (SPLIT) gather_frag_2x2<HD, PITCH>(ldsVh, d * 32 + l31, ldsVl, d * 32 + l31, h2, vh, vl);
            else gather_frag_s2<HD, PITCH>(ldsVh, h2, d * 32 + l31, vh);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                if constexpr (SPLIT) {
                    o[d] = MFMA32(vl[s2], ph[s2].v, o[d]);
                    o[d] = MFMA32(vh[s2], pl[s2].v, o[d]);
                }
                o[d] = MFMA32(vh[s2], ph[s2].v, o[d]);
            }
        }
        if constexpr (NBUF == 1) __syncthreads();                     // every wave is done reading the only buffer
        if (more) st.lstore(smem + (NBUF == 2 ? ((kt + 1) & 1) : 0) * ST::BUF_BYTES, tid);
        __syncthreads();
    }
    if (active && qok) {
        const float inv = ((DROP && p.drop_thr) ? p.drop_scale : 1.0f) / l_i;
        const long orow = ((long)b * p.sb + (long)qrow * p.st) * p.ldo + h * HD;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                union { uint2 u; bf16_t h[4]; } hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) split_bf16(o[d][4 * c + i] * inv, hi.h[i], lo.h[i]);
                const int dcol = d * 32 + 8 * c + 4 * h2;
                if (HD % 32 != 0 && dcol >= HD) continue;
                const long off = orow + dcol;
                *reinterpret_cast<uint2*>(p.out_hi + off) = hi.u;
                if (p.out_lo) *reinterpret_cast<uint2*>(p.out_lo + off) = lo.u;
            }
        if (h2 == 0 && p.lse) p.lse[(long)bh * p.N + qrow] = m_i * 0.6931471805599453f + logf(l_i);      // m_i is in log2 units
    }
}

// Round 6: the split forward of the long sequences, SOFTWARE-PIPELINED.  attn_fwd_coop_kernel's key tile is S MFMAs -> softmax / mask VALU ->
// P V MFMAs in program order, and at hd = 192 a wave holds 430+ registers (Q hi + lo fragments 96, the output tile 96, staging 48 ...): ONE
// wave per SIMD, so nothing runs beside anything -- per key tile ~2300 MFMA cycles + ~2500 VALU cycles back to back (ISA count: 72 MFMAs,
// 590 VALU instructions; 27 ms per launch at cfg-3).  Here the scores of tile t + 1 are issued INSIDE the softmax of tile t: one branch-free
// region holds 36 independent MFMAs and the ~330 VALU instructions of max / exp2 / sum / dropout hash / mask word / bf16 split, which hipcc's
// scheduler interleaves (the matrix pipe runs while the wave issues VALU).  What that needs:
//   * K and V rings decoupled: K(t + 1) is read during tile t's softmax, V(t) after it -- two K buffers and two V buffers (the same 102 KB),
//     K(t + 2) and V(t + 1) wait in registers, one barrier per tile as before;
//   * no branch between the first score MFMA and the last probability: the ragged last key tile is a peeled instantiation of the body, the
//     mask-word store is unconditional (duplicate lanes / clamped waves store identical words), DROP means drop_thr != 0;
//   * the (rare) rescale of the running output stays a wave-uniform branch, between the region and the P V MFMAs.
// P1: S3dAttnArgs::p_single_plane -- the probabilities as one bf16 plane in the P V product (two MFMAs per product: 60 instead of 72 per key tile)
template <int HD, bool DROP, bool MASKOUT, int ABL = 0, bool P1 = false, bool GA = true>      // ABL: timing ablations (s3d_debug_knob 1; results wrong), 0 = the product; GA = false: round-6a staging addresses (A/B, knob 4)
__global__ __launch_bounds__(256) void attn_fwd_coop_pipe_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NS = HD / 16, NDB = (HD + 31) / 32;
    constexpr int PITCH = HD + 8;
    using ST = CoopStage<HD, 4, 256>;                                  // staging registers: K_hi K_lo (tile t + 2), V_hi V_lo (tile t + 1)
    constexpr int TILE = ST::TILE, PLANE = TILE * 2;                   // bytes of one [32][PITCH] plane
    constexpr int KBUF = 2 * PLANE;                                    // bytes of one K buffer (hi + lo) = of one V buffer
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h2 = lane >> 5, l31 = lane & 31;
    const int QT = (p.N + 31) / 32, QTB = (QT + 3) / 4;
    const int bh = blockIdx.x / QTB;
    int qt = (blockIdx.x % QTB) * 4 + wave;
    const bool active = qt < QT;
    qt = min(qt, QT - 1);
    const int h = bh % p.H, b = bh / p.H;
    const long st_ld = p.st * p.ld;
    const long base = (long)b * p.sb * p.ld + h * HD;
    const int q0 = qt * 32, qrow = q0 + l31;
    const bool qok = qrow < p.N;
    const int qrow_c = min(qrow, p.N - 1);
    const unsigned long long dkey = DROP ? drop_key(p.drop_seed, p.drop_site) : 0ull;
    const DropRow drow = drop_row(dkey, ((unsigned long long)bh * p.N + qrow_c) * p.N);
    bf16x8 qh[NS], ql[NS];
    {
        const long off = base + (long)qrow_c * st_ld + h2 * 8;
#pragma unroll
        for (int s = 0; s < NS; ++s) { qh[s] = ld_frag(p.qkv_hi + off + 16 * s); ql[s] = ld_frag(p.qkv_lo + off + 16 * s); }
    }
    f32x16 o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_i = -INFINITY, l_i = 0.f;
    const float sc2 = p.scale * 1.4426950408889634f;
    const int KT = (p.N + 31) / 32;
    // LDS: [K ring: 2 x (hi, lo)] [V ring: 2 x (hi, lo)]
    unsigned char* kring = smem;
    unsigned char* vring = smem + 2 * KBUF;
    const long koff = base + (long)p.D, voff = base + 2 * (long)p.D;

    // staging: chunk c of a [32][HD] tile -> (row, 16-byte column); K planes from key tile tk, V planes from key tile tv.  (LDS-DMA instead of
    // registers was built and measured: 23.3 ms per cfg-3 launch with the thirteen 1 KB pieces per wave issued together, 24.5 ms with one
    // piece behind the first MFMA of every k-step, against 22.4 ms here -- a wave sits in the issue stage for every piece, MFMA shadow or not.)
    u32x4 rk[2][ST::NCH], rv[2][ST::NCH];
    // a whole tile inside the sequence: wave-uniform plane bases (SGPR pairs) + one 32-bit byte offset per chunk (row in the tile, 16-byte
    // column) that never changes -- the twelve 64-bit row addresses per tile (min, two multiplies, a 64-bit mad and two 64-bit adds each:
    // ~70 VALU instructions of an in-order wave's key-tile iteration, and 24 registers of address pairs) are only built for a ragged tile
    unsigned coff[ST::NCH], coff_last[ST::NCH];                        // (coff_last: rows of the ragged last tile clamped to the last key)
    const int KT_ = (p.N + 31) / 32;
#pragma unroll
    for (int i = 0; i < ST::NCH; ++i) {
        const int c = min(tid + 256 * i, 32 * ST::CPR - 1);
        const int row = c / ST::CPR, cc = c % ST::CPR;
        coff[i] = (unsigned)(((long)row * st_ld + cc * 8) * 2);
        coff_last[i] = (unsigned)(((long)min(row, p.N - 1 - (KT_ - 1) * 32) * st_ld + cc * 8) * 2);
    }
    const char* const gkh = reinterpret_cast<const char*>(p.qkv_hi + koff), *const gkl = reinterpret_cast<const char*>(p.qkv_lo + koff);
    const char* const gvh = reinterpret_cast<const char*>(p.qkv_hi + voff), *const gvl = reinterpret_cast<const char*>(p.qkv_lo + voff);
    const long tile_bytes = 64 * st_ld;                                // 32 rows x st_ld elements x 2 bytes
    auto gload = [&](int tk, int tv) {                                 // (tiles past the end -- the prefetch overshoots by up to three -- read the last one)
        if constexpr (!GA) {
#pragma unroll
            for (int i = 0; i < ST::NCH; ++i) {
                const int c = min(tid + 256 * i, 32 * ST::CPR - 1);
                const int row = c / ST::CPR, cc = c % ST::CPR;
                const long rowk = min(tk * 32 + row, p.N - 1), rowv = min(tv * 32 + row, p.N - 1);
                rk[0][i] = *reinterpret_cast<const u32x4*>(p.qkv_hi + koff + rowk * st_ld + cc * 8);
                rk[1][i] = *reinterpret_cast<const u32x4*>(p.qkv_lo + koff + rowk * st_ld + cc * 8);
                rv[0][i] = *reinterpret_cast<const u32x4*>(p.qkv_hi + voff + rowv * st_ld + cc * 8);
                rv[1][i] = *reinterpret_cast<const u32x4*>(p.qkv_lo + voff + rowv * st_ld + cc * 8);
            }
            return;
        }
        const long kb = min(tk, KT_ - 1) * tile_bytes, vb = min(tv, KT_ - 1) * tile_bytes;
        const bool klast = tk >= KT_ - 1, vlast = tv >= KT_ - 1;       // block-uniform
#pragma unroll
        for (int i = 0; i < ST::NCH; ++i) {
            const unsigned ko = klast ? coff_last[i] : coff[i], vo = vlast ? coff_last[i] : coff[i];
            rk[0][i] = *reinterpret_cast<const u32x4*>(gkh + kb + ko);
            rk[1][i] = *reinterpret_cast<const u32x4*>(gkl + kb + ko);
            rv[0][i] = *reinterpret_cast<const u32x4*>(gvh + vb + vo);
            rv[1][i] = *reinterpret_cast<const u32x4*>(gvl + vb + vo);
        }
    };
    auto lstore = [&](const u32x4 (&r)[2][ST::NCH], unsigned char* buf) {
        bf16_t* bp = reinterpret_cast<bf16_t*>(buf);
#pragma unroll
        for (int i = 0; i < ST::NCH; ++i) {
            const int c = tid + 256 * i;
            if (32 * ST::CPR % 256 == 0 || c < 32 * ST::CPR) {
                const int row = c / ST::CPR, cc = c % ST::CPR;
                *reinterpret_cast<u32x4*>(bp + row * PITCH + cc * 8) = r[0][i];
                *reinterpret_cast<u32x4*>(bp + TILE + row * PITCH + cc * 8) = r[1][i];
            }
        }
    };
    auto scores = [&](const unsigned char* kb, f32x16& acc) {           // S^T = K Q^T, three MFMAs per product
        const bf16_t* ldsKh = reinterpret_cast<const bf16_t*>(kb);
        const bf16_t* ldsKl = ldsKh + PLANE / 2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int off = l31 * PITCH + 16 * s + h2 * 8;
            const bf16x8 kh = *reinterpret_cast<const bf16x8*>(ldsKh + off);
            const bf16x8 kl = *reinterpret_cast<const bf16x8*>(ldsKl + off);
            acc = MFMA32(kl, qh[s], acc);
            acc = MFMA32(kh, ql[s], acc);
            acc = MFMA32(kh, qh[s], acc);
        }
    };

    // prologue: K(0), V(0), K(1) in LDS; S(0) in registers; K(2), V(1) on their way
    gload(0, 0);
    lstore(rk, kring); lstore(rv, vring);
    gload(1, 1);
    lstore(rk, kring + KBUF);
    __syncthreads();
    f32x16 scur, snext;
    scores(kring, scur);
    // K(2) goes into the buffer S(0) was just read from, at the END of iteration 0 -- without this barrier a wave that is a whole key tile behind
    // scores tile 0 against rows of K(2).  It happens: the Q fragments are first used above, and in the first workgroups of a launch (cold TLB; a
    // wave's 32 query rows are 32 pages of their own) one wave's Q data arrives microseconds after its neighbours'.  Measured at cfg-3 before
    // the barrier was here: one launch in ~40 with 32 - 96 rows off by 3e-5 - 1e-4 in the log-sum-exp, always key tile 0, always the k-steps at
    // the end of the tile, in blocks < 400 of 5880 (tools/r6/attn_repro_stress.py, attn_repro_diag.py; profiles/r06_attn_prologue_race.txt).
    __syncthreads();
    gload(2, 1);
    uint32_t zz[8], zh[2];                                             // dropout hashes of the CURRENT tile's key pairs (computed one tile ahead)
    if constexpr (DROP) {
#pragma unroll
        for (int j = 0; j < 8; ++j) zz[j] = drop_hash_at(drow, (uint32_t)acc_row(2 * j, h2));
    }

    auto body = [&](int kt, auto ragged_tag, auto more_tag) {
        constexpr bool RAGGED = decltype(ragged_tag)::value, MORE = decltype(more_tag)::value;
        const int k0 = kt * 32;
        // ---- region 1 (branch-free): the 36 score MFMAs of tile kt + 1, each followed by ONE slot of tile kt's softmax / mask work.
        // hipcc's scheduler does not interleave the two by itself (it clusters the MFMAs in front of the VALU, and an in-order wave
        // sits out every MFMA's 8 passes before the next one issues -- measured, also with sched_group_barrier templates), so the order
        // is written out and pinned with sched_barrier(0): MFMA, ~8 VALU instructions (= its 32 cycles), MFMA, ...
        float sv[16], mloc = -INFINITY, mnew = m_i, alpha = 1.f, lsum = 0.f, nmn = 0.f;
        const float mold = m_i;
        uint32_t wb = 0u;
        U128 ph[2], pl[2];
        if (ABL == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { ph[0].w[q] = ph[1].w[q] = 0x3f803f80u; pl[0].w[q] = pl[1].w[q] = 0u; }
        }
        if constexpr (RAGGED) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (k0 + acc_row(r, h2) >= p.N) scur[r] = -INFINITY;
        }
        auto slot = [&](auto jt) {
            constexpr int J = decltype(jt)::value;
            if constexpr (J < 2) {                                    // row maximum, two halves
                constexpr int r0 = J * 8;
#pragma unroll
                for (int r = r0; r < r0 + 8; ++r) mloc = fmaxf(mloc, scur[r]);
            } else if constexpr (J == 2) {
                mloc = half_max(mloc) * sc2;
                mnew = fmaxf(m_i, mloc);
                alpha = __builtin_amdgcn_exp2f(m_i - mnew);
                nmn = -mnew;
            } else if constexpr (J < 19) {                            // one probability per slot
                constexpr int r0 = J - 3;
                sv[r0] = __builtin_amdgcn_exp2f(fmaf(scur[r0], sc2, nmn));
                lsum += sv[r0];
            } else if constexpr (J == 19) {
                lsum = half_sum(lsum);
                l_i = l_i * alpha + lsum;
                m_i = mnew;
            } else if constexpr (J < 28) {                            // keep decisions of pair J - 20 (hash zz[] computed beside the previous
                if constexpr (DROP) {                                 // tile's P V MFMAs): probabilities, mask bits
                    constexpr int r = (J - 20) * 2;
                    const bool keep0 = drop_half(zz[J - 20], 0u, p.drop_thr), keep1 = drop_half(zz[J - 20], 1u, p.drop_thr);
                    sv[r] = keep0 ? sv[r] : 0.f;
                    sv[r + 1] = keep1 ? sv[r + 1] : 0.f;
                    wb |= (keep0 ? (1u << acc_row(r, 0)) : 0u) | (keep1 ? (1u << acc_row(r + 1, 0)) : 0u);
                }
            } else {                                                  // J = 28 .. 35: bf16 hi / lo of pair J - 28
                constexpr int q = J - 28, s2 = q / 4, j = (q % 4) * 2;
                if constexpr (P1) ph[s2].w[j / 2] = f2bf2(sv[8 * s2 + j], sv[8 * s2 + j + 1]);
                else split_bf16x2(sv[8 * s2 + j], sv[8 * s2 + j + 1], ph[s2].w[j / 2], pl[s2].w[j / 2]);
            }
        };
        if constexpr (MORE) {
            const bf16_t* ldsKh = reinterpret_cast<const bf16_t*>(kring + ((kt + 1) & 1) * KBUF);
            const bf16_t* ldsKl = ldsKh + PLANE / 2;
#pragma unroll
            for (int r = 0; r < 16; ++r) snext[r] = 0.f;
            bf16x8 kfh[2], kfl[2];                                     // K fragments of k-step S in [S & 1]: read one step ahead
            kfh[0] = *reinterpret_cast<const bf16x8*>(ldsKh + l31 * PITCH + h2 * 8);
            kfl[0] = *reinterpret_cast<const bf16x8*>(ldsKl + l31 * PITCH + h2 * 8);
            auto kstep = [&](auto st) {
                constexpr int S = decltype(st)::value;
                if constexpr (S + 1 < NS) {                           // the next step's fragments are in flight while this step's MFMAs run
                    const int off = l31 * PITCH + 16 * (S + 1) + h2 * 8;
                    kfh[(S + 1) & 1] = *reinterpret_cast<const bf16x8*>(ldsKh + off);
                    kfl[(S + 1) & 1] = *reinterpret_cast<const bf16x8*>(ldsKl + off);
                }
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 kh = kfh[S & 1], kl = kfl[S & 1];
                if (ABL != 3) snext = MFMA32(kl, qh[S], snext);
                __builtin_amdgcn_sched_barrier(0);
                if (ABL != 1) slot(std::integral_constant<int, 3 * S>{});
                __builtin_amdgcn_sched_barrier(0);
                if (ABL != 3) snext = MFMA32(kh, ql[S], snext);
                __builtin_amdgcn_sched_barrier(0);
                if (ABL != 1) slot(std::integral_constant<int, 3 * S + 1>{});
                __builtin_amdgcn_sched_barrier(0);
                if (ABL != 3) snext = MFMA32(kh, qh[S], snext);
                __builtin_amdgcn_sched_barrier(0);
                if (ABL != 1) slot(std::integral_constant<int, 3 * S + 2>{});
                __builtin_amdgcn_sched_barrier(0);
            };
            kstep(std::integral_constant<int, 0>{}); kstep(std::integral_constant<int, 1>{}); kstep(std::integral_constant<int, 2>{});
            kstep(std::integral_constant<int, 3>{}); kstep(std::integral_constant<int, 4>{}); kstep(std::integral_constant<int, 5>{});
            kstep(std::integral_constant<int, 6>{}); kstep(std::integral_constant<int, 7>{}); kstep(std::integral_constant<int, 8>{});
            kstep(std::integral_constant<int, 9>{}); kstep(std::integral_constant<int, 10>{}); kstep(std::integral_constant<int, 11>{});
            static_assert(NS == 12, "the slot table is written for hd = 192 (12 k-steps of three MFMAs)");
        } else {
            [&]<int... Js>(std::integer_sequence<int, Js...>) { (slot(std::integral_constant<int, Js>{}), ...); }(std::make_integer_sequence<int, 36>{});
        }
        if constexpr (DROP && MASKOUT) {
            wb <<= 4 * h2;
            const auto pr = __builtin_amdgcn_permlane32_swap(wb, wb, false, false);
            // every lane stores: the two half-waves (and a clamped, inactive wave) write the SAME word to the same address
            p.drop_mask[(((long)bh * QT + qt) * KT + kt) * 32 + l31] = pr[0] | pr[1];
        }
        // ---- rescale of the running output (wave-uniform, rare after the first key tiles)
        if (__builtin_amdgcn_ballot_w64(mnew != mold) != 0ull) {
            asm volatile("" ::: "memory");                          // (keeps the branch: hipcc otherwise speculates the 96 multiplies -- through
            //                                                         AGPR reads and writes, 240 instructions -- into every tile)
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        }
        // ---- region 2: O^T += V^T P^T
        const bf16_t* ldsVh = reinterpret_cast<const bf16_t*>(vring + (kt & 1) * KBUF);
        const bf16_t* ldsVl = ldsVh + PLANE / 2;
        Frag2x2 vf[2];                                                 // V fragments of d-block d in [d & 1]: read one block ahead
        gather_issue_2x2<HD, PITCH>(ldsVh, l31, ldsVl, l31, h2, vf[0]);
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
            bf16x8 vh[2], vl[2];
            gather_wait(vf[d & 1], vh, vl);
            if (d + 1 < NDB) gather_issue_2x2<HD, PITCH>(ldsVh, (d + 1) * 32 + l31, ldsVl, (d + 1) * 32 + l31, h2, vf[(d + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);                        // (the six MFMAs stay between this block's reads and the next wait)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                // the dropout hash of the NEXT tile's key pair 2 d + s2 in the shadow of this block's MFMAs (it depends on nothing here):
                // drop_hash_at() in two halves, one behind each of the first two MFMAs
                const int hj = 2 * d + s2;
                if (ABL != 2) o[d] = MFMA32(vl[s2], ph[s2].v, o[d]);
                __builtin_amdgcn_sched_barrier(0);
                if (DROP && MORE && hj < 8) {
                    const uint32_t lo = drow.lo + (uint32_t)(k0 + 32 + acc_row(2 * hj, h2));
                    uint32_t x = (lo ^ drow.kx) + (lo < drow.lo ? drow.mix_b : drow.mix_a);
                    x ^= x >> 16; x *= 0x21f0aaadu;
                    zh[s2] = x;
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!P1) { if (ABL != 2) o[d] = MFMA32(vh[s2], pl[s2].v, o[d]); }
                __builtin_amdgcn_sched_barrier(0);
                if (DROP && MORE && hj < 8) {
                    uint32_t x = zh[s2];
                    x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15;
                    zz[hj] = x;
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ABL != 2) o[d] = MFMA32(vh[s2], ph[s2].v, o[d]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MORE) {
            // K(kt + 2) -> the K buffer tile kt used (read last in iteration kt - 1); V(kt + 1) -> the V buffer tile kt - 1 used
            if (ABL != 4) {
                lstore(rk, kring + (kt & 1) * KBUF);
                lstore(rv, vring + ((kt + 1) & 1) * KBUF);
                gload(ABL == 5 ? 0 : kt + 3, ABL == 5 ? 0 : kt + 2);      // (ABL 5: the staging mechanism on cache-hot tiles)
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) scur[r] = snext[r];
        }
        __syncthreads();
    };
    const bool ragged = (p.N & 31) != 0;
    for (int kt = 0; kt + 1 < KT; ++kt) body(kt, std::false_type{}, std::true_type{});
    if (ragged) body(KT - 1, std::true_type{}, std::false_type{});
    else body(KT - 1, std::false_type{}, std::false_type{});

    if (active && qok) {
        const float inv = (DROP ? p.drop_scale : 1.0f) / l_i;
        const long orow = ((long)b * p.sb + (long)qrow * p.st) * p.ldo + h * HD;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                union { uint2 u; bf16_t h[4]; } hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) split_bf16(o[d][4 * c + i] * inv, hi.h[i], lo.h[i]);
                const int dcol = d * 32 + 8 * c + 4 * h2;
                if (HD % 32 != 0 && dcol >= HD) continue;
                const long off = orow + dcol;
                *reinterpret_cast<uint2*>(p.out_hi + off) = hi.u;
                if (p.out_lo) *reinterpret_cast<uint2*>(p.out_lo + off) = lo.u;
            }
        if (h2 == 0 && p.lse) p.lse[(long)bh * p.N + qrow] = m_i * 0.6931471805599453f + logf(l_i);
    }
}

template <int HD, int NWV = 4, bool MASK = false>                   // MASK: see attn_bwd_dkv_coop_kernel
__global__ __launch_bounds__(64 * NWV) void attn_bwd_dq_coop_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NS = HD / 16, NDB = (HD + 31) / 32;
    constexpr bool GA = HD >= 192;                                     // uniform-base staging addresses (CoopStageU)
    using ST = std::conditional_t<GA, CoopStageU<HD, 2, 64 * NWV>, CoopStage<HD, 2, 64 * NWV>>;        // K, V
    constexpr int PITCH = ST::PITCH, TILE = ST::TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h2 = lane >> 5, l31 = lane & 31;

    const int QT = (p.N + 31) / 32, QTB = (QT + NWV - 1) / NWV;
    const int bh = blockIdx.x / QTB;
    int qt = (blockIdx.x % QTB) * NWV + wave;
    const bool active = qt < QT;
    qt = min(qt, QT - 1);
    const int h = bh % p.H, b = bh / p.H;
    const long st_ld = p.st * p.ld;
    const long base = (long)b * p.sb * p.ld + h * HD;
    const int q0 = qt * 32, qrow = q0 + l31;
    const bool qok = qrow < p.N;
    const int qrow_c = min(qrow, p.N - 1);
    const long tokrow = (long)b * p.sb + (long)qrow_c * p.st;
    const unsigned long long dkey = p.drop_thr ? drop_key(p.drop_seed, p.drop_site) : 0ull;
    const DropRow drow = drop_row(dkey, ((unsigned long long)bh * p.N + qrow_c) * p.N);     // mask index = row base + key column

    bf16x8 qf[NS], dof[NS];
    float delta = 0.f;
    {
        const long off = base + (long)qrow_c * st_ld + h2 * 8;
        const long doff = tokrow * p.lddo + h * HD + h2 * 8;
        const long ooff = tokrow * p.ldo + h * HD + h2 * 8;
        const float lo_on = p.out_lo ? 1.f : 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            qf[s] = ld_frag(p.qkv_hi + off + 16 * s);
            dof[s] = ld_frag(p.dout + doff + 16 * s);
            U128 a, ol, d;
            a.v = ld_frag(p.out_hi + ooff + 16 * s);
            ol.v = ld_frag((p.out_lo ? p.out_lo : p.out_hi) + ooff + 16 * s);
            d.v = dof[s];
#pragma unroll
            for (int j = 0; j < 8; ++j) delta += bf2f(d.h[j]) * (bf2f(a.h[j]) + lo_on * bf2f(ol.h[j]));
        }
        delta = half_sum(delta);
    }
    const float lse_q = p.lse[(long)bh * p.N + qrow_c];
    if (active && qok && h2 == 0) p.delta[(long)bh * p.N + qrow] = delta;

    f32x16 dq[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;

    ST st;
    const bf16_t* src[2] = {p.qkv_hi, p.qkv_hi};
    long rowoff[2] = {base + p.D, base + 2 * (long)p.D}, pitch[2] = {st_ld, st_ld};
    const int KT = (p.N + 31) / 32;
    // S3dAttnArgs::drop_mask: this query's word of tile (qt, kt), fetched MD key tiles ahead into a ring of MR registers.  One tile ahead (rounds
    // 5 - 6) left the wave waiting: the word is a first touch of its cache line every time, a key-tile iteration lasts ~1.4 us at hd = 192, and the
    // K / V loads of the next tile queue up behind it (a wave's loads return in order) -- see attn_bwd_dkv_coop_kernel's DEEP.  The ring needs
    // static indices: the loop is unrolled MR times (tags).
    constexpr bool DEEP = MASK && HD >= 192 && NWV == 4;
    constexpr int MR = DEEP ? 4 : 1, MD = DEEP ? 3 : 1;
    const unsigned int* mrow = MASK ? p.drop_mask + ((long)bh * QT + qt) * KT * 32 + l31 : nullptr;
    uint32_t wring[MR] = {};
    if constexpr (MASK) {
#pragma unroll
        for (int j = 0; j < MD; ++j) if (j < KT) wring[j % MR] = mrow[(long)j * 32];
    }
    if constexpr (GA) { st.init(src, rowoff, pitch, p.N, tid); st.gload_tile(0); }
    else st.gload(src, rowoff, pitch, 0, p.N, tid);
    st.lstore(smem, tid);
    __syncthreads();
    auto body = [&](int kt, auto ring_tag) {
        constexpr int J = decltype(ring_tag)::value;
        const int k0 = kt * 32;
        const bool more = kt + 1 < KT;
        if (more) { if constexpr (GA) st.gload_tile(kt + 1); else st.gload(src, rowoff, pitch, k0 + 32, p.N, tid); }
        const uint32_t wcur = wring[J] >> (4 * h2);                    // bit acc_row(r, 0) = key acc_row(r, h2)
        if constexpr (MASK) { if (kt + MD < KT) wring[(J + MD) % MR] = mrow[(long)(kt + MD) * 32]; }   // (behind the tile's loads: they are stored first)
        const bf16_t* ldsK = reinterpret_cast<const bf16_t*>(smem + (kt & 1) * ST::BUF_BYTES);
        const bf16_t* ldsV = ldsK + TILE;
        const bool ragged = k0 + 32 > p.N;
        const float sc2 = p.scale * 1.4426950408889634f, lse2 = lse_q * 1.4426950408889634f;
        f32x16 sacc, dpacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
        if constexpr (HD >= 192) {   // fragments of k-step s + 1 are read before the MFMAs of step s (round 6: hipcc waits for every pair right in
            // front of its MFMAs, and one wave per SIMD has nothing else to run during an LDS round trip -- 2 x 12 of them per key tile; the
            // hd = 64 kernels of the point path run three / two waves per SIMD and keep their register counts)
            bf16x8 kf[2], vf2[2];
            kf[0] = *reinterpret_cast<const bf16x8*>(ldsK + l31 * PITCH + h2 * 8);
            vf2[0] = *reinterpret_cast<const bf16x8*>(ldsV + l31 * PITCH + h2 * 8);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s + 1 < NS) {
                    const int off = l31 * PITCH + 16 * (s + 1) + h2 * 8;
                    kf[(s + 1) & 1] = *reinterpret_cast<const bf16x8*>(ldsK + off);
                    vf2[(s + 1) & 1] = *reinterpret_cast<const bf16x8*>(ldsV + off);
                }
                __builtin_amdgcn_sched_barrier(0);
                sacc = MFMA32(kf[s & 1], qf[s], sacc);                                        // S^T  = K . Q^T
                dpacc = MFMA32(vf2[s & 1], dof[s], dpacc);                                    // dP^T = V . dO^T
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int off = l31 * PITCH + 16 * s + h2 * 8;
                sacc = MFMA32(*reinterpret_cast<const bf16x8*>(ldsK + off), qf[s], sacc);
                dpacc = MFMA32(*reinterpret_cast<const bf16x8*>(ldsV + off), dof[s], dpacc);
            }
        }
        U128 dsf[2];
        auto ds_tile = [&](auto from_mask) {                           // the mask from the stored bits (block-uniform choice) or from the hash
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    float dsv[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int r = 8 * s2 + j + e;
                        float pr = __builtin_amdgcn_exp2f(fmaf(sacc[r], sc2, -lse2));        // exp(S * scale - lse), one fma + v_exp
                        if (ragged) pr = (k0 + acc_row(r, h2)) < p.N ? pr : 0.f;             // last key tile only (wave-uniform)
                        float dpn = dpacc[r];
                        if constexpr (decltype(from_mask)::value)     // v_bfe_i32 (0 / ~0) & scale bits: three instructions instead of five
                            dpn *= __uint_as_float((uint32_t)__builtin_amdgcn_sbfe((int)wcur, acc_row(r, 0), 1) & __float_as_uint(p.drop_scale));
                        else if (p.drop_thr) dpn = drop_half(drop_hash_at(drow, (uint32_t)(k0 + (acc_row(r, h2) & ~1))), (uint32_t)r & 1u, p.drop_thr) ? dpn * p.drop_scale : 0.f;
                        dsv[e] = pr * (dpn - delta) * p.scale;
                    }
                    dsf[s2].w[j / 2] = f2bf2(dsv[0], dsv[1]);
                }
        };
        ds_tile(std::integral_constant<bool, MASK>{});
        if constexpr (NDB % 2 == 0 && HD % 32 == 0 && HD >= 192) {
            // dQ^T = K^T . dS^T, two d-blocks per step; the transposed K fragments of the NEXT pair are in flight during this pair's MFMAs
            Frag2x2 kq[2];
            gather_issue_2x2<HD, PITCH>(ldsK, l31, ldsK, 32 + l31, h2, kq[0]);
#pragma unroll
            for (int d = 0; d < NDB; d += 2) {
                bf16x8 k0[2], k1[2];
                gather_wait(kq[(d >> 1) & 1], k0, k1);
                if (d + 2 < NDB) gather_issue_2x2<HD, PITCH>(ldsK, (d + 2) * 32 + l31, ldsK, (d + 3) * 32 + l31, h2, kq[((d >> 1) + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    dq[d] = MFMA32(k0[s2], dsf[s2].v, dq[d]);
                    dq[d + 1] = MFMA32(k1[s2], dsf[s2].v, dq[d + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int d = 0; d < NDB; d += 2) {
                bf16x8 k0[2], k1[2];
                if (d + 1 < NDB) gather_frag_2x2<HD, PITCH>(ldsK, d * 32 + l31, ldsK, (d + 1) * 32 + l31, h2, k0, k1);
                else gather_frag_s2<HD, PITCH>(ldsK, h2, d * 32 + l31, k0);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    dq[d] = MFMA32(k0[s2], dsf[s2].v, dq[d]);                                 // dQ^T = K^T . dS^T
                    if (d + 1 < NDB) dq[d + 1] = MFMA32(k1[s2], dsf[s2].v, dq[d + 1]);
                }
            }
        }
        if (more) st.lstore(smem + ((kt + 1) & 1) * ST::BUF_BYTES, tid);
        __syncthreads();
    };
    {
        int kt = 0;
        if constexpr (MR == 4) {
            for (; kt + 3 < KT; kt += 4) {
                body(kt, std::integral_constant<int, 0>{}); body(kt + 1, std::integral_constant<int, 1>{});
                body(kt + 2, std::integral_constant<int, 2>{}); body(kt + 3, std::integral_constant<int, 3>{});
            }
            if (kt < KT) body(kt, std::integral_constant<int, 0>{});
            if (kt + 1 < KT) body(kt + 1, std::integral_constant<int, 1>{});
            if (kt + 2 < KT) body(kt + 2, std::integral_constant<int, 2>{});
        } else {
            for (; kt < KT; ++kt) body(kt, std::integral_constant<int, 0>{});
        }
    }
    if (active && qok) {
        const long orow = tokrow * p.lddq + h * HD;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                union { uint2 u; bf16_t h[4]; } v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v.h[i] = f2bf(dq[d][4 * c + i]);
                const int dcol = d * 32 + 8 * c + 4 * h2;
                if (HD % 32 != 0 && dcol >= HD) continue;
                *reinterpret_cast<uint2*>(p.dqkv + orow + dcol) = v.u;
            }
    }
}

// ------------------------------------------------------------------------------------------- backward, N <= 32: one launch
// With a single 32-token tile per (batch, head) -- cfg-1/2: N = 26 -- the two backward kernels above are each a few
// microseconds of launch ramp and memory latency around a handful of MFMAs.  Here one wave does both for its (batch, head):
// phase A = the dQ kernel's body (lane = query; also yields delta), phase B = the dK/dV kernel's body (lane = key), with
// delta / lse handed over through LDS instead of a global round trip.
constexpr int small_bwd_wave_lds(int HD) { return 3 * 32 * (HD + 8) * 2 + 256; }

template <int HD, bool SEG = false>
__global__ __launch_bounds__(256) void attn_bwd_small_kernel(const AttnArgs p, const AdamFill fill) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    {   // workgroups behind the main grid run a share of the optimizer update (adam_fill.h)
        const int nmain = (int)gridDim.x - fill.blocks;
        if ((int)blockIdx.x >= nmain) { adam_fill_run(fill, (int)blockIdx.x - nmain); return; }
    }
    constexpr int NS = HD / 16, NDB = (HD + 31) / 32;
    // hd = 192 / 256 (cfg-3's 15-token groups, 37 632 of them per block): the row fragments alone are 192 - 256 registers, so the
    // gradients are accumulated two d-blocks at a time; three waves (49 KB of tiles each) share a CU
    constexpr int NDBC = NDB <= 3 ? NDB : 2;
    static_assert(NDB % NDBC == 0, "d-blocks split evenly");
    // tile rows are padded by 16 bytes: the row fragments below are 16-byte reads of 32 different rows (a pitch of HD * 2 bytes would
    // put them all on the same banks), and the transposed reads of the gradient GEMMs stay conflict-free
    constexpr int PITCH = HD + 8;
    constexpr int WAVE_LDS = small_bwd_wave_lds(HD);                   // K | Q | dO tiles (bf16) + delta / lse (fp32)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h2 = lane >> 5, l31 = lane & 31;
    bf16_t* ldsK = reinterpret_cast<bf16_t*>(smem + wave * WAVE_LDS);
    bf16_t* ldsQ = ldsK + 32 * PITCH;
    bf16_t* ldsDO = ldsQ + 32 * PITCH;
    float* ldsR = reinterpret_cast<float*>(ldsDO + 32 * PITCH);        // [0..31] delta, [32..63] lse

    const long W = (long)p.Bb * p.H;
    long item = (long)blockIdx.x * (blockDim.x >> 6) + wave;
    const bool active = item < W;
    if (!active) item = W - 1;
    const int bh = (int)item;
    const int h = bh % p.H, b = bh / p.H;
    const long st_ld = p.st * p.ld;
    const long base = (long)b * p.sb * p.ld + h * HD;
    const long dobase = (long)b * p.sb * p.lddo + h * HD;
    const long st_lddo = p.st * p.lddo;
    const bool tok_ok = l31 < p.N;                                     // this lane's token (query in A, key in B)
    const int tok = min(l31, p.N - 1);
    const long tokrow = (long)b * p.sb + (long)tok * p.st;
    const unsigned long long dkey = p.drop_thr ? drop_key(p.drop_seed, p.drop_site) : 0ull;   // the seed lives on the device

    // row-fragments of this lane's token: Q, dO (phase A operands), K, V (phase B operands), all out of tiles staged with coalesced
    // loads (a fragment gathered straight from memory -- every lane its own row, 32 rows per load instruction -- is what the
    // texture-address path serves slowest: the six gathers per 16 features this kernel used to do cost more than its arithmetic,
    // DESIGN.md section 6).  V is only ever needed as row fragments: it passes through the K tile's space first (the LDS operations of
    // a wave stay in order).
    bf16x8 qf[NS], dof[NS], kf[NS], vf[NS];
    const int lo = l31 * PITCH + h2 * 8;
    stage_tile<HD, PITCH>(ldsK, p.qkv_hi, p.ld, base + 2 * p.D, st_ld, 0, p.N, lane);
    stage_tile<HD, PITCH>(ldsQ, p.qkv_hi, p.ld, base, st_ld, 0, p.N, lane);
    stage_tile<HD, PITCH>(ldsDO, p.dout, p.lddo, dobase, st_lddo, 0, p.N, lane);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        U128 t;
        t.u = *reinterpret_cast<const u32x4*>(ldsK + lo + 16 * s); vf[s] = t.v;
    }
    __builtin_amdgcn_wave_barrier();
    stage_tile<HD, PITCH>(ldsK, p.qkv_hi, p.ld, base + p.D, st_ld, 0, p.N, lane);
    __builtin_amdgcn_wave_barrier();
    {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            U128 t;
            t.u = *reinterpret_cast<const u32x4*>(ldsQ + lo + 16 * s); qf[s] = t.v;
            t.u = *reinterpret_cast<const u32x4*>(ldsK + lo + 16 * s); kf[s] = t.v;
            t.u = *reinterpret_cast<const u32x4*>(ldsDO + lo + 16 * s); dof[s] = t.v;
        }
    }
    const float lse_q = p.lse[(long)bh * p.N + tok];
    if (h2 == 0) ldsR[32 + l31] = lse_q;
    __syncthreads();                                                   // tiles + lse are staged

    // ---- phase A: lane = query.  S^T = K . Q^T, dP^T = V . dO^T, dQ^T = K^T . dS^T
    {
        f32x16 sacc, dpacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            sacc = MFMA32(kf[s], qf[s], sacc);
            dpacc = MFMA32(vf[s], dof[s], dpacc);
        }
        // delta[q] = sum_k P[q][k] dP[q][k] (= rowsum(dO * O) in exact arithmetic, with the dropout mask on both): no read of the
        // forward's output, and the rows of dS sum to zero exactly for the P this backward recomputes
        float pq[16], delta = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = acc_row(r, h2);
            pq[r] = (key < p.N && seg_ok<SEG>(p, tok, key)) ? fast_exp(sacc[r] * p.scale - lse_q) : 0.f;
            if (p.drop_thr)
                dpacc[r] = drop_keep_attn(dkey, ((unsigned long long)bh * p.N + tok) * p.N, (uint32_t)min(key, p.N - 1), p.drop_thr) ? dpacc[r] * p.drop_scale : 0.f;
            delta += pq[r] * dpacc[r];
        }
        delta = half_sum(delta);
        if (h2 == 0) ldsR[l31] = delta;                                // read by this wave's phase B (LDS operations of a wave stay in order)
        __builtin_amdgcn_wave_barrier();
        if (active && tok_ok && h2 == 0 && p.delta) p.delta[(long)bh * p.N + l31] = delta;
        U128 dsf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 8 * s2 + j;
                dsf[s2].h[j] = f2bf(pq[r] * (dpacc[r] - delta) * p.scale);
            }
        const long orow = tokrow * p.lddq + h * HD;
#pragma unroll
        for (int d0 = 0; d0 < NDB; d0 += NDBC) {                       // NDBC d-blocks of dQ at a time (all of them for hd <= 96)
            f32x16 dq[NDBC];
#pragma unroll
            for (int d = 0; d < NDBC; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;
#pragma unroll
            for (int d = 0; d < NDBC; d += 2) {
                bf16x8 k0f[2], k1f[2];
                if (d + 1 < NDBC) gather_frag_2x2<HD, PITCH>(ldsK, (d0 + d) * 32 + l31, ldsK, (d0 + d + 1) * 32 + l31, h2, k0f, k1f);
                else gather_frag_s2<HD, PITCH>(ldsK, h2, (d0 + d) * 32 + l31, k0f);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    dq[d] = MFMA32(k0f[s2], dsf[s2].v, dq[d]);
                    if (d + 1 < NDBC) dq[d + 1] = MFMA32(k1f[s2], dsf[s2].v, dq[d + 1]);
                }
            }
            if (active && tok_ok) {
#pragma unroll
                for (int d = 0; d < NDBC; ++d)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        union { uint2 u; bf16_t h[4]; } v;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v.h[i] = f2bf(dq[d][4 * c + i]);
                        const int dcol = (d0 + d) * 32 + 8 * c + 4 * h2;
                        if (HD % 32 != 0 && dcol >= HD) continue;
                        *reinterpret_cast<uint2*>(p.dqkv + orow + dcol) = v.u;
                    }
            }
        }
    }
    // ---- phase B: lane = key.  S = Q . K^T, dP = dO . V^T, dV^T = dO^T . P, dK^T = Q^T . dS
    {
        f32x16 sacc, dpacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            sacc = MFMA32(qf[s], kf[s], sacc);
            dpacc = MFMA32(dof[s], vf[s], dpacc);
        }
        U128 pf[2], dsf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 8 * s2 + j, q = acc_row(r, h2);
                const bool ok = tok_ok && (q < p.N) && seg_ok<SEG>(p, q, tok);
                const int qc = min(q, p.N - 1);
                const float pr = ok ? fast_exp(sacc[r] * p.scale - ldsR[32 + qc]) : 0.f;
                float dm = 1.f;
                if (p.drop_thr) dm = drop_keep_attn(dkey, ((unsigned long long)bh * p.N + qc) * p.N, (uint32_t)tok, p.drop_thr) ? p.drop_scale : 0.f;
                pf[s2].h[j] = f2bf(pr * dm);
                dsf[s2].h[j] = f2bf(pr * (dpacc[r] * dm - ldsR[qc]) * p.scale);
            }
        const long orow = tokrow * p.lddq + h * HD;
#pragma unroll
        for (int d0 = 0; d0 < NDB; d0 += NDBC) {
            f32x16 dk[NDBC], dv[NDBC];
#pragma unroll
            for (int d = 0; d < NDBC; ++d) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { dk[d][r] = 0.f; dv[d][r] = 0.f; }
                bf16x8 fo[2], fq[2];
                gather_frag_2x2<HD, PITCH>(ldsDO, (d0 + d) * 32 + l31, ldsQ, (d0 + d) * 32 + l31, h2, fo, fq);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    dv[d] = MFMA32(fo[s2], pf[s2].v, dv[d]);
                    dk[d] = MFMA32(fq[s2], dsf[s2].v, dk[d]);
                }
            }
            if (active && tok_ok) {
#pragma unroll
                for (int d = 0; d < NDBC; ++d)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        union { uint2 u; bf16_t h[4]; } a, v;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { a.h[i] = f2bf(dk[d][4 * c + i]); v.h[i] = f2bf(dv[d][4 * c + i]); }
                        const int dcol = (d0 + d) * 32 + 8 * c + 4 * h2;
                        if (HD % 32 != 0 && dcol >= HD) continue;
                        const long off = orow + dcol;
                        *reinterpret_cast<uint2*>(p.dqkv + off + p.D) = a.u;
                        *reinterpret_cast<uint2*>(p.dqkv + off + 2 * p.D) = v.u;
                    }
            }
        }
    }
}

// One wave per work item.  With few items (cfg-2: 384 = 64 samples x 6 heads) four-wave workgroups would occupy only 96 of the
// 256 CUs; single-wave workgroups spread them over the chip.  S3D_ATTN_WPB overrides (tuning).
int waves_per_block(long W) {
    static const int forced = s3d_tune_int("S3D_ATTN_WPB");
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    return W <= 1024 ? 1 : (W <= 2048 ? 2 : 4);
}

// Cooperative (shared-stream) kernels pay off once a (batch, head) has enough tiles to stream; S3D_ATTN_COOP=0/1 forces.
bool use_coop(int N) {
    static const int env = s3d_tune_int("S3D_ATTN_COOP");
    static const int min_tiles = s3d_tune_int("S3D_ATTN_COOP_MIN_TILES") > 0 ? s3d_tune_int("S3D_ATTN_COOP_MIN_TILES") : 6;
    if (env == 0) return false;
    return env > 0 || (N + 31) / 32 >= min_tiles;
}

template <typename K>
void set_lds(K kern, int bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// waves per workgroup of the cooperative long-sequence kernels: S3D_ATTN_COOP_WAVES=4|8 in the tuning builds (measured and rejected,
// the hd = 192 / 256 kernels spill at the 256 registers an eight-wave workgroup leaves them); the product library
// does not even instantiate the eight-wave kernels
#ifdef S3D_EXPERIMENTAL_TILES
#define S3D_COOP8(HD) ((HD) <= 192 && coop_waves() == 8)
#else
#define S3D_COOP8(HD) false
#endif
static int coop_waves() {
    static const int w = s3d_tune_int("S3D_ATTN_COOP_WAVES") == 8 ? 8 : 4;
    return w;
}

// S3dAttnArgs::drop_mask is honoured where BOTH passes run the cooperative kernels (the forward does not at hd = 256); everywhere else
// every kernel evaluates the hash, whatever the caller passed
template <int HD>
static bool mask_ok(const AttnArgs& a) { return HD < 256 && a.drop_thr != 0 && a.drop_mask != nullptr && use_coop(a.N) && a.N > 32; }

// Nine waves per workgroup for the point path's 257-token sequences (nine query tiles of 32): ONE workgroup per (batch, head) streams the keys
// once instead of three times, and no wave idles through the key loop.  hd = 64 without dropout only.  Measured (tools/r5/attn_ab.sh, cfg-4):
// forward 59.5 -> 50.7 us per launch (step -0.7 %); dQ 49.0 -> 53.8 and dK/dV 61.2 -> 123 (240 B of scratch per lane) stay on four waves;
// at 513 tokens (17 tiles: two workgroups of nine waves against five of four) the forward is no faster.
// S3D_ATTN_NINE (tuning builds): bit 0 forward, bit 1 dQ, bit 2 dK/dV.
constexpr int coop_nine_default = 1;
template <int HD>
static bool coop_nine(const AttnArgs& a, int bit) {
    static const int env = s3d_tune_int("S3D_ATTN_NINE");
    const int mask = env >= 0 ? env : coop_nine_default;
    const int T = (a.N + 31) / 32;
    return HD == 64 && a.drop_thr == 0 && a.drop_mask == nullptr && ((mask >> bit) & 1) != 0 && (env >= 0 ? (T + 8) / 9 < (T + 3) / 4 : T == 9);
}

template <int HD>
int fwd_hd(const AttnArgs& a_in, bool split, hipStream_t s) {
    AttnArgs a = a_in;
    if (!mask_ok<HD>(a)) a.drop_mask = nullptr;
    const long W = (long)a.Bb * a.H * ((a.N + 31) / 32);
    // (not at hd = 256: the cooperative forward needs 540 registers there -- 28 B of scratch per lane with, 152 B with the dropout
    // mask -- and its only user, the 197-token second pass of group_embed, is 0.1 % of the cfg-3 step on the per-wave kernel)
    if constexpr (HD < 256) if (use_coop(a.N)) {                // long sequences: four query tiles share one key stream
        const int QT = (a.N + 31) / 32;
#ifdef S3D_EXPERIMENTAL_TILES
        if (S3D_COOP8(HD)) {                                    // two waves per SIMD (the hd = 256 kernels are register-bound at one)
            dim3 g((unsigned)((long)a.Bb * a.H * ((QT + 7) / 8)));
            if (split) {
                const int lds = 2 * CoopStage<HD, 4>::BUF_BYTES;
                set_lds((attn_fwd_coop_kernel<HD, true, 8>), lds);
                hipLaunchKernelGGL((attn_fwd_coop_kernel<HD, true, 8>), g, dim3(512), lds, s, a);
            } else {
                const int lds = 2 * CoopStage<HD, 2>::BUF_BYTES;
                set_lds((attn_fwd_coop_kernel<HD, false, 8>), lds);
                hipLaunchKernelGGL((attn_fwd_coop_kernel<HD, false, 8>), g, dim3(512), lds, s, a);
            }
            S3D_CHECK_LAUNCH("attention_fwd_coop8");
            return 0;
        }
#endif
        if constexpr (HD == 64) if (coop_nine<HD>(a, 0)) {
            dim3 g9((unsigned)((long)a.Bb * a.H * ((QT + 8) / 9)));
            if (split) {
                const int lds = 2 * CoopStage<HD, 4, 576>::BUF_BYTES;
                set_lds((attn_fwd_coop_kernel<HD, true, 9, false>), lds);
                hipLaunchKernelGGL((attn_fwd_coop_kernel<HD, true, 9, false>), g9, dim3(576), lds, s, a);
            } else {
                const int lds = 2 * CoopStage<HD, 2, 576>::BUF_BYTES;
                set_lds((attn_fwd_coop_kernel<HD, false, 9, false>), lds);
                hipLaunchKernelGGL((attn_fwd_coop_kernel<HD, false, 9, false>), g9, dim3(576), lds, s, a);
            }
            S3D_CHECK_LAUNCH_V("attention_fwd_coop", HD * 100 + (split ? 10 : 0) + 9);
            return 0;
        }
        dim3 g((unsigned)((long)a.Bb * a.H * ((QT + 3) / 4)));
        if (split) {
            const int lds = 2 * CoopStage<HD, 4>::BUF_BYTES;
            if constexpr (HD == 192) if (s3d_knob(0) != 2 && (a.N + 31) / 32 >= 3) {      // software-pipelined (see attn_fwd_coop_pipe_kernel)
#define S3D_PIPE_LAUNCH(DROP_, MASK_)                                                                              \
    do {                                                                                                           \
        set_lds((attn_fwd_coop_pipe_kernel<HD, DROP_, MASK_>), lds);                                               \
        hipLaunchKernelGGL((attn_fwd_coop_pipe_kernel<HD, DROP_, MASK_>), g, dim3(256), lds, s, a);                \
    } while (0)
                if (a.drop_thr && a.drop_mask) {
                    bool done = false;
#ifdef S3D_EXPERIMENTAL_TILES      // tuning builds (make EXP=1): timing ablations (wrong results) and the per-tile staging addresses, by knob
                    switch (s3d_knob(1)) {
#define S3D_ABL(N_) case N_: set_lds((attn_fwd_coop_pipe_kernel<HD, true, true, N_>), lds); hipLaunchKernelGGL((attn_fwd_coop_pipe_kernel<HD, true, true, N_>), g, dim3(256), lds, s, a); done = true; break;
                        S3D_ABL(1) S3D_ABL(2) S3D_ABL(3) S3D_ABL(4) S3D_ABL(5)
#undef S3D_ABL
                        default: break;
                    }
                    if (!done && a.p_single_plane && s3d_knob(4) == 0) {
                        set_lds((attn_fwd_coop_pipe_kernel<HD, true, true, 0, true, false>), lds);
                        hipLaunchKernelGGL((attn_fwd_coop_pipe_kernel<HD, true, true, 0, true, false>), g, dim3(256), lds, s, a);
                        done = true;
                    }
#endif
                    if (done) {
                    } else if (a.p_single_plane) {
                        set_lds((attn_fwd_coop_pipe_kernel<HD, true, true, 0, true>), lds);
                        hipLaunchKernelGGL((attn_fwd_coop_pipe_kernel<HD, true, true, 0, true>), g, dim3(256), lds, s, a);
                    } else S3D_PIPE_LAUNCH(true, true);
                }
                else if (a.drop_thr) S3D_PIPE_LAUNCH(true, false);
                else if (a.p_single_plane) {
                    set_lds((attn_fwd_coop_pipe_kernel<HD, false, false, 0, true>), lds);
                    hipLaunchKernelGGL((attn_fwd_coop_pipe_kernel<HD, false, false, 0, true>), g, dim3(256), lds, s, a);
                } else S3D_PIPE_LAUNCH(false, false);
#undef S3D_PIPE_LAUNCH
                S3D_CHECK_LAUNCH_V("attention_fwd_coop", HD * 100 + 10 + (a.drop_thr ? 1 : 0) + (a.drop_mask ? 2 : 0) + 4 + (a.p_single_plane && (!a.drop_thr || a.drop_mask) ? 1000 : 0));
                return 0;
            }
            if (a.drop_thr) {
                set_lds((attn_fwd_coop_kernel<HD, true, 4, true>), lds);
                hipLaunchKernelGGL((attn_fwd_coop_kernel<HD, true, 4, true>), g, dim3(256), lds, s, a);
            } else {
                set_lds((attn_fwd_coop_kernel<HD, true, 4, false>), lds);
                hipLaunchKernelGGL((attn_fwd_coop_kernel<HD, true, 4, false>), g, dim3(256), lds, s, a);
            }
        } else {
            const int lds = 2 * CoopStage<HD, 2>::BUF_BYTES;
            if (a.drop_thr) {
                set_lds((attn_fwd_coop_kernel<HD, false, 4, true>), lds);
                hipLaunchKernelGGL((attn_fwd_coop_kernel<HD, false, 4, true>), g, dim3(256), lds, s, a);
            } else {
                set_lds((attn_fwd_coop_kernel<HD, false, 4, false>), lds);
                hipLaunchKernelGGL((attn_fwd_coop_kernel<HD, false, 4, false>), g, dim3(256), lds, s, a);
            }
        }
        S3D_CHECK_LAUNCH_V("attention_fwd_coop", HD * 100 + (split ? 10 : 0) + (a.drop_thr ? 1 : 0) + (a.drop_mask ? 2 : 0));
        return 0;
    }
    const int wpb = waves_per_block(W);
    dim3 grid((unsigned)((W + wpb - 1) / wpb));
    if (a.drop_thr) {                     // the masked variant (no segment packing: dropout sites are long-sequence / test shapes)
        S3D_REQUIRE(!a.seg, "attention_fwd: dropout with packed segments is not built");
        if (split) {
            set_lds((attn_fwd_kernel<HD, true, false, true>), 4 * 2 * 32 * fwd_pitch(HD) * 2);
            hipLaunchKernelGGL((attn_fwd_kernel<HD, true, false, true>), grid, dim3(64 * wpb), wpb * 2 * 32 * fwd_pitch(HD) * 2, s, a);
        } else {
            set_lds((attn_fwd_kernel<HD, false, false, true>), 4 * 32 * fwd_pitch(HD) * 2);
            hipLaunchKernelGGL((attn_fwd_kernel<HD, false, false, true>), grid, dim3(64 * wpb), wpb * 32 * fwd_pitch(HD) * 2, s, a);
        }
    } else if (split && HD == 256 && a.N <= 32 && (a.D & 7) == 0 && s3d_knob(8) != 0) {     // one tile: Q and K requested together, V under the scores
        const int lds = wpb * 2 * 32 * fwd_pitch(256) * 2;
        set_lds(attn_fwd_tile256_kernel<false>, 4 * 2 * 32 * fwd_pitch(256) * 2);
        set_lds(attn_fwd_tile256_kernel<true>, 4 * 2 * 32 * fwd_pitch(256) * 2);
        if (a.seg) hipLaunchKernelGGL((attn_fwd_tile256_kernel<true>), grid, dim3(64 * wpb), lds, s, a);
        else hipLaunchKernelGGL((attn_fwd_tile256_kernel<false>), grid, dim3(64 * wpb), lds, s, a);
        S3D_CHECK_LAUNCH_V("attention_fwd_tile256", a.seg ? 1 : 0);
        return 0;
    } else if (split) {
        const int lds = wpb * 2 * 32 * fwd_pitch(HD) * 2;
        set_lds(attn_fwd_kernel<HD, true>, 4 * 2 * 32 * fwd_pitch(HD) * 2);
        set_lds(attn_fwd_kernel<HD, true, true>, 4 * 2 * 32 * fwd_pitch(HD) * 2);
        if (a.seg) hipLaunchKernelGGL((attn_fwd_kernel<HD, true, true>), grid, dim3(64 * wpb), lds, s, a);
        else hipLaunchKernelGGL((attn_fwd_kernel<HD, true>), grid, dim3(64 * wpb), lds, s, a);
    } else {
        const int lds = wpb * 32 * fwd_pitch(HD) * 2;
        set_lds(attn_fwd_kernel<HD, false>, 4 * 32 * fwd_pitch(HD) * 2);
        set_lds(attn_fwd_kernel<HD, false, true>, 4 * 32 * fwd_pitch(HD) * 2);
        if (a.seg) hipLaunchKernelGGL((attn_fwd_kernel<HD, false, true>), grid, dim3(64 * wpb), lds, s, a);
        else hipLaunchKernelGGL((attn_fwd_kernel<HD, false>), grid, dim3(64 * wpb), lds, s, a);
    }
    S3D_CHECK_LAUNCH_V("attention_fwd", HD * 1000 + (split ? 100 : 0) + (a.seg ? 10 : 0) + (a.drop_thr ? 1 : 0));
    return 0;
}

template <int HD, int DSPLIT>
int bwd_hd(const AttnArgs& a_in, hipStream_t s, AdamFillQueue* fillq) {
    AttnArgs a = a_in;
    if (!mask_ok<HD>(a)) a.drop_mask = nullptr;
    const long W = (long)a.Bb * a.H * ((a.N + 31) / 32);
    const int wpb = waves_per_block(W);
    dim3 grid((unsigned)((W + wpb - 1) / wpb));
    {
        static const bool no_small = s3d_tune_int("S3D_ATTN_NO_SMALL") >= 0;
        static const bool no_small_big = s3d_tune_int("S3D_ATTN_NO_SMALL_BIG") >= 0;       // hd > 96: back to the two-kernel path
        if (a.N <= 32 && !no_small && (HD <= 96 || !no_small_big)) {
            constexpr int WAVE_LDS = small_bwd_wave_lds(HD);
            constexpr int MAXW = (160 * 1024) / WAVE_LDS >= 4 ? 4 : (160 * 1024) / WAVE_LDS;      // 3 waves per workgroup at hd = 256
            // hd = 256 (cfg-3's first pass): ONE wave per workgroup.  The kernel's only workgroup barrier puts the three waves a CU holds into
            // lockstep -- all loading, then all computing -- while three single-wave workgroups drift apart and overlap one's arithmetic with the
            // others' loads: 677 -> 634 us per launch (two waves per workgroup: 740; tools/r6/attn_pass1_ab.py, profiles/r06_attn_pass1.txt)
            const int w = HD >= 256 ? 1 : (wpb < MAXW ? wpb : MAXW);
            dim3 gs((unsigned)((W + w - 1) / w));
            set_lds(attn_bwd_small_kernel<HD>, MAXW * WAVE_LDS);
            set_lds(attn_bwd_small_kernel<HD, true>, MAXW * WAVE_LDS);
            AdamFill fill = adam_fill_none();
            if (fillq) fill = fillq->take(64 * w);
            gs.x += (unsigned)fill.blocks;
            if (a.seg) hipLaunchKernelGGL((attn_bwd_small_kernel<HD, true>), gs, dim3(64 * w), w * WAVE_LDS, s, a, fill);
            else hipLaunchKernelGGL((attn_bwd_small_kernel<HD>), gs, dim3(64 * w), w * WAVE_LDS, s, a, fill);
            S3D_CHECK_LAUNCH_V("attention_bwd_small", HD * 10 + (a.seg ? 1 : 0));
            return 0;
        }
    }
    const int KT = (a.N + 31) / 32;
    if (use_coop(a.N)) {
        const int lds = 2 * CoopStage<HD, 2>::BUF_BYTES;
#ifdef S3D_EXPERIMENTAL_TILES
        if (S3D_COOP8(HD)) {
            set_lds((attn_bwd_dq_coop_kernel<HD, 8>), lds);
            dim3 g((unsigned)((long)a.Bb * a.H * ((KT + 7) / 8)));
            hipLaunchKernelGGL((attn_bwd_dq_coop_kernel<HD, 8>), g, dim3(512), lds, s, a);
        } else if (coop_nine<HD>(a, 1)) {       // (tuning builds only: the product library ships the nine-wave FORWARD alone)
            if constexpr (HD == 64) {
                set_lds((attn_bwd_dq_coop_kernel<HD, 9>), lds);
                dim3 g9((unsigned)((long)a.Bb * a.H * ((KT + 8) / 9)));
                hipLaunchKernelGGL((attn_bwd_dq_coop_kernel<HD, 9>), g9, dim3(576), lds, s, a);
            }
        } else
#endif
        {
            dim3 g((unsigned)((long)a.Bb * a.H * ((KT + 3) / 4)));
            if constexpr (HD < 256) {
                if (a.drop_mask) {
                    set_lds((attn_bwd_dq_coop_kernel<HD, 4, true>), lds);
                    hipLaunchKernelGGL((attn_bwd_dq_coop_kernel<HD, 4, true>), g, dim3(256), lds, s, a);
                }
            }
            if (!a.drop_mask) {
                set_lds(attn_bwd_dq_coop_kernel<HD>, lds);
                hipLaunchKernelGGL((attn_bwd_dq_coop_kernel<HD>), g, dim3(256), lds, s, a);
            }
        }
        S3D_CHECK_LAUNCH_V("attention_bwd_dq_coop", HD * 10 + (a.drop_mask ? 1 : 0));
    } else {
        const int lds = wpb * 32 * HD * 2;
        set_lds(attn_bwd_dq_kernel<HD>, 4 * 32 * HD * 2);
        set_lds(attn_bwd_dq_kernel<HD, true>, 4 * 32 * HD * 2);
        if (a.seg) hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, true>), grid, dim3(64 * wpb), lds, s, a);
        else hipLaunchKernelGGL((attn_bwd_dq_kernel<HD>), grid, dim3(64 * wpb), lds, s, a);
        S3D_CHECK_LAUNCH_V("attention_bwd_dq", HD * 10 + (a.seg ? 1 : 0));
    }
    if (use_coop(a.N)) {          // long sequences: four key tiles share one query stream
        const int lds = 2 * (2 * 32 * (HD + 8) * 2 + 256 + (a.drop_mask ? 4 * 128 : 0));
#ifdef S3D_EXPERIMENTAL_TILES
        if (S3D_COOP8(HD)) {
            set_lds((attn_bwd_dkv_coop_kernel<HD, DSPLIT, 8>), lds);
            dim3 g2((unsigned)((long)a.Bb * a.H * ((KT + 7) / 8)), DSPLIT);
            hipLaunchKernelGGL((attn_bwd_dkv_coop_kernel<HD, DSPLIT, 8>), g2, dim3(512), lds, s, a);
        } else if (coop_nine<HD>(a, 2)) {       // (tuning builds only: 240 B of scratch per lane, 61 -> 123 us)
            if constexpr (HD == 64) {
                set_lds((attn_bwd_dkv_coop_kernel<HD, DSPLIT, 9>), lds);
                dim3 g9((unsigned)((long)a.Bb * a.H * ((KT + 8) / 9)), DSPLIT);
                hipLaunchKernelGGL((attn_bwd_dkv_coop_kernel<HD, DSPLIT, 9>), g9, dim3(576), lds, s, a);
            }
        } else
#endif
        {
            dim3 g2((unsigned)((long)a.Bb * a.H * ((KT + 3) / 4)), DSPLIT);
            if constexpr (HD < 256) {
                if (a.drop_mask) {
                    bool done = false;
#ifdef S3D_EXPERIMENTAL_TILES
                    if constexpr (HD == 192 && DSPLIT == 1) {
#define S3D_ABL(N_) case N_: set_lds((attn_bwd_dkv_coop_kernel<HD, DSPLIT, 4, true, N_>), lds); hipLaunchKernelGGL((attn_bwd_dkv_coop_kernel<HD, DSPLIT, 4, true, N_>), g2, dim3(256), lds, s, a); done = true; break;
                        switch (s3d_knob(3)) { S3D_ABL(1) S3D_ABL(2) S3D_ABL(3) S3D_ABL(4) S3D_ABL(5) S3D_ABL(6) S3D_ABL(7) S3D_ABL(8) default: break; }
#undef S3D_ABL
                    }
#endif
                    if (!done) {
                    set_lds((attn_bwd_dkv_coop_kernel<HD, DSPLIT, 4, true>), lds);
                    hipLaunchKernelGGL((attn_bwd_dkv_coop_kernel<HD, DSPLIT, 4, true>), g2, dim3(256), lds, s, a);
                    }
                }
            }
            if (!a.drop_mask) {
                set_lds(attn_bwd_dkv_coop_kernel<HD, DSPLIT>, lds);
                hipLaunchKernelGGL((attn_bwd_dkv_coop_kernel<HD, DSPLIT>), g2, dim3(256), lds, s, a);
            }
        }
        S3D_CHECK_LAUNCH_V("attention_bwd_dkv_coop", HD * 100 + DSPLIT * 10 + (a.drop_mask ? 1 : 0));
    } else {
        // the per-wave kernel splits the d range in two from hd = 192 on (one half = 252 registers + 60 B of scratch there)
        constexpr int DS = HD >= 192 ? 2 : DSPLIT;
        const int lds = wpb * (2 * 32 * HD * 2 + 256);
        set_lds(attn_bwd_dkv_kernel<HD, DS>, 4 * (2 * 32 * HD * 2 + 256));
        set_lds(attn_bwd_dkv_kernel<HD, DS, true>, 4 * (2 * 32 * HD * 2 + 256));
        dim3 g2(grid.x, DS);
        if (a.seg) hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, DS, true>), g2, dim3(64 * wpb), lds, s, a);
        else hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, DS>), g2, dim3(64 * wpb), lds, s, a);
        S3D_CHECK_LAUNCH_V("attention_bwd_dkv", HD * 100 + DS * 10 + (a.seg ? 1 : 0));
    }
    return 0;
}


// ------------------------------------------------------------------------------------------- split-precision backward (parity mode)
// Reference-grade attention backward for the split-precision parity mode (S3dAttnArgs::dqkv_lo): every operand is read as hi + lo
// (fp32 value), all arithmetic is fp32 VALU, results are stored as hi + lo pairs.  One thread per (batch, head, token, 16-wide slice of
// the head dimension); the scores of a row / column are recomputed by every slice.  Same mathematics as the MFMA kernels above
// (P = exp(q k^T * scale - lse) from the forward's lse, delta = sum(dO * O), dS = P * (dP - delta) * scale, the dropout mask on P and
// dP, block-diagonal segments), any N / head dimension / layout.  Slow by design: tests only.
struct RefView {
    const bf16_t *hi, *lo; long ld;
    __device__ __forceinline__ float at(long row, int col) const { return bf2f(hi[row * ld + col]) + bf2f(lo[row * ld + col]); }
};
__device__ __forceinline__ bool ref_valid(const AttnArgs& p, int q, int k) { return p.seg == 0 || (q >= p.seg) == (k >= p.seg); }

template <bool KEY_SIDE>
__global__ __launch_bounds__(256) void attn_bwd_ref_kernel(const AttnArgs p) {
    const int hd = p.D / p.H, nch = hd / 16;
    const long total = (long)p.Bb * p.H * p.N * nch;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % nch), t = (int)((e / nch) % p.N), h = (int)((e / ((long)nch * p.N)) % p.H), b = (int)(e / ((long)nch * p.N * p.H));
    const long bh = (long)b * p.H + h;
    const RefView Q{p.qkv_hi, p.qkv_lo, p.ld}, O{p.out_hi, p.out_lo, p.ldo}, DO{p.dout, p.dout_lo, p.lddo};
    const int qc = h * hd, kc = p.D + h * hd, vc = 2 * p.D + h * hd;
    auto rowof = [&](int tok) { return (long)b * p.sb + (long)tok * p.st; };
    const unsigned long long dkey = p.drop_thr ? drop_key(p.drop_seed, p.drop_site) : 0ull;
    float acc0[16], acc1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
    if (!KEY_SIDE && c == 0 && p.delta) {
        float dl = 0.f;
        for (int d = 0; d < hd; ++d) dl += DO.at(rowof(t), qc + d) * O.at(rowof(t), qc + d);
        p.delta[bh * p.N + t] = dl;
    }
    // the other index runs over the whole sequence; (i, j) = (query, key)
    for (int u = 0; u < p.N; ++u) {
        const int i = KEY_SIDE ? u : t, j = KEY_SIDE ? t : u;
        if (!ref_valid(p, i, j)) continue;
        const long ri = rowof(i), rj = rowof(j);
        float sc = 0.f, dp = 0.f, delta = 0.f;
        for (int d = 0; d < hd; ++d) {
            const float dov = DO.at(ri, qc + d);
            sc += Q.at(ri, qc + d) * Q.at(rj, kc + d);
            dp += dov * Q.at(rj, vc + d);
            delta += dov * O.at(ri, qc + d);
        }
        float pr = expf(sc * p.scale - p.lse[bh * p.N + i]);
        float dm = 1.f;
        if (p.drop_thr) dm = drop_keep_attn(dkey, ((unsigned long long)bh * p.N + i) * p.N, (uint32_t)j, p.drop_thr) ? p.drop_scale : 0.f;
        const float ds = pr * (dp * dm - delta) * p.scale;
        if (KEY_SIDE) {
#pragma unroll
            for (int d = 0; d < 16; ++d) {
                acc0[d] += ds * Q.at(ri, qc + 16 * c + d);               // dK
                acc1[d] += pr * dm * DO.at(ri, qc + 16 * c + d);        // dV
            }
        } else {
#pragma unroll
            for (int d = 0; d < 16; ++d) acc0[d] += ds * Q.at(rj, kc + 16 * c + d);      // dQ
        }
    }
    const long orow = rowof(t) * p.lddq;
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        bf16_t hi, lo;
        if (KEY_SIDE) {
            split_bf16(acc0[d], hi, lo);
            p.dqkv[orow + kc + 16 * c + d] = hi; p.dqkv_lo[orow + kc + 16 * c + d] = lo;
            split_bf16(acc1[d], hi, lo);
            p.dqkv[orow + vc + 16 * c + d] = hi; p.dqkv_lo[orow + vc + 16 * c + d] = lo;
        } else {
            split_bf16(acc0[d], hi, lo);
            p.dqkv[orow + qc + 16 * c + d] = hi; p.dqkv_lo[orow + qc + 16 * c + d] = lo;
        }
    }
}

int launch_bwd_ref(const AttnArgs& a, hipStream_t s) {
    S3D_REQUIRE(a.qkv_lo && a.out_lo && a.dout_lo && a.dqkv_lo, "attention_bwd (split precision): qkv_lo, out_lo, dout_lo and dqkv_lo are required");
    const int hd = a.D / a.H;
    S3D_REQUIRE(hd % 16 == 0, "attention_bwd (split precision): head dim %d", hd);
    const long total = (long)a.Bb * a.H * a.N * (hd / 16);
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(attn_bwd_ref_kernel<false>, dim3(grid), dim3(256), 0, s, a);
    hipLaunchKernelGGL(attn_bwd_ref_kernel<true>, dim3(grid), dim3(256), 0, s, a);
    S3D_CHECK_LAUNCH("attention_bwd_ref");
    return 0;
}

int check(const AttnArgs& a) {
    S3D_REQUIRE(a.H > 0 && a.D % a.H == 0, "attention: D=%d not divisible by H=%d", a.D, a.H);
    const int hd = a.D / a.H;
    S3D_REQUIRE(hd == 48 || hd == 64 || hd == 96 || hd == 192 || hd == 256, "attention: head dim %d not built (48/64/96/192/256)", hd);
    S3D_REQUIRE(a.N > 0 && a.Bb > 0, "attention: empty problem");
    S3D_REQUIRE(a.ld % 8 == 0 && a.ldo % 8 == 0, "attention: leading dims must be multiples of 8");
    S3D_REQUIRE(a.seg == 0 || (a.seg <= a.N && a.N <= 2 * a.seg && a.N <= 32),
                "attention: seg=%d needs seg <= N <= 2*seg and N <= 32 (N=%d)", a.seg, a.N);
    return 0;
}

// Two short sequences per 32-row MFMA tile: with N <= 16 (group_embed pass 1: 15 tokens x 12 544 groups x 3 heads per block)
// a (batch, head) problem fills a quarter of the 32x32 score tile and every wave pays the full tile's MFMAs, fragment loads and
// latency.  When consecutive batch entries are contiguous in memory (sb == N * st), the pair (2b, 2b+1) IS a 2N-token
// sequence; the block-diagonal mask (seg = N) keeps the two apart.  Half the waves, the same bytes.  S3D_ATTN_PACK=0 disables.
AttnArgs pack_pairs(const AttnArgs& a) {
    static const int on = s3d_tune_int("S3D_ATTN_PACK");                    // 0 disables (tuning builds)
    if (on == 0 || a.seg != 0 || a.N > 16 || (a.Bb & 1) || a.sb != (long)a.N * a.st || a.drop_thr) return a;
    AttnArgs b = a;
    b.Bb = a.Bb / 2;
    b.seg = a.N;
    b.N = 2 * a.N;
    b.sb = 2 * a.sb;
    return b;
}

}  // namespace

// Whether forward / backward of this problem run in the pair-packed form -- lse / delta are then indexed [(b / 2) * H + h][(b & 1) * N + t].
// A producer of lse other than s3d_launch_attention_fwd (the fused block launch) has to write that layout.
bool s3d_attention_pairs_packed(const AttnArgs& a) { return pack_pairs(a).seg != a.seg; }

int s3d_launch_attention_fwd(const AttnArgs& a0, bool split, hipStream_t s) {
    if (int e = check(a0)) return e;
    const AttnArgs a = pack_pairs(a0);
    if (split) S3D_REQUIRE(a.qkv_lo != nullptr, "attention: split mode needs the lo plane");
    switch (a.D / a.H) {
        case 48: return fwd_hd<48>(a, split, s);
        case 64: return fwd_hd<64>(a, split, s);
        case 96: return fwd_hd<96>(a, split, s);
        case 192: return fwd_hd<192>(a, split, s);
        default: return fwd_hd<256>(a, split, s);
    }
}

int s3d_launch_attention_bwd(const AttnArgs& a0, hipStream_t s, AdamFillQueue* fillq) {
    if (int e = check(a0)) return e;
    const AttnArgs a = pack_pairs(a0);
    if (a.dqkv_lo != nullptr) return launch_bwd_ref(a, s);               // parity mode: fp32 reference kernels on hi + lo operands
    S3D_REQUIRE(a.lddo % 8 == 0 && a.lddq % 8 == 0, "attention: leading dims must be multiples of 8");
    switch (a.D / a.H) {
        case 48: return bwd_hd<48, 1>(a, s, fillq);
        case 64: return bwd_hd<64, 1>(a, s, fillq);
        case 96: return bwd_hd<96, 1>(a, s, fillq);
        case 192: return bwd_hd<192, 1>(a, s, fillq);   // 252 VGPRs, no recompute of S / dP per d-half: 262 -> 138 ms at cfg-3
        default: return bwd_hd<256, 2>(a, s, fillq);
    }
}
