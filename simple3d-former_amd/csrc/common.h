// Common device helpers for the gfx950 (MI355X / CDNA4) kernels of the Simple3D-Former hot path.
// Wave = 64 lanes everywhere.  bf16 values travel as raw uint16_t in memory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
// NOTE: use these native vectors (not HIP's uint4/float4 structs) wherever a value is selected with ?: -- a ternary on
// the struct types lowers to a select between ADDRESSES (one of them a stack temporary) and drags the operands into
// scratch memory (measured: 80-300 B/lane of scratch and 3-5x slower GEMMs).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define S3D_WAVE 64

// ---- bf16 <-> f32 (round-to-nearest-even; inputs are finite in this path) ----
// gfx950 converts in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even): one instruction instead of the add / and / shift
// sequence -- the GELU / LayerNorm / split epilogues are VALU-bound (tools/timeline_probe.py: 20 % of the fc1 GEMM)
__device__ __forceinline__ bf16_t f2bf(float f) {
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
// two values -> one 32-bit word (low half = a), a single v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t f2bf2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
    const f32x2_ v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_));
}
// split-bf16 of a pair: hi / lo words hold (a, b) in (low, high) halves
__device__ __forceinline__ void split_bf16x2(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = f2bf2(a, b);
    lo = f2bf2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// split-bf16: x ~= hi + lo with hi = rne(x), lo = rne(x - hi)  (16 mantissa bits in total)
__device__ __forceinline__ void split_bf16(float x, bf16_t& hi, bf16_t& lo) {
    hi = f2bf(x);
    lo = f2bf(x - bf2f(hi));
}

union U128 {
    u32x4 u;
    bf16x8 v;
    bf16_t h[8];
    uint32_t w[4];
};

// Wave-wide reductions, result in every lane.  Within a row of 16 lanes: four DPP steps (quad_perm xor 1, xor 2, row_half_mirror,
// row_mirror -- each is an exchange between halves that already agree, i.e. a butterfly); the four row results are then read
// with v_readlane and combined as scalars.  ~11 cheap instructions instead of six ds_bpermute round trips through the LDS
// crossbar (the __shfl_xor butterfly), which matters for the few-microsecond LayerNorm / softmax / loss kernels whose
// critical path is two or three of these.
template <typename F>
__device__ __forceinline__ float wave_reduce(float v, F op) {
    int x = __float_as_int(v);
#define S3D_DPP_STEP(ctrl) x = __float_as_int(op(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(x, x, ctrl, 0xf, 0xf, false))));
    S3D_DPP_STEP(0xB1)   // quad_perm:[1,0,3,2]
    S3D_DPP_STEP(0x4E)   // quad_perm:[2,3,0,1]
    S3D_DPP_STEP(0x141)  // row_half_mirror
    S3D_DPP_STEP(0x140)  // row_mirror
#undef S3D_DPP_STEP
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(x, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(x, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(x, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(x, 48));
    return op(op(r0, r1), op(r2, r3));
}
// Reduction across the two 32-lane halves (the 32x32 MFMA layouts keep a row's values in lanes l and l + 32): gfx950's
// v_permlane32_swap hands every lane the lower-half and the upper-half value (tools/probes/permlane_probe.hip) -- one VALU
// instruction instead of the ds_bpermute behind __shfl_xor(v, 32).
__device__ __forceinline__ float half_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
    return __int_as_float(r[0]) + __int_as_float(r[1]);
}
__device__ __forceinline__ float half_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
    return fmaxf(__int_as_float(r[0]), __int_as_float(r[1]));
}
__device__ __forceinline__ float wave_sum(float v) { return wave_reduce(v, [](float a, float b) { return a + b; }); }
__device__ __forceinline__ float wave_max(float v) { return wave_reduce(v, [](float a, float b) { return fmaxf(a, b); }); }

// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7, one exp + 5 FMAs): ~3x fewer VALU instructions than erff in
// the GELU epilogues, with an error far below the bf16/1e-3 budgets of this path.  The reciprocal is the hardware's (v_rcp_f32,
// 1 ulp) instead of an IEEE division (~10 instructions), and gelu' shares the one exponential erf and the density both need:
// the fc1 / fc2-dgrad epilogues evaluate these 64 times per thread on a 128x128 tile (a fifth of the workgroup's lifetime).
__device__ __forceinline__ float erf_poly(float ax, float e) {           // 1 - erf(ax) for ax >= 0, e = exp(-ax^2)
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    return t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f)))) * e;
}
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    return copysignf(1.0f - erf_poly(ax, __expf(-ax * ax)), x);
}
// gelu(x) = x Phi(x) = max(x, 0) - |x| h with h = erfc(|x| / sqrt 2) / 2 = 2^R(|x|): R = log2 of it, a degree-7 minimax fit on
// |x| <= 4 sqrt 2 (weighted for the absolute error of h; beyond that h < 8e-9).  ONE transcendental (v_exp_f32 is base 2 already) + 7 FMAs
// instead of exp + rcp + 8: the erf-GELU epilogues (cfg-3 fc1: 30 k of a 256 x 256 tile's 148 k cycles) are VALU-bound, and a
// transcendental costs four issue slots.  Max abs error against fp64 erf-GELU over [-12, 12]: 3.8e-7 (the A&S 7.1.26 form: 4.6e-7).
__device__ __forceinline__ float gelu_erf(float x) {
    const float ax = fminf(fabsf(x), 5.656854249f);
    float r = 8.469293712e-06f;
    r = fmaf(r, ax, -5.389662502e-05f);
    r = fmaf(r, ax, -4.212578507e-04f);
    r = fmaf(r, ax, 7.389304524e-03f);
    r = fmaf(r, ax, -5.269113505e-02f);
    r = fmaf(r, ax, -4.591531407e-01f);
    r = fmaf(r, ax, -1.151110944e+00f);
    r = fmaf(r, ax, -9.999998964e-01f);
    return fmaf(-fabsf(x), __builtin_amdgcn_exp2f(r), fmaxf(x, 0.f));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float z = x * 0.70710678118654752f, az = fabsf(z);
    const float e = __expf(-az * az);                                    // = exp(-x^2 / 2): erf's exponential AND the normal density's
    const float cdf = 0.5f * (1.0f + copysignf(1.0f - erf_poly(az, e), z));
    return cdf + x * (0.39894228040143268f * e);
}

// parity mode (split-precision backward): libm erf / exp instead of the 1.5e-7 polynomial and the hardware exponential
__device__ __forceinline__ float gelu_erf_grad_exact(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * (0.39894228040143268f * expf(-0.5f * x * x));
}

// counter-based dropout mask (hash of the element index; the oracle evaluates the same function, voxel_oracle.hash_keep_mask)
__device__ __forceinline__ unsigned long long drop_key(const unsigned long long* seed, int site) {
    return (*seed) * 0x9E3779B97F4A7C15ull + (unsigned long long)site * 0xD1B54A32D192ED03ull + 0x632BE59BD9B4E019ull;
}
// 32-bit mixing (two rounds of multiply / xor-shift, "lowbias32" constants) of the element index folded with the key: the
// attention-probability mask of the group encoder layer is evaluated (196 B)^2 * 60 times per pass, where the earlier splitmix64
// (three 64-bit multiplies = a dozen 32-bit ones per element) was 20 - 30 % of the three long-sequence attention kernels.
__device__ __forceinline__ uint32_t drop_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15;
    return x;
}
// The key enters twice: xor-ed into the low index word BEFORE the add of the (keyed) high-word hash -- with a purely additive key the
// masks of two seeds / sites / ranks are shifted copies of one 2^32-periodic sequence (mask_k'(i) == mask_k(i + delta)), and windows of
// ~1.5e8 elements from different steps overlap with a probability of a few per cent.  Same instruction count (the xor replaces nothing
// on the hoisted path: lo ^ klo is formed once per row in DropRow).
__device__ __forceinline__ uint32_t drop_hash(unsigned long long key, unsigned long long idx) {
    const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
    const uint32_t hi = drop_mix32((uint32_t)(idx >> 32) ^ khi) ^ klo;   // changes every 2^32 elements
    return drop_mix32(((uint32_t)idx ^ (klo * 0x9E3779B9u)) + hi);       // lowbias32 avalanches consecutive integers by itself
}
__device__ __forceinline__ bool drop_keep(unsigned long long key, unsigned long long idx, unsigned thr) { return drop_hash(key, idx) >= thr; }
// Attention weights (site 0 of the encoder layer, [b, h, query, key]): ONE hash per pair of adjacent keys (2j, 2j + 1) of a query row --
// the hash of the even key's index -- and 16 bits of it per decision (low half: even key).  The long-sequence forward evaluates it
// for 9.4e9 weights per step and holds a pair in the same lane; p is resolved to 2^-16 (0.1 -> 6553 / 65536).
__device__ __forceinline__ bool drop_half(uint32_t z, uint32_t odd, unsigned thr) { return (odd ? (z >> 16) : (z & 0xffffu)) >= (thr >> 16); }
__device__ __forceinline__ bool drop_keep_attn(unsigned long long key, unsigned long long rowbase, uint32_t k, unsigned thr) {
    return drop_half(drop_hash(key, rowbase + (k & ~1u)), k & 1u, thr);
}

// The same mask for element (base + off), off < 2^32, with the 64-bit part hoisted: the long-sequence attention kernels evaluate the
// mask (196 B)^2 * 60 times per pass, and a 64-bit multiply-add per element to form the index cost more than the hash itself.
struct DropRow { uint32_t lo, mix_a, mix_b, kx; };
__device__ __forceinline__ DropRow drop_row(unsigned long long key, unsigned long long base) {
    const uint32_t hi = (uint32_t)(base >> 32), khi = (uint32_t)(key >> 32), klo = (uint32_t)key;
    return DropRow{(uint32_t)base, drop_mix32(hi ^ khi) ^ klo, drop_mix32((hi + 1u) ^ khi) ^ klo, klo * 0x9E3779B9u};
}
__device__ __forceinline__ uint32_t drop_hash_at(const DropRow& r, uint32_t off) {
    const uint32_t lo = r.lo + off;
    return drop_mix32((lo ^ r.kx) + (lo < r.lo ? r.mix_b : r.mix_a));                  // lo < r.lo: the add carried into the high word
}
__device__ __forceinline__ bool drop_keep_at(const DropRow& r, uint32_t off, unsigned thr) { return drop_hash_at(r, off) >= thr; }

// fp32 atomic add that lowers to global_atomic_add_f32 (built with -munsafe-fp-atomics)
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

// Element counts up to which a grid-stride loop may keep a 32-bit induction variable (`unsigned i; i += gridDim.x * blockDim.x`): the
// increment past the last element must not wrap, and the launchers use at most 2^13 workgroups of 2^8 threads (stride <= 2^21).
#define S3D_U32_LOOP_MAX ((1L << 32) - (1L << 22))

// ---- tuning knobs ----
// Tile / ring-depth / dispatch overrides (the experiments measured and rejected in DESIGN.md section 6, tools/gemm_bench.py ...) exist
// only in the tuning builds (make EXP=1 / TL=1): there s3d_tune_int(name) = atoi(getenv(name)), -1 when unset.  In the product library
// it is the constant -1 and every knob branch folds away; the only environment variable the product library reads is
// S3D_DETERMINISTIC (capi.hip), the documented parity mode.
#ifdef S3D_EXPERIMENTAL_TILES
#include <stdlib.h>
static inline int s3d_tune_int(const char* name) { const char* v = getenv(name); return v ? atoi(v) : -1; }
#else
static inline int s3d_tune_int(const char*) { return -1; }
#endif

// ---- host-side error plumbing (capi.hip owns the storage) ----
void s3d_set_error(const char* fmt, ...);
// Deterministic mode (s3d_set_deterministic / S3D_DETERMINISTIC=1): every reduction that would otherwise combine partial sums
// with fp32 atomics from several workgroups (split-K wgrads, token / bias gradients, loss, final-norm gamma/beta) takes a
// single-writer path instead, so a training step is bitwise reproducible run to run.  Slower; for parity tests.
bool s3d_deterministic();
int s3d_knob(int id);                    // s3d_debug_knob (capi.hip): -1 = the shipped rule
// Launch coverage (s3d_cov_enable / s3d_cov_collect, tests/test_gpu_zz_coverage.py): while enabled, every launcher notes the kernel
// family it dispatched to and an instantiation key (GEMMs: the profiling key = tiles | transposes | split | epilogue; attention /
// LayerNorm / BatchNorm: the template choice), so that a test can prove that every kernel variant a benched training step launches
// has also been launched inside a test that compares with the oracle.  One predictable branch per launch when off.
extern bool g_s3d_cov_on;
void s3d_cov_note(const char* name, long long variant);
#define S3D_CHECK_LAUNCH_V(name, variant)                                              \
    do {                                                                               \
        if (g_s3d_cov_on) s3d_cov_note(name, (long long)(variant));                    \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) {                                                       \
            s3d_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
            return 1;                                                                  \
        }                                                                              \
    } while (0)
#define S3D_CHECK_LAUNCH(name) S3D_CHECK_LAUNCH_V(name, 0)
#define S3D_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            s3d_set_error(__VA_ARGS__);   \
            return 2;                     \
        }                                 \
    } while (0)

// exp(x) through the hardware base-2 exponential (v_exp_f32, ~1 ulp): one multiply + one transcendental instead of libm's
// ~dozen VALU instructions.  exp2(-inf) = 0, which the softmax masking relies on.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
