// LayerNorm of ONE row by one 64-lane wave, shared by the stand-alone kernel (ln.hip) and the fused GEMM epilogue (gemm.hip: the
// last-arriving tile of a row band normalises the band).  v[c] holds columns c*256 + lane*4 .. +3 of the row (already masked to
// zero beyond D); writes mean / rstd, the normalised row as split-bf16 planes and / or fp32.
#pragma once
#include "common.h"

template <int MC>
__device__ __forceinline__ void ln_row_finish(const float4 (&v)[MC], const int lane, const int D, const float eps,
                                              const float* __restrict__ gamma, const float* __restrict__ beta, float* mean_out,
                                              float* rstd_out, bf16_t* out_hi, bf16_t* out_lo, float* out_f32) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MC; ++c) s += v[c].x + v[c].y + v[c].z + v[c].w;
    const float inv_d = 1.0f / (float)D;                  // one division per row instead of two (each ~10 instructions on every lane)
    const float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MC; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < D) {
            const float a = v[c].x - mean, b = v[c].y - mean, cc = v[c].z - mean, d = v[c].w - mean;
            q += a * a + b * b + cc * cc + d * d;
        }
    }
    const float rstd = rsqrtf(wave_sum(q) * inv_d + eps);
    if (lane == 0) {
        if (mean_out) *mean_out = mean;
        if (rstd_out) *rstd_out = rstd;
    }
#pragma unroll
    for (int c = 0; c < MC; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < D) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + col);
            const float4 b = *reinterpret_cast<const float4*>(beta + col);
            float y[4] = {(v[c].x - mean) * rstd * g.x + b.x, (v[c].y - mean) * rstd * g.y + b.y,
                          (v[c].z - mean) * rstd * g.z + b.z, (v[c].w - mean) * rstd * g.w + b.w};
            if (out_f32) *reinterpret_cast<float4*>(out_f32 + col) = make_float4(y[0], y[1], y[2], y[3]);
            if (out_hi) {
                union { uint2 u; bf16_t h[4]; } hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) split_bf16(y[i], hi.h[i], lo.h[i]);
                *reinterpret_cast<uint2*>(out_hi + col) = hi.u;
                if (out_lo) *reinterpret_cast<uint2*>(out_lo + col) = lo.u;
            }
        }
    }
}
