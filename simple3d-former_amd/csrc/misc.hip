// Small HBM-bound kernels of the hot path: tokenizer patch gather, positional-embedding gradient, fp32->split-bf16,
// classification head, cross-entropy, fused Adam.
#include "kernels.h"
#include "adam_fill.h"

#include <string.h>

namespace {

// ------------------------------------------------------------------------------------------- tokenizer fold
// One workgroup per (sample b, x-row X < P*c): the V*V-float slab x[b][X][:][:] is contiguous in HBM, is read once
// with coalesced float4 loads into LDS, and the patch-ordered GEMM operand rows are produced from LDS.
__global__ __launch_bounds__(256) void fold_kernel(const FoldArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sl = reinterpret_cast<float*>(smem);
    const int V = p.V, c = p.c, P = p.P, Pc = P * c;
    const int b = blockIdx.x / Pc, X = blockIdx.x % Pc;
    const int px = X / c, i = X % c;
    const float* slab = p.x + ((long)b * V + X) * V * V;
    for (int t = threadIdx.x; t < V * V / 4; t += 256)
        reinterpret_cast<float4*>(sl)[t] = reinterpret_cast<const float4*>(slab)[t];
    __syncthreads();

    if (p.mode == FOLD_ZMEAN) {
        const long rowbase = (long)b * (P * P + 1) + 1 + px * P;
        for (int t = threadIdx.x; t < Pc * c; t += 256) {
            const int Y = t / c, k = t % c, py = Y / c, j = Y % c;
            float s = 0.f;
            for (int pz = 0; pz < P; ++pz) s += sl[Y * V + pz * c + k];
            bf16_t hi, lo;
            split_bf16(s, hi, lo);
            const long off = (rowbase + py) * p.lda + (i * c + j) * c + k;
            p.a_hi[off] = hi;
            p.a_lo[off] = lo;
        }
    } else if (p.mode == FOLD_NAIVE) {
        const long rowbase = (long)b * (P * P + 1) + 1 + px * P;
        for (int Y = threadIdx.x; Y < Pc; Y += 256) {
            float s = 0.f;
            for (int z = 0; z < V; ++z) s += sl[Y * V + z];
            s = fminf(fmaxf(s, 0.f), 1.f);
            bf16_t hi, lo;
            split_bf16(s, hi, lo);
            const long off = (rowbase + Y / c) * p.lda + i * c + (Y % c);
            p.a_hi[off] = hi;
            p.a_lo[off] = lo;
        }
    } else {
        for (int t = threadIdx.x; t < Pc * Pc; t += 256) {
            const int Y = t / Pc, Z = t % Pc, py = Y / c, j = Y % c, pz = Z / c, k = Z % c;
            long row;
            if (p.mode == FOLD_PATCH) row = (long)b * ((long)P * P * P + 1) + 1 + ((long)px * P + py) * P + pz;
            else row = (((long)b * P + px) * P + py) * (P + 1) + 1 + pz;
            bf16_t hi, lo;
            split_bf16(sl[Y * V + Z], hi, lo);
            const long off = row * p.lda + (i * c + j) * c + k;
            p.a_hi[off] = hi;
            p.a_lo[off] = lo;
        }
    }
}

// ------------------------------------------------------------------------------------------- 2-D patchify (LwF branch)
// timm PatchEmbed = Conv2d(C, D, p, stride p) -> flatten(2).transpose(1, 2): as a GEMM operand, row b*(np+1) + 1 + py*PW + px
// holds the patch in (c, i, j) order (the conv weight's own flattening); row b*(np+1) is the cls slot (zeros: the TOKEN epilogue
// substitutes the cls token).  One thread per 8 consecutive k (32 contiguous bytes of an image row).
__global__ void patchify_kernel(const float* __restrict__ img, bf16_t* __restrict__ a_hi, bf16_t* __restrict__ a_lo, long lda,
                                int B, int C, int H, int W, int p) {
    const int PW = W / p, np_ = (H / p) * PW, K = C * p * p, K8 = K / 8;
    const long total = (long)B * (np_ + 1) * K8;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int k8 = (int)(idx % K8) * 8;
        const long row = idx / K8;
        const int t = (int)(row % (np_ + 1));
        const long b = row / (np_ + 1);
        union { u32x4 u; bf16_t h[8]; } hi, lo;
        if (t == 0) {
            hi.u = u32x4{0u, 0u, 0u, 0u};
            lo.u = hi.u;
        } else {
            const int q = t - 1, py = q / PW, px = q % PW;
            const int c = k8 / (p * p), rem = k8 % (p * p), i = rem / p, j = rem % p;
            const float* src = img + (((long)b * C + c) * H + (py * p + i)) * W + px * p + j;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) { split_bf16(v0[r], hi.h[r], lo.h[r]); split_bf16(v1[r], hi.h[4 + r], lo.h[4 + r]); }
        }
        *reinterpret_cast<u32x4*>(a_hi + row * lda + k8) = hi.u;
        if (a_lo) *reinterpret_cast<u32x4*>(a_lo + row * lda + k8) = lo.u;
    }
}

// ------------------------------------------------------------------------------------------- pos / cls / bias grads
// One wave per (token, 64 feature columns, slice of groups): a thread sums its slice's rows of one (token, column) with
// independent loads and issues one atomic per destination.  Few slices (<= 4) keep the same-address depth low -- the bias
// gradient collects (ntok - 1) * slices atomics per column, and same-address fp32 atomics serialise at ~12 ns each -- while
// ntok * D / 64 * slices waves keep the loads parallel (a token loop inside the thread ran 26 dependent rounds: 14 us).
__global__ __launch_bounds__(64) void posgrad_kernel(const PosGradArgs p, long gchunk) {
    const int t = blockIdx.x, d = blockIdx.y * 64 + threadIdx.x;
    if (d >= p.D) return;
    const long g0 = (long)blockIdx.z * gchunk, g1 = min(p.groups, g0 + gchunk);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long g = g0; g < g1; g += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long gg = g + u;
            const float v = p.dx[(min(gg, g1 - 1) * p.ntok + t) * p.D + d];
            acc[u] += gg < g1 ? v : 0.f;
        }
    }
    const float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    if (p.dpos) atomic_add_f32(p.dpos + (long)t * p.D + d, s);
    if (t == 0) { if (p.dcls) atomic_add_f32(p.dcls + d, s); }
    else if (p.dbias) atomic_add_f32(p.dbias + d, s);
}

// deterministic mode: single writer per destination, fixed summation order.  One workgroup per 16 feature columns, 16 token lanes: thread
// (column, lane) walks the groups of tokens lane, lane + 16, .. in order (eight independent partial sums, combined in a fixed tree), the
// per-token sums meet in LDS and the bias / class-token sums are taken over the tokens in order.  (Until round 5 one THREAD per column
// walked everything: 400 us of a 2.07 ms deterministic cfg-2 step.)
__global__ __launch_bounds__(256) void posgrad_det_kernel(const PosGradArgs p) {
    extern __shared__ float tok_sum[];                                  // [ntok][16]
    const int c = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int d = blockIdx.x * 16 + c;
    const bool ok = d < p.D;
    for (int t = lane; t < p.ntok; t += 16) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ok) {
            for (long g = 0; g < p.groups; g += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const long gg = g + u;
                    const float v = p.dx[(min(gg, p.groups - 1) * p.ntok + t) * p.D + d];
                    acc[u] += gg < p.groups ? v : 0.f;
                }
            }
        }
        const float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        tok_sum[t * 16 + c] = s;
        if (ok && p.dpos) p.dpos[(long)t * p.D + d] += s;
    }
    __syncthreads();
    if (lane == 0 && ok) {
        float bias = 0.f;
        for (int t = 1; t < p.ntok; ++t) bias += tok_sum[t * 16 + c];
        if (p.dcls) p.dcls[d] += tok_sum[c];
        if (p.dbias) p.dbias[d] += bias;
    }
}

// ------------------------------------------------------------------------------------------- token assemble (pass 2)
// out[b*(n+1) + t][:] = (t == 0 ? cls : src[b*n + t - 1]) + pos[t]      ('(b px py) c -> b (px py) c' + cat cls + pos add,
// vit_3d_2d_pretrain.py:485-491);  bwd: dsrc[b*n + j] = dout[b*(n+1) + 1 + j]
__global__ void assemble_tokens_kernel(const float* __restrict__ src, const float* __restrict__ cls,
                                       const float* __restrict__ pos, float* __restrict__ out, long B, int n, int D) {
    const long total = B * (n + 1) * (long)(D / 4);
    if (total < S3D_U32_LOOP_MAX) {                                   // 32-bit index arithmetic (two 64-bit divisions per float4 otherwise)
        const unsigned q = (unsigned)D / 4, n1 = (unsigned)n + 1;
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {
            const unsigned row = i / q, d4 = i - row * q, b = row / n1, t = row - b * n1;
            const f32x4 p4 = reinterpret_cast<const f32x4*>(pos + (long)t * D)[d4];
            const f32x4 s4 = (t == 0) ? reinterpret_cast<const f32x4*>(cls)[d4]
                                      : reinterpret_cast<const f32x4*>(src + ((long)b * n + t - 1) * D)[d4];
            reinterpret_cast<f32x4*>(out + (long)row * D)[d4] = s4 + p4;
        }
        return;
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d4 = (int)(i % (D / 4));
        const long row = i / (D / 4);
        const int t = (int)(row % (n + 1));
        const long b = row / (n + 1);
        const f32x4 p4 = reinterpret_cast<const f32x4*>(pos + (long)t * D)[d4];
        const f32x4 s4 = (t == 0) ? reinterpret_cast<const f32x4*>(cls)[d4]
                                  : reinterpret_cast<const f32x4*>(src + (b * n + t - 1) * D)[d4];
        reinterpret_cast<f32x4*>(out + row * D)[d4] = s4 + p4;
    }
}
__global__ void assemble_tokens_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dsrc, long B, int n, int D) {
    const long total = B * n * (long)(D / 4);
    if (total < S3D_U32_LOOP_MAX) {
        const unsigned q = (unsigned)D / 4, un = (unsigned)n;
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {
            const unsigned row = i / q, d4 = i - row * q, b = row / un, j = row - b * un;
            reinterpret_cast<f32x4*>(dsrc + (long)row * D)[d4] = reinterpret_cast<const f32x4*>(dout + ((long)b * (n + 1) + 1 + j) * D)[d4];
        }
        return;
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d4 = (int)(i % (D / 4));
        const long row = i / (D / 4);
        const long b = row / n;
        const int j = (int)(row % n);
        reinterpret_cast<f32x4*>(dsrc + row * D)[d4] = reinterpret_cast<const f32x4*>(dout + (b * (n + 1) + 1 + j) * D)[d4];
    }
}

// ------------------------------------------------------------------------------------------- fp32 -> split bf16
__global__ void split_kernel(const float* __restrict__ src, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, long rows,
                             long cols, long ld) {
    const long n = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols, cidx = i % cols;
        bf16_t h, l;
        split_bf16(src[i], h, l);
        hi[r * ld + cidx] = h;
        if (lo) lo[r * ld + cidx] = l;
    }
}

// ------------------------------------------------------------------------------------------- head
// one wave per (sample, class)
__global__ __launch_bounds__(256) void head_fwd_kernel(const HeadArgs p) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (long)p.B * p.C) return;
    const int b = (int)(item / p.C), c = (int)(item % p.C);
    const float* x = p.feat + (long)b * p.D;
    float dot = 0.f, xx = 0.f, ww = 0.f;
    for (int d = lane; d < p.D; d += 64) {
        const float xv = x[d];
        const float wv = p.am_softmax ? p.W[(long)d * p.C + c] : p.W[(long)c * p.D + d];
        dot += xv * wv; xx += xv * xv; ww += wv * wv;
    }
    dot = wave_sum(dot);
    if (p.am_softmax) {
        xx = wave_sum(xx); ww = wave_sum(ww);
        dot = p.am_scale * dot / (fmaxf(sqrtf(xx), 1e-12f) * fmaxf(sqrtf(ww), 1e-12f));
    } else {
        dot += p.bias[c];
    }
    if (lane == 0) p.logits[item] = dot;
}

// linear head backward: dfeat[b][d] = sum_c dl[b][c] W[c][d];  dW[c][d] += sum_b dl[b][c] feat[b][d];  db[c] += sum_b dl[b][c]
__global__ void head_bwd_linear_kernel(const HeadArgs p) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nf = (long)p.B * p.D, nw = (long)p.C * p.D;
    if (idx < nf) {
        const int b = (int)(idx / p.D), d = (int)(idx % p.D);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;          // four independent chains: the loop is latency-bound
        int c = 0;
        for (; c + 3 < p.C; c += 4) {
            s0 += p.dlogits[(long)b * p.C + c] * p.W[(long)c * p.D + d];
            s1 += p.dlogits[(long)b * p.C + c + 1] * p.W[(long)(c + 1) * p.D + d];
            s2 += p.dlogits[(long)b * p.C + c + 2] * p.W[(long)(c + 2) * p.D + d];
            s3 += p.dlogits[(long)b * p.C + c + 3] * p.W[(long)(c + 3) * p.D + d];
        }
        for (; c < p.C; ++c) s0 += p.dlogits[(long)b * p.C + c] * p.W[(long)c * p.D + d];
        p.dfeat[idx] = (s0 + s1) + (s2 + s3);
    } else if (idx < nf + nw) {
        const long j = idx - nf;
        const int c = (int)(j / p.D), d = (int)(j % p.D);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int b = 0;
        for (; b + 3 < p.B; b += 4) {
            s0 += p.dlogits[(long)b * p.C + c] * p.feat[(long)b * p.D + d];
            s1 += p.dlogits[(long)(b + 1) * p.C + c] * p.feat[(long)(b + 1) * p.D + d];
            s2 += p.dlogits[(long)(b + 2) * p.C + c] * p.feat[(long)(b + 2) * p.D + d];
            s3 += p.dlogits[(long)(b + 3) * p.C + c] * p.feat[(long)(b + 3) * p.D + d];
        }
        for (; b < p.B; ++b) s0 += p.dlogits[(long)b * p.C + c] * p.feat[(long)b * p.D + d];
        if (p.dW) atomic_add_f32(p.dW + j, (s0 + s1) + (s2 + s3));
    } else if (idx < nf + nw + p.C) {
        const int c = (int)(idx - nf - nw);
        float s = 0.f;
        for (int b = 0; b < p.B; ++b) s += p.dlogits[(long)b * p.C + c];
        if (p.dbias) atomic_add_f32(p.dbias + c, s);
    }
}

// AM-softmax backward.  y[b][c] = s * xn[b].wn[c], xn = x/|x|, wn = W[:,c]/|W[:,c]| (clamp at 1e-12 ignored in bwd,
// exactly as autograd does when the clamp is inactive).
//   dx[b] = (g - xn (xn.g)) / |x|        with g  = s * sum_c dl[b][c] wn[c]
//   dW[:,c] = (gw - wn (wn.gw)) / |w_c|  with gw = s * sum_b dl[b][c] xn[b]
__global__ __launch_bounds__(256) void head_bwd_am_kernel(const HeadArgs p) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item < p.B) {                      // one wave per sample: dfeat
        const int b = (int)item;
        const float* x = p.feat + (long)b * p.D;
        float xx = 0.f;
        for (int d = lane; d < p.D; d += 64) xx += x[d] * x[d];
        const float nx = fmaxf(sqrtf(wave_sum(xx)), 1e-12f);
        float xg = 0.f;
        for (int d = lane; d < p.D; d += 64) {
            float g = 0.f;
            for (int c = 0; c < p.C; ++c) {
                // inverse column norms 1/|w_c| are cached in scratch[0..C) by am_norms_kernel
                g += p.dlogits[(long)b * p.C + c] * p.W[(long)d * p.C + c] * p.scratch[c];
            }
            g *= p.am_scale;
            xg += (x[d] / nx) * g;
        }
        xg = wave_sum(xg);
        for (int d = lane; d < p.D; d += 64) {
            float g = 0.f;
            for (int c = 0; c < p.C; ++c) g += p.dlogits[(long)b * p.C + c] * p.W[(long)d * p.C + c] * p.scratch[c];
            g *= p.am_scale;
            p.dfeat[(long)b * p.D + d] = (g - (x[d] / nx) * xg) / nx;
        }
    } else if (item < (long)p.B + p.C) {   // one wave per class: dW[:, c]
        const int c = (int)(item - p.B);
        const float inw = p.scratch[c];    // 1/|w_c|
        float wg = 0.f;
        for (int d = lane; d < p.D; d += 64) {
            float g = 0.f;
            for (int b = 0; b < p.B; ++b) g += p.dlogits[(long)b * p.C + c] * p.feat[(long)b * p.D + d] * p.scratch[p.C + b];
            g *= p.am_scale;
            wg += p.W[(long)d * p.C + c] * inw * g;
        }
        wg = wave_sum(wg);
        for (int d = lane; d < p.D; d += 64) {
            float g = 0.f;
            for (int b = 0; b < p.B; ++b) g += p.dlogits[(long)b * p.C + c] * p.feat[(long)b * p.D + d] * p.scratch[p.C + b];
            g *= p.am_scale;
            atomic_add_f32(p.dW + (long)d * p.C + c, (g - p.W[(long)d * p.C + c] * inw * wg) * inw);
        }
    }
}

// inverse norms for the AM-softmax backward: inv_w[c] = 1/max(|W[:,c]|,eps), inv_x[b] = 1/max(|x_b|,eps)
__global__ __launch_bounds__(256) void am_norms_kernel(const HeadArgs p, float* inv_w, float* inv_x) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item < p.C) {
        float s = 0.f;
        for (int d = lane; d < p.D; d += 64) { const float w = p.W[(long)d * p.C + item]; s += w * w; }
        s = wave_sum(s);
        if (lane == 0) inv_w[item] = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    } else if (item < (long)p.C + p.B) {
        const long b = item - p.C;
        float s = 0.f;
        for (int d = lane; d < p.D; d += 64) { const float x = p.feat[b * p.D + d]; s += x * x; }
        s = wave_sum(s);
        if (lane == 0) inv_x[b] = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    }
}

// ------------------------------------------------------------------------------------------- cross entropy
__global__ void ce_den_kernel(const CeArgs p) {   // loss[0] = 0, loss[1] = sum_b w[t_b]  (or rows)
    __shared__ float part[256];
    float s = 0.f;
    if (p.weight) for (long r = threadIdx.x; r < p.rows; r += 256) s += p.weight[p.target[r]];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) { p.loss[0] = 0.f; p.loss[1] = p.weight ? part[0] : (float)p.rows; }
}
__global__ __launch_bounds__(256) void ce_main_kernel(const CeArgs p) {   // one wave per row
    const int lane = threadIdx.x & 63;
    const long nw = (long)gridDim.x * 4;
    const float den = p.loss[1];
    float acc = 0.f;
    const int ld = p.ld > 0 ? p.ld : p.C;
    for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < p.rows; row += nw) {
        const float* l = p.logits + row * ld;
        float m = -INFINITY;
        for (int c = lane; c < p.C; c += 64) m = fmaxf(m, l[c]);
        m = wave_max(m);
        float s = 0.f;
        for (int c = lane; c < p.C; c += 64) s += __expf(l[c] - m);
        s = wave_sum(s);
        const long t = p.target[row];
        const float w = p.weight ? p.weight[t] : 1.f;
        const float lse = m + logf(s);
        acc += w * (lse - l[t]);
        if (p.dlogits) {
            // softmax from the exponentials' sum (one fast exp and one multiply per class; the per-row factors are hoisted): this
            // kernel walks 65 536 rows x 50 classes in the part-segmentation step
            const float inv_s = 1.0f / s, gw = p.grad_scale * w / den;
            for (int c = lane; c < ld; c += 64)
                p.dlogits[row * ld + c] = (c < p.C) ? gw * (__expf(l[c] - m) * inv_s - (c == t ? 1.f : 0.f)) : 0.f;
        }
    }
    if (gridDim.x == 1) {                        // deterministic mode: the four wave sums are combined in a fixed order
        __shared__ float part[4];
        if (lane == 0) part[threadIdx.x >> 6] = acc / den;
        __syncthreads();
        if (threadIdx.x == 0) p.loss[0] += (part[0] + part[1]) + (part[2] + part[3]);
        return;
    }
    if (lane == 0 && acc != 0.f) atomic_add_f32(p.loss, acc / den);
}

// ------------------------------------------------------------------------------------------- fused loss end of the step
// block-wide sum of one float per thread (256 threads), result in every thread; `red` = 4 floats of LDS
__device__ __forceinline__ float block_sum4(float v, float* red, int tid) {
    v = wave_sum(v);
    __syncthreads();                                   // `red` may still be read from the previous use
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// one workgroup per sample: final norm -> logits -> cross entropy -> d(logits) -> d(feat) -> d(x) of the class row
__global__ __launch_bounds__(256) void head_loss_sample_kernel(const S3dHeadLossArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* feat = reinterpret_cast<float*>(smem);      // [D]
    float* xhat = feat + p.D;                           // [D]
    float* dfe = xhat + p.D;                            // [D]
    float* lg = dfe + p.D;                              // [C] logits, then d(logits)
    float* red = lg + ((p.C + 3) & ~3);                 // [4]
    float* Wl = red + 4;                                // [C][D] copy of the head weight (when it fits: p.scratch_w != 0)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x, D = p.D, C = p.C;
    const float* x = p.x + (long)b * p.ldx;
    const bool w_lds = (long)C * D * 4 <= 96 * 1024;    // uniform
    if (w_lds) {                                        // every load of the copy is in flight before anything waits (latency-bound kernel)
        const int n4 = C * D / 4;
        for (int i0 = tid; i0 < n4; i0 += 256 * 8) {
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = reinterpret_cast<const f32x4*>(p.W)[min(i0 + u * 256, n4 - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + u * 256 < n4) reinterpret_cast<f32x4*>(Wl)[i0 + u * 256] = t[u];
        }
    }
    const float* Wsrc = w_lds ? Wl : p.W;
    if (p.zero_tokens > 0) {                            // d(x_out) of the sample's other tokens: zero (instead of a fill launch in front of this one)
        const long n4 = (long)p.zero_tokens * D / 4;
        f32x4* z = reinterpret_cast<f32x4*>(p.dx + (long)b * p.lddx + D);
        if (p.dx) for (long i = tid; i < n4; i += 256) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.dx_bf) {
            u32x2* zb = reinterpret_cast<u32x2*>(p.dx_bf + (long)b * p.lddx + D);
            for (long i = tid; i < n4; i += 256) zb[i] = u32x2{0u, 0u};
        }
    }
    // ---- final LayerNorm of the class row (two-pass statistics, as ln_row_finish)
    float xv[4], s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int d = tid + i * 256; xv[i] = d < D ? x[d] : 0.f; s += xv[i]; }
    const float mean = block_sum4(s, red, tid) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int d = tid + i * 256; if (d < D) { const float a = xv[i] - mean; q += a * a; } }
    const float rstd = rsqrtf(block_sum4(q, red, tid) / D + p.eps);
    if (tid == 0) { if (p.mean) p.mean[b] = mean; if (p.rstd) p.rstd[b] = rstd; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int d = tid + i * 256;
        if (d < D) {
            const float xh = (xv[i] - mean) * rstd, f = xh * p.gamma[d] + p.beta[d];
            xhat[d] = xh; feat[d] = f;
            if (p.feat) p.feat[(long)b * D + d] = f;
            p.scratch[(long)b * (2 * D + 1) + D + d] = xh;
        }
    }
    __syncthreads();
    // ---- logits: one wave per class (round robin)
    for (int c = wave; c < C; c += 4) {
        const float* w = Wsrc + (long)c * D;
        float dot = 0.f;
        for (int d = lane; d < D; d += 64) dot += feat[d] * w[d];
        dot = wave_sum(dot);
        if (lane == 0) { const float l = dot + (p.bias ? p.bias[c] : 0.f); lg[c] = l; p.logits[(long)b * C + c] = l; }
    }
    // ---- denominator of the (weighted) mean: every workgroup derives it from the targets (B is a batch size)
    float dsum = 0.f;
    if (p.weight) for (int r = tid; r < p.B; r += 256) dsum += p.weight[p.target[r]];
    const float den = p.weight ? block_sum4(dsum, red, tid) : (float)p.B;
    __syncthreads();                                   // logits complete
    // ---- cross entropy of the row (wave 0) and d(logits)
    if (wave == 0) {
        float mx = -INFINITY;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, lg[c]);
        mx = wave_max(mx);
        float se = 0.f;
        for (int c = lane; c < C; c += 64) se += expf(lg[c] - mx);
        se = wave_sum(se);
        const long t = p.target[b];
        const float w = p.weight ? p.weight[t] : 1.f, lse = mx + logf(se);
        if (lane == 0) p.scratch[(long)b * (2 * D + 1) + 2 * D] = w * (lse - lg[t]) / den;       // this sample's share of the loss
        float dl[4];                                     // C <= 256
        for (int k = 0, c = lane; c < C; c += 64, ++k) dl[k] = p.grad_scale * w * (expf(lg[c] - lse) - (c == t ? 1.f : 0.f)) / den;
        for (int k = 0, c = lane; c < C; c += 64, ++k) { lg[c] = dl[k]; p.dlogits[(long)b * C + c] = dl[k]; }
    }
    __syncthreads();
    // ---- d(feat) = d(logits) . W, then the LayerNorm backward of the row
    float g[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int d = tid + i * 256;
        g[i] = 0.f;
        if (d < D) {
            float a0 = 0.f, a1 = 0.f;
            int c = 0;
            for (; c + 1 < C; c += 2) { a0 += lg[c] * Wsrc[(long)c * D + d]; a1 += lg[c + 1] * Wsrc[(long)(c + 1) * D + d]; }
            if (c < C) a0 += lg[c] * Wsrc[(long)c * D + d];
            const float df = a0 + a1;
            p.scratch[(long)b * (2 * D + 1) + d] = df;
            g[i] = df * p.gamma[d];
            s1 += g[i]; s2 += g[i] * xhat[d];
        }
    }
    s1 = block_sum4(s1, red, tid) / D;
    s2 = block_sum4(s2, red, tid) / D;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int d = tid + i * 256;
        if (d < D) {
            const float dx = rstd * (g[i] - s1 - xhat[d] * s2);
            if (p.dx) p.dx[(long)b * p.lddx + d] = dx;
            if (p.dx_bf) p.dx_bf[(long)b * p.lddx + d] = f2bf(dx);
        }
    }
}

// reductions over the batch (fixed order: deterministic): dW [C][D], dbias [C], dgamma / dbeta [D], loss.  A workgroup owns 64
// consecutive outputs; its four waves each take a quarter of the batch (b = wave, wave + 4, ...) with 16 independent load pairs in
// flight per trip, and the four partials are added in wave order -- one trip at B = 64 instead of a 64-deep latency chain.
__global__ __launch_bounds__(256) void head_loss_reduce_kernel(const S3dHeadLossArgs p) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long idx = (long)blockIdx.x * 64 + lane;
    const int D = p.D, C = p.C, B = p.B;
    const long nw = (long)C * D, st = 2 * D + 1, nout = nw + C + 2 * D;
    // what this output sums: u[b] * v[b] with u / v rows of stride su / sv (v == nullptr: plain sum of u)
    const float *u = nullptr, *v = nullptr;
    long su = 0, sv = 0;
    if (idx < nw) { u = p.dlogits + idx / D; su = C; v = p.feat + idx % D; sv = D; }
    else if (idx < nw + C) { u = p.dlogits + (idx - nw); su = C; }
    else if (idx < nw + C + D) { u = p.scratch + (idx - nw - C); su = st; v = p.scratch + D + (idx - nw - C); sv = st; }
    else if (idx < nout) { u = p.scratch + (idx - nw - C - D); su = st; }
    float acc = 0.f;
    if (u) {
        for (int b0 = wave; b0 < B; b0 += 64) {
            float uu[16], vv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const long bb = min(b0 + 4 * k, B - 1);
                uu[k] = u[bb * su]; vv[k] = v ? v[bb * sv] : 1.f;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += (b0 + 4 * k < B) ? uu[k] * vv[k] : 0.f;
        }
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && u) {
        const float t = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        if (idx < nw) { if (p.dW) p.dW[idx] += t; }
        else if (idx < nw + C) { if (p.dbias) p.dbias[idx - nw] += t; }
        else if (idx < nw + C + D) { if (p.dgamma) p.dgamma[idx - nw - C] += t; }
        else { if (p.dbeta) p.dbeta[idx - nw - C - D] += t; }
    }
    if (blockIdx.x == gridDim.x - 1 && wave == 3) {      // the loss: B per-sample shares; lane-strided partials + a fixed butterfly
        float a = 0.f, dn = 0.f;                           // (deterministic; one load per lane instead of a B-deep latency chain)
        for (int bb = lane; bb < B; bb += 64) {
            a += p.scratch[bb * st + 2 * D];
            if (p.weight) dn += p.weight[p.target[bb]];
        }
        a = wave_sum(a); dn = wave_sum(dn);
        if (lane == 0) { p.loss[0] = a; p.loss[1] = p.weight ? dn : (float)B; }
    }
}

// ------------------------------------------------------------------------------------------- Adam
__global__ void adam_prelude_kernel(AdamState* st) {
    st->step += 1;
    const double b1 = st->beta1, b2 = st->beta2;
    const double bc1 = 1.0 - pow(b1, (double)st->step);
    const double bc2 = 1.0 - pow(b2, (double)st->step);
    st->step_size = (float)((double)st->lr / bc1);
    st->bc2_sqrt = (float)sqrt(bc2);
}
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, bf16_t* __restrict__ hi,
                                                   bf16_t* __restrict__ lo, long n4, const AdamState* st, int zero_grad,
                                                   const bf16_t* __restrict__ gw) {
    const float b1 = st->beta1, b2 = st->beta2, eps = st->eps, gs = st->grad_scale;
    const float step_size = st->step_size, bc2s = st->bc2_sqrt;
    // back to front: the arena is laid out in forward order, so the tokenizer's and the first blocks' weight planes are the LAST
    // thing this 0.8 GB stream leaves in the 256 MB Infinity Cache -- where the next step's forward looks for them first
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < n4; j += (long)gridDim.x * 256)
        adam_update4(n4 - 1 - j, p, g, m, v, hi, lo, gw, b1, b2, eps, gs, step_size, bc2s, zero_grad);
}
// the same update over up to 64 disjoint ranges of the arena in ONE launch (what the filler shares of adam_fill.h left over: the
// LayerNorm parameters of every block, the block whose backward ran last, tokenizer / head)
struct AdamRanges { long off4[64]; long cum4[65]; int n; };
__global__ __launch_bounds__(256) void adam_ranges_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                          bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, const AdamRanges r, const AdamState* st,
                                                          int zero_grad) {
    const float b1 = st->beta1, b2 = st->beta2, eps = st->eps, gs = st->grad_scale;
    const float step_size = st->step_size, bc2s = st->bc2_sqrt;
    const long total = r.cum4[r.n];
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < total; j += (long)gridDim.x * 256) {
        int k = 0;
        while (k + 1 < r.n && j >= r.cum4[k + 1]) ++k;                  // <= 64 entries, wave-mostly-uniform
        adam_update4(r.off4[k] + (j - r.cum4[k]), p, g, m, v, hi, lo, nullptr, b1, b2, eps, gs, step_size, bc2s, zero_grad);
    }
}

}  // namespace

int s3d_launch_fold(const FoldArgs& a, hipStream_t s) {
    S3D_REQUIRE(a.V % 2 == 0 && a.P * a.c <= a.V, "fold: V=%d must be even and P*c=%d <= V", a.V, a.P * a.c);
    const int lds = a.V * a.V * 4;
    S3D_REQUIRE(lds <= 160 * 1024, "fold: V=%d slab does not fit LDS", a.V);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fold_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL(fold_kernel, dim3((unsigned)(a.B * a.P * a.c)), dim3(256), lds, s, a);
    S3D_CHECK_LAUNCH("fold");
    return 0;
}

int s3d_launch_patchify(const float* img, bf16_t* a_hi, bf16_t* a_lo, long lda, int B, int C, int H, int W, int p, hipStream_t s) {
    S3D_REQUIRE(p > 0 && p % 8 == 0 && H % p == 0 && W % p == 0, "patchify: patch %d must be a multiple of 8 dividing %dx%d", p, H, W);
    S3D_REQUIRE(lda >= (long)C * p * p && lda % 8 == 0, "patchify: lda=%ld too small for K=%d", lda, C * p * p);
    const long total = (long)B * ((H / p) * (W / p) + 1) * (C * p * p / 8);
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)blocks), dim3(256), 0, s, img, a_hi, a_lo, lda, B, C, H, W, p);
    S3D_CHECK_LAUNCH("patchify");
    return 0;
}

int s3d_launch_posgrad(const PosGradArgs& a, hipStream_t s) {
    long gs = (a.groups + 15) / 16;            // >= 16 rows per thread
    if (gs > 4) gs = 4;
    if (gs < 1) gs = 1;
    const long gchunk = (a.groups + gs - 1) / gs;
    gs = (a.groups + gchunk - 1) / gchunk;
    S3D_REQUIRE(a.ntok <= 65535, "posgrad: ntok=%d too large", a.ntok);
    if (s3d_deterministic()) {
        S3D_REQUIRE((size_t)a.ntok * 64 <= 160 * 1024, "posgrad (deterministic): ntok=%d too large", a.ntok);
        static bool attr = false;               // above 64 KB of dynamic LDS a launch needs the attribute (ntok > 1024)
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(posgrad_det_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
        hipLaunchKernelGGL(posgrad_det_kernel, dim3((unsigned)((a.D + 15) / 16)), dim3(256), (size_t)a.ntok * 64, s, a);
        S3D_CHECK_LAUNCH("posgrad (deterministic)");
        return 0;
    }
    hipLaunchKernelGGL(posgrad_kernel, dim3((unsigned)a.ntok, (unsigned)((a.D + 63) / 64), (unsigned)gs), dim3(64), 0, s, a, gchunk);
    S3D_CHECK_LAUNCH("posgrad");
    return 0;
}

int s3d_launch_assemble(const float* src, const float* cls, const float* pos, float* out, long B, int n, int D, hipStream_t s) {
    S3D_REQUIRE(D % 4 == 0, "assemble: D must be a multiple of 4");
    long blocks = (B * (n + 1) * (long)(D / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(assemble_tokens_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, cls, pos, out, B, n, D);
    S3D_CHECK_LAUNCH("assemble_tokens");
    return 0;
}
int s3d_launch_assemble_bwd(const float* dout, float* dsrc, long B, int n, int D, hipStream_t s) {
    long blocks = (B * n * (long)(D / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(assemble_tokens_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dout, dsrc, B, n, D);
    S3D_CHECK_LAUNCH("assemble_tokens_bwd");
    return 0;
}

int s3d_launch_split(const float* src, bf16_t* hi, bf16_t* lo, long rows, long cols, long ld_out, hipStream_t s) {
    const long n = rows * cols;
    if (n <= 0) return 0;
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, hi, lo, rows, cols, ld_out);
    S3D_CHECK_LAUNCH("split");
    return 0;
}

int s3d_launch_head_fwd(const HeadArgs& a, hipStream_t s) {
    const long items = (long)a.B * a.C;
    hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, a);
    S3D_CHECK_LAUNCH("head_fwd");
    return 0;
}

int s3d_launch_head_bwd(const HeadArgs& a, hipStream_t s) {
    if (!a.am_softmax) {
        const long n = (long)a.B * a.D + (long)a.C * a.D + a.C;
        hipLaunchKernelGGL(head_bwd_linear_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
        S3D_CHECK_LAUNCH("head_bwd");
        return 0;
    }
    S3D_REQUIRE(a.scratch != nullptr, "head_bwd(am): needs scratch of C + B floats");
    const long items = (long)a.B + a.C;
    hipLaunchKernelGGL(am_norms_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, a, a.scratch, a.scratch + a.C);
    hipLaunchKernelGGL(head_bwd_am_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, a);
    S3D_CHECK_LAUNCH("head_bwd_am");
    return 0;
}

int s3d_launch_ce(const CeArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(ce_den_kernel, dim3(1), dim3(256), 0, s, a);
    long blocks = (a.rows + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    if (s3d_deterministic() && a.rows <= 4096) blocks = 1;       // one workgroup: four fixed-order partial sums (large row counts keep
                                                                  // the parallel grid: only the reported loss value, never a gradient, depends on the order)
    hipLaunchKernelGGL(ce_main_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    S3D_CHECK_LAUNCH("cross_entropy");
    return 0;
}

int s3d_launch_head_loss(const S3dHeadLossArgs& a, hipStream_t s) {
    S3D_REQUIRE(a.x && a.gamma && a.beta && a.W && a.target && a.feat && a.logits && a.dlogits && a.loss && a.scratch,
                "head_loss_fused: null argument");
    S3D_REQUIRE(a.B > 0 && a.D > 0 && a.D <= 1024 && a.C > 0 && a.C <= 256, "head_loss_fused: B=%d D=%d (<= 1024) C=%d (<= 256)", a.B, a.D, a.C);
    size_t lds = (size_t)(3 * a.D + ((a.C + 3) & ~3) + 4) * sizeof(float);
    if ((long)a.C * a.D * 4 <= 96 * 1024) lds += (size_t)a.C * a.D * sizeof(float);       // the head weight rides along in LDS
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(head_loss_sample_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL(head_loss_sample_kernel, dim3((unsigned)a.B), dim3(256), lds, s, a);
    const long n = (long)a.C * a.D + a.C + 2 * a.D;
    hipLaunchKernelGGL(head_loss_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, a);
    S3D_CHECK_LAUNCH("head_loss_fused");
    return 0;
}

int s3d_launch_adam_begin(AdamState* st, hipStream_t s) {
    hipLaunchKernelGGL(adam_prelude_kernel, dim3(1), dim3(1), 0, s, st);
    S3D_CHECK_LAUNCH("adam prelude");
    return 0;
}
int s3d_launch_adam_apply(float* p, float* g, float* m, float* v, bf16_t* hi, bf16_t* lo, long n, const AdamState* st,
                          int zero_grad, const bf16_t* g_wire, int max_blocks, hipStream_t s) {
    S3D_REQUIRE(n % 4 == 0, "adam: slice length %ld must be a multiple of 4", n);
    if (n == 0) return 0;
    long blocks = (n / 4 + 255) / 256;
    static const int tuned = s3d_tune_int("S3D_ADAM_BLOCKS");
    // one float4 per thread up to 33 M parameters: with the non-temporal accesses the grid-stride walk on 2048 workgroups left the
    // update at 5.3 TB/s (cfg-2 step 1.738 -> 1.712 ms with 32768; 16384: 1.717)
    const long cap = max_blocks > 0 ? max_blocks : (tuned > 0 ? tuned : 32768);
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, hi, lo, n / 4, st, zero_grad, g_wire);
    S3D_CHECK_LAUNCH("adam");
    return 0;
}
int s3d_launch_adam_ranges(float* p, float* g, float* m, float* v, bf16_t* hi, bf16_t* lo, const long* ranges, int n, const AdamState* st,
                           int zero_grad, hipStream_t s) {
    S3D_REQUIRE(n >= 0 && n <= 64, "adam_ranges: %d ranges (at most 64)", n);
    AdamRanges r;
    memset(&r, 0, sizeof(r));
    for (int i = 0; i < n; ++i) {
        S3D_REQUIRE(ranges[2 * i] % 4 == 0 && ranges[2 * i + 1] % 4 == 0 && ranges[2 * i + 1] >= 0, "adam_ranges: range %d is not float4-aligned", i);
        if (ranges[2 * i + 1] == 0) continue;
        r.off4[r.n] = ranges[2 * i] / 4;
        r.cum4[r.n + 1] = r.cum4[r.n] + ranges[2 * i + 1] / 4;
        ++r.n;
    }
    if (r.n == 0) return 0;
    long blocks = (r.cum4[r.n] + 255) / 256;
    if (blocks > 32768) blocks = 32768;
    hipLaunchKernelGGL(adam_ranges_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, hi, lo, r, st, zero_grad);
    S3D_CHECK_LAUNCH("adam_ranges");
    return 0;
}
int s3d_launch_adam(float* p, float* g, float* m, float* v, bf16_t* hi, bf16_t* lo, long n, AdamState* st,
                    int zero_grad, const bf16_t* g_wire, hipStream_t s) {
    S3D_REQUIRE(n % 4 == 0, "adam: arena length %ld must be a multiple of 4", n);
    if (int rc = s3d_launch_adam_begin(st, s)) return rc;
    return s3d_launch_adam_apply(p, g, m, v, hi, lo, n, st, zero_grad, g_wire, 0, s);
}

// fp32 -> bf16 (rne), 8 elements per thread: the gradient wire format of the data-parallel trainer
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const f32x4 a = reinterpret_cast<const f32x4*>(src)[2 * i], b = reinterpret_cast<const f32x4*>(src)[2 * i + 1];
        U128 o;
        o.h[0] = f2bf(a[0]); o.h[1] = f2bf(a[1]); o.h[2] = f2bf(a[2]); o.h[3] = f2bf(a[3]);
        o.h[4] = f2bf(b[0]); o.h[5] = f2bf(b[1]); o.h[6] = f2bf(b[2]); o.h[7] = f2bf(b[3]);
        reinterpret_cast<u32x4*>(dst)[i] = o.u;
    }
}
int s3d_launch_pack_bf16(const float* src, bf16_t* dst, long n, hipStream_t s) {
    S3D_REQUIRE(n % 8 == 0 && ((uintptr_t)src & 31) == 0 && ((uintptr_t)dst & 15) == 0,
                "pack_bf16: n=%ld must be a multiple of 8 and the buffers 32 / 16-byte aligned", n);
    if (n == 0) return 0;
    long blocks = (n / 8 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, n / 8);
    S3D_CHECK_LAUNCH("pack_bf16");
    return 0;
}

// ------------------------------------------------------------------------------------------- AM-softmax as a per-row head
// AMSoftmaxLayer.forward (models/3DViT/model.py:134-142; the same class at models/vit_3d_2d_pretrain.py:50-56) on many rows (the
// per-point head of PointTransformerSeg: B*N = 65 536 rows): logits = s * (x / max(|x|, 1e-12)) @ (W / max(|W[:, c]|, 1e-12)).
// Factored as a row normalisation (these kernels) around the ordinary Linear-layer GEMMs with the weight Wl[c][d] = s * W[d][c] / |W[:, c]|:
//   s3d_l2norm_rows_fwd   xn = x / |x|  -> split-bf16 GEMM operand planes, 1 / |x| saved per row
//   s3d_l2norm_rows_bwd   dx = (dxn - xn (xn . dxn)) / |x|
//   s3d_am_weight_fwd     Wl (fp32 [C][ldw]) and 1 / |W[:, c]|
//   s3d_am_weight_bwd     dW[d][c] += s * (dWl[c][d] - wn[d][c] (wn[:, c] . dWl[c][:])) / |W[:, c]|
// (the clamps are inactive for any non-degenerate input and, like autograd's, have no backward term.)
__global__ __launch_bounds__(256) void l2norm_rows_fwd_kernel(const float* __restrict__ x, long ldx, long rows, int D, float* __restrict__ inv_norm,
                                                              bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, long ldo) {
    const int lane = threadIdx.x & 63;
    for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * 4) {
        const float* xr = x + r * ldx;
        float ss = 0.f;
        for (int d = lane; d < D; d += 64) ss += xr[d] * xr[d];
        const float inv = 1.0f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
        if (lane == 0) inv_norm[r] = inv;
        for (int d = lane; d < D; d += 64) {
            bf16_t h, l;
            split_bf16(xr[d] * inv, h, l);
            hi[r * ldo + d] = h;
            if (lo) lo[r * ldo + d] = l;
        }
    }
}
__global__ __launch_bounds__(256) void l2norm_rows_bwd_kernel(const float* __restrict__ dxn, long lddxn, const float* __restrict__ x, long ldx,
                                                              const float* __restrict__ inv_norm, long rows, int D, float* __restrict__ dx, long lddx) {
    const int lane = threadIdx.x & 63;
    for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * 4) {
        const float inv = inv_norm[r];
        float dot = 0.f;
        for (int d = lane; d < D; d += 64) dot += x[r * ldx + d] * inv * dxn[r * lddxn + d];
        dot = wave_sum(dot);
        for (int d = lane; d < D; d += 64) dx[r * lddx + d] = (dxn[r * lddxn + d] - x[r * ldx + d] * inv * dot) * inv;   // in place is fine: one reader per entry
    }
}
// one wave per class
__global__ __launch_bounds__(256) void am_weight_fwd_kernel(const float* __restrict__ W, int D, int C, float s, float* __restrict__ Wl, int ldw,
                                                            float* __restrict__ inv_w) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    float ss = 0.f;
    for (int d = lane; d < D; d += 64) { const float w = W[(long)d * C + c]; ss += w * w; }
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
    if (lane == 0) inv_w[c] = inv;
    for (int d = lane; d < D; d += 64) Wl[(long)c * ldw + d] = s * W[(long)d * C + c] * inv;
}
__global__ __launch_bounds__(256) void am_weight_bwd_kernel(const float* __restrict__ dWl, int ldw, const float* __restrict__ W, const float* __restrict__ inv_w,
                                                            int D, int C, float s, float* __restrict__ dW) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    const float inv = inv_w[c];
    float dot = 0.f;
    for (int d = lane; d < D; d += 64) dot += W[(long)d * C + c] * inv * dWl[(long)c * ldw + d];
    dot = wave_sum(dot);
    for (int d = lane; d < D; d += 64) dW[(long)d * C + c] += s * (dWl[(long)c * ldw + d] - W[(long)d * C + c] * inv * dot) * inv;
}

int s3d_launch_l2norm_rows_fwd(const float* x, long ldx, long rows, int D, float* inv_norm, bf16_t* hi, bf16_t* lo, long ldo, hipStream_t s) {
    S3D_REQUIRE(x && inv_norm && hi && D > 0 && ldx >= D && ldo >= D, "l2norm_rows_fwd: x, inv_norm, hi required; ldx, ldo >= D = %d", D);
    if (rows <= 0) return 0;
    long blocks = (rows + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(l2norm_rows_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, ldx, rows, D, inv_norm, hi, lo, ldo);
    S3D_CHECK_LAUNCH("l2norm_rows_fwd");
    return 0;
}
int s3d_launch_l2norm_rows_bwd(const float* dxn, long lddxn, const float* x, long ldx, const float* inv_norm, long rows, int D, float* dx, long lddx,
                               hipStream_t s) {
    S3D_REQUIRE(dxn && x && inv_norm && dx && D > 0, "l2norm_rows_bwd: null pointer");
    if (rows <= 0) return 0;
    long blocks = (rows + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(l2norm_rows_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dxn, lddxn, x, ldx, inv_norm, rows, D, dx, lddx);
    S3D_CHECK_LAUNCH("l2norm_rows_bwd");
    return 0;
}
int s3d_launch_am_weight_fwd(const float* W, int D, int C, float scale, float* Wl, int ldw, float* inv_w, hipStream_t s) {
    S3D_REQUIRE(W && Wl && inv_w && D > 0 && C > 0 && ldw >= D, "am_weight_fwd: W [D][C], Wl [C][ldw >= D], inv_w [C] required");
    hipLaunchKernelGGL(am_weight_fwd_kernel, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, s, W, D, C, scale, Wl, ldw, inv_w);
    S3D_CHECK_LAUNCH("am_weight_fwd");
    return 0;
}
int s3d_launch_am_weight_bwd(const float* dWl, int ldw, const float* W, const float* inv_w, int D, int C, float scale, float* dW, hipStream_t s) {
    S3D_REQUIRE(dWl && W && inv_w && dW && D > 0 && C > 0 && ldw >= D, "am_weight_bwd: null pointer / ldw < D");
    hipLaunchKernelGGL(am_weight_bwd_kernel, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, s, dWl, ldw, W, inv_w, D, C, scale, dW);
    S3D_CHECK_LAUNCH("am_weight_bwd");
    return 0;
}
