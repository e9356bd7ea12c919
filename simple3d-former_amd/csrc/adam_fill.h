// The optimizer update as FILLER work inside the backward launches (round 4).
//
// torch.optim.Adam.step() (train_cls_voxel.py:288) streams 40 bytes per parameter -- 0.86 GB at cfg-2, 113 - 124 us as a launch of its
// own at ~7 TB/s, 7 % of the step -- while the small-batch backward before it is a chain of latency-bound launches that leaves the
// HBM almost idle (0.4 TB/s over the step).  The gradients of block i+1 are final once its backward launches have retired, so their
// update can ride on the launches of block i: every such launch gets a few extra workgroups BEHIND its main grid (same kernel, same
// stream, no second stream / graph branch: those cost more than they hide on this runtime) that each run one
// share of the slice with exactly the arithmetic of adam_kernel (adam_update4 below is shared, results are bitwise equal).
// Non-temporal loads / stores keep the stream out of L2 / the Infinity Cache.
//
// MEASURED (profiles/r04_adam_fill.txt), and therefore OFF by default (VoxelEngine.adam_fill / S3D_ADAM_FILL=1 turns it on): the stand-alone
// update shrinks from 112.5 us to a 14.7 us launch over the left-over ranges, but every launch that carries a share gets slower by about what
// its share costs as a stream of its own -- pair kernels +1.5 / +2.4 / +0.9 us for 13 / 18 / 9 MB, LayerNorm backward +0.9, attention backward
// +0.8, and the next forward's kernels +0.4 each -- 1.744 -> 1.754 .. 1.768 ms per step whatever the number of filler workgroups (192 .. 2048)
// or groups per thread.  The "idle" HBM is not free bandwidth for these launches: their critical paths are chains of L2 misses through the
// same memory pipeline, and a stream beside them lengthens every one of those round trips.
#pragma once
#include <string.h>

#include "common.h"
#include "s3d_hip.h"

struct AdamFill {                    // kernel argument: one share of a parameter slice (all pointers already offset to the share)
    float *p, *g, *m, *v;
    bf16_t *hi, *lo;
    const S3dAdamState* st;
    long n4;                         // float4 groups in the share
    int zero_grad;
    int blocks;                      // filler workgroups appended to the launch's grid (0 = none)
};
inline AdamFill adam_fill_none() { return AdamFill{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0}; }

// one float4 group of torch.optim.Adam (no weight decay, no amsgrad) + split-bf16 plane refresh + gradient zeroing
__device__ __forceinline__ void adam_update4(long i, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                             bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, const bf16_t* __restrict__ gw, float b1, float b2,
                                             float eps, float gs, float step_size, float bc2s, int zero_grad) {
    // no FMA contraction here: this body is inlined into several kernels (adam_kernel, adam_ranges_kernel, the filler branches of the
    // LayerNorm / attention / GEMM-pair backward kernels) and every one of them has to round exactly alike -- the compiler's choice of
    // which multiply-adds to fuse depends on the surrounding code (measured: a third of the moments differed in the last bit)
#pragma clang fp contract(off)
    const f32x4 Pn = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + i);
    float4 P = make_float4(Pn[0], Pn[1], Pn[2], Pn[3]), G;
    if (gw) {                                       // bf16 wire format: the all-reduced gradient arrives as bf16 (uniform branch)
        union { uint2 u; bf16_t h[4]; } W;
        W.u = reinterpret_cast<const uint2*>(gw)[i];
        G = make_float4(bf2f(W.h[0]), bf2f(W.h[1]), bf2f(W.h[2]), bf2f(W.h[3]));
    } else {
        const f32x4 Gn = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
        G = make_float4(Gn[0], Gn[1], Gn[2], Gn[3]);
    }
    const f32x4 Mn = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m) + i);
    const f32x4 Vn = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v) + i);
    float4 M = make_float4(Mn[0], Mn[1], Mn[2], Mn[3]), Vv = make_float4(Vn[0], Vn[1], Vn[2], Vn[3]);
    float pp[4] = {P.x, P.y, P.z, P.w}, gg[4] = {G.x * gs, G.y * gs, G.z * gs, G.w * gs};
    float mm[4] = {M.x, M.y, M.z, M.w}, vv[4] = {Vv.x, Vv.y, Vv.z, Vv.w};
    union { uint2 u; bf16_t h[4]; } H, L;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mm[k] = mm[k] * b1 + (1.f - b1) * gg[k];
        vv[k] = vv[k] * b2 + (1.f - b2) * gg[k] * gg[k];
        const float denom = sqrtf(vv[k]) / bc2s + eps;
        pp[k] -= step_size * (mm[k] / denom);
        split_bf16(pp[k], H.h[k], L.h[k]);
    }
    // p / m / v / g are touched once per step: non-temporal, so that the 0.7 GB they stream does not push the weight planes and
    // the saved activations out of L2 / the Infinity Cache
    __builtin_nontemporal_store(f32x4{pp[0], pp[1], pp[2], pp[3]}, reinterpret_cast<f32x4*>(p) + i);
    __builtin_nontemporal_store(f32x4{mm[0], mm[1], mm[2], mm[3]}, reinterpret_cast<f32x4*>(m) + i);
    __builtin_nontemporal_store(f32x4{vv[0], vv[1], vv[2], vv[3]}, reinterpret_cast<f32x4*>(v) + i);
    if (zero_grad) __builtin_nontemporal_store(f32x4{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<f32x4*>(g) + i);
    if (hi) reinterpret_cast<uint2*>(hi)[i] = H.u;
    if (lo) reinterpret_cast<uint2*>(lo)[i] = L.u;
}

// body of a filler workgroup: fb = its index among the f.blocks filler workgroups of the launch
__device__ __forceinline__ void adam_fill_run(const AdamFill& f, int fb) {
    const S3dAdamState* st = f.st;
    const float b1 = st->beta1, b2 = st->beta2, eps = st->eps, gs = st->grad_scale, step_size = st->step_size, bc2s = st->bc2_sqrt;
    const long stride = (long)f.blocks * blockDim.x;
    for (long i = (long)fb * blockDim.x + threadIdx.x; i < f.n4; i += stride)
        adam_update4(i, f.p, f.g, f.m, f.v, f.hi, f.lo, nullptr, b1, b2, eps, gs, step_size, bc2s, f.zero_grad);
}

// ---- host side: the ranges that are ready to be updated, handed out in shares to the launches that follow
struct AdamFillQueue {
    const S3dAdamFill* base = nullptr;          // arena base pointers
    long seg_off4[4] = {0, 0, 0, 0}, seg_n4[4] = {0, 0, 0, 0};
    int nseg = 0, cur = 0;
    long pos4 = 0, total4 = 0;
    int share16 = 0;                            // the next launch's share in sixteenths of total4 (16 = everything left)
    void reset() { nseg = cur = 0; pos4 = total4 = 0; share16 = 0; }
    void push(long off, long n) {               // [off, off + n) floats of the arena; multiples of 4
        if (n <= 0 || nseg >= 4) return;
        seg_off4[nseg] = off / 4; seg_n4[nseg] = n / 4; total4 += n / 4; ++nseg;
    }
    bool empty() const { return cur >= nseg; }
    // the next share for a launch whose workgroups have `threads` threads; blocks == 0 when there is nothing to hand out
    AdamFill take(int threads) {
        AdamFill f = adam_fill_none();
        if (base == nullptr || empty() || share16 <= 0) return f;
        long want = share16 >= 16 ? (1L << 60) : (total4 * share16 + 15) / 16;
        const long left = seg_n4[cur] - pos4;
        const long n4 = want < left ? want : left;
        const long o = (seg_off4[cur] + pos4) * 4;
        f.p = base->p + o; f.g = base->g + o; f.m = base->m + o; f.v = base->v + o;
        f.hi = base->hi ? base->hi + o : nullptr; f.lo = base->lo ? base->lo + o : nullptr;
        f.st = base->state; f.n4 = n4; f.zero_grad = base->zero_grad;
        long blocks = (n4 + (long)threads * 4 - 1) / ((long)threads * 4);       // ~4 float4 groups per thread (1 / 2 / 4 per thread, 192 .. 2048 workgroups: within 0.5 % of each other)
        if (blocks > 192) blocks = 192;          // fewer make the share the launch's critical path (96: +3 %, 32: +24 %, 4: 4.3x the step)
        f.blocks = (int)blocks;
        pos4 += n4;
        if (pos4 >= seg_n4[cur]) { ++cur; pos4 = 0; }
        share16 = 0;
        return f;
    }
};
