// How fast can many workgroups accumulate fp32 partial tiles into a shared [rows][384] residual stream with global atomics?
// (round 3: the fused block kernels add their proj / fc2 partials straight into the residual stream -- the residual add IS the
// split-K reduction.)   hipcc --offload-arch=gfx950 -O2 -munsafe-fp-atomics -o /tmp/ap tools/probes/atomic_resid_probe.hip && /tmp/ap
//   mode 0: MFMA natural layout -- a wave instruction adds 4 rows x 16 consecutive floats (64-byte runs)
//   mode 1: row-contiguous -- a wave instruction adds 64 consecutive floats of one row (256 bytes)
//   mode 2: plain stores in the mode-1 pattern (what the same bytes cost without the read-modify-write)
//   mode 3: MFMA swapped layout -- a lane adds 4 consecutive floats of one row (4 scalar atomics per 16 bytes)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ __launch_bounds__(512) void k(float* x, int rows_per_wg, int sharers, int D) {
    const int band = blockIdx.x / sharers;
    float* base = x + (long)band * rows_per_wg * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const float v = 1.0f + (float)(blockIdx.x % sharers);
    if (MODE == 0) {
        // wave w owns 16-row x 16-col blocks w, w + nw, ...
        const int nblk = (rows_per_wg / 16) * (D / 16);
        for (int b = wave; b < nblk; b += nw) {
            const int r0 = (b / (D / 16)) * 16, c0 = (b % (D / 16)) * 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) unsafeAtomicAdd(base + (long)(r0 + (lane >> 4) * 4 + r) * D + c0 + (lane & 15), v);
        }
    } else if (MODE == 3) {
        const int nblk = (rows_per_wg / 16) * (D / 16);
        for (int b = wave; b < nblk; b += nw) {
            const int r0 = (b / (D / 16)) * 16, c0 = (b % (D / 16)) * 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) unsafeAtomicAdd(base + (long)(r0 + (lane & 15)) * D + c0 + (lane >> 4) * 4 + r, v);
        }
    } else {
        const int per_row = D / 64;
        for (int i = wave; i < rows_per_wg * per_row; i += nw) {
            float* p = base + (long)(i / per_row) * D + (i % per_row) * 64 + lane;
            if (MODE == 1) unsafeAtomicAdd(p, v); else *p = v;
        }
    }
}

template <int MODE>
static void run(const char* name, float* x, int wgs, int rows_per_wg, int sharers, int threads) {
    const int D = 384;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 12; ++it) {
        hipMemsetAsync(x, 0, (size_t)(wgs / sharers) * rows_per_wg * D * 4, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(threads), 0, 0, x, rows_per_wg, sharers, D);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2 && ms < best) best = ms;
    }
    const double mb = (double)wgs * rows_per_wg * D * 4 / 1e6;
    printf("%-34s wgs %4d x %3d rows, %d sharers, %3d thr: %7.2f us  (%6.1f MB of adds, %5.2f TB/s)\n", name, wgs, rows_per_wg, sharers, threads,
           best * 1e3, mb, mb / 1e6 / (best * 1e-3));
}

int main() {
    float* x;
    hipMalloc(&x, 64 << 20);
    // MLP half: 26 bands x 8 hidden slices of 64 rows; attention half: 64 samples x 6 heads of 32 (26 valid) rows
    run<0>("mlp  natural (4 rows x 64 B)", x, 208, 64, 8, 512);
    run<1>("mlp  row-contiguous (256 B)", x, 208, 64, 8, 512);
    run<3>("mlp  swapped (lane = 4 floats)", x, 208, 64, 8, 512);
    run<2>("mlp  plain stores", x, 208, 64, 8, 512);
    run<0>("mlp  natural, 256 thr", x, 208, 64, 8, 256);
    run<0>("mlp  natural, 4 sharers 32 rows", x, 208, 32, 4, 512);
    run<0>("mlp  natural, 16 sharers 64 rows", x, 416, 64, 16, 256);
    run<0>("attn natural", x, 384, 32, 6, 256);
    run<1>("attn row-contiguous", x, 384, 32, 6, 256);
    run<2>("attn plain stores", x, 384, 32, 6, 256);
    run<0>("one sharer (no contention)", x, 208, 64, 1, 512);
    return 0;
}
