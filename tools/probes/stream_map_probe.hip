// Which thread -> address map streams a [R][C] fp32 tensor into a [R][C] bf16 tensor fastest?  (round 3: the BatchNorm passes of the
// point path sit at 2.5 - 3 TB/s whatever their grid, unroll or cache hints; Adam's flat float4 walk reaches 6 TB/s.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/smp tools/probes/stream_map_probe.hip && /tmp/smp
//   mode 0: BatchNorm map -- thread = (row lane, channel quad), C/4 quads, 256 / (C/4) rows per workgroup trip, grid-stride over rows
//   mode 1: flat map -- thread walks float4 index i = blockIdx * 256 + tid, += gridDim * 256 (channel = (4 i) % C recomputed per trip)
//   mode 2: BatchNorm map with the row lanes padded so that every wave starts a row (64 lanes = one row of up to 256 channels)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

__device__ inline unsigned pk(float a, float b) { return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u); }

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, unsigned short* __restrict__ y, long rows, int C, const float* __restrict__ sc) {
    if (MODE == 0) {
        const int c4 = C >> 2, rpb = 256 / c4, q = threadIdx.x % c4, sub = threadIdx.x / c4;
        if (sub >= rpb) return;
        const f32x4 s = *reinterpret_cast<const f32x4*>(sc + 4 * q);
        for (long r = (long)blockIdx.x * rpb + sub; r < rows; r += (long)gridDim.x * rpb) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + 4 * q);
            *reinterpret_cast<u32x2*>(y + r * C + 4 * q) = u32x2{pk(v[0] * s[0], v[1] * s[1]), pk(v[2] * s[2], v[3] * s[3])};
        }
    } else if (MODE == 1) {
        const long n4 = rows * C / 4;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
            const int c = (int)((i * 4) % C);
            const f32x4 s = *reinterpret_cast<const f32x4*>(sc + c);
            const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
            reinterpret_cast<u32x2*>(y)[i] = u32x2{pk(v[0] * s[0], v[1] * s[1]), pk(v[2] * s[2], v[3] * s[3])};
        }
    } else {
        const int c4 = C >> 2, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane >= c4) return;
        const f32x4 s = *reinterpret_cast<const f32x4*>(sc + 4 * lane);
        for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + 4 * lane);
            *reinterpret_cast<u32x2*>(y + r * C + 4 * lane) = u32x2{pk(v[0] * s[0], v[1] * s[1]), pk(v[2] * s[2], v[3] * s[3])};
        }
    }
}

// BatchNorm-backward apply, ingredient by ingredient (map of mode 0):
//   STEP 1: + a second input stream (dy, bf16)        STEP 2: + ReLU mask, the dx formula, round-to-nearest-even bf16
//   STEP 3: + the per-thread prologue (4 x float4 constants, 8 fp64 sums -> means)     STEP 4: + 64-bit row-pitch arithmetic from kernel args
__device__ inline unsigned short f2bf_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
template <int STEP>
__global__ __launch_bounds__(256) void kb(const float* __restrict__ x, const unsigned short* __restrict__ dy, unsigned short* __restrict__ dx,
                                          long rows, int C, int ld, int lddy, int lddx, const float* __restrict__ cst,
                                          const double* __restrict__ sums, double inv_rows) {
    const int c4 = C >> 2, rpb = 256 / c4, q = threadIdx.x % c4, sub = threadIdx.x / c4;
    if (sub >= rpb) return;
    const int c = 4 * q;
    f32x4 m = {0, 0, 0, 0}, rs = {1, 1, 1, 1}, a = {1, 1, 1, 1}, b = {0, 0, 0, 0}, s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
    if (STEP >= 3) {
        m = *reinterpret_cast<const f32x4*>(cst + c); rs = *reinterpret_cast<const f32x4*>(cst + 256 + c);
        const f32x4 g = *reinterpret_cast<const f32x4*>(cst + 512 + c), be = *reinterpret_cast<const f32x4*>(cst + 768 + c);
        for (int i = 0; i < 4; ++i) {
            a[i] = rs[i] * g[i]; b[i] = be[i] - m[i] * a[i];
            s1[i] = (float)(sums[c + i] * inv_rows); s2[i] = (float)(sums[C + c + i] * inv_rows);
        }
    }
    const long pitch_x = STEP >= 4 ? ld : C, pitch_dy = STEP >= 4 ? lddy : C, pitch_dx = STEP >= 4 ? lddx : C;
    for (long r = (long)blockIdx.x * rpb + sub; r < rows; r += (long)gridDim.x * rpb) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + r * pitch_x + c);
        const u32x2 w = *reinterpret_cast<const u32x2*>(dy + r * pitch_dy + c);
        const f32x4 d = {__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xffff0000u), __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xffff0000u)};
        unsigned short o[4];
        for (int i = 0; i < 4; ++i) {
            if (STEP >= 2) {
                const float gq = (xv[i] * a[i] + b[i] > 0.f) ? d[i] : 0.f;
                const float xh = (xv[i] - m[i]) * rs[i];
                o[i] = f2bf_rne(a[i] * (gq - s1[i] - xh * s2[i]));
            } else {
                o[i] = (unsigned short)(__float_as_uint(xv[i] + d[i]) >> 16);
            }
        }
        *reinterpret_cast<u32x2*>(dx + r * pitch_dx + c) = u32x2{(unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16)};
    }
}
template <int STEP>
static void runb(const char* name, const float* x, const unsigned short* dy, unsigned short* dx, long rows, int C, const float* cst,
                 const double* sums, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 8; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kb<STEP>, dim3(grid), dim3(256), 0, 0, x, dy, dx, rows, C, C, C, C, cst, sums, 1.0 / rows);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2 && ms < best) best = ms;
    }
    const double gb = (double)rows * C * 8 / 1e9;
    printf("%-44s C %3d grid %6d: %7.1f us  %5.2f TB/s\n", name, C, grid, best * 1e3, gb / best);
}

template <int MODE>
static void run(const char* name, const float* x, unsigned short* y, long rows, int C, const float* sc, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 8; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, x, y, rows, C, sc);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2 && ms < best) best = ms;
    }
    const double gb = (double)rows * C * 6 / 1e9;
    printf("%-44s C %3d grid %6d: %7.1f us  %5.2f TB/s\n", name, C, grid, best * 1e3, gb / best);
}

int main() {
    const long rows = 524288;
    float *x, *sc; unsigned short* y;
    hipMalloc(&x, rows * 256 * 4); hipMalloc(&y, rows * 256 * 2); hipMalloc(&sc, 1024);
    hipMemset(x, 0, rows * 256 * 4); hipMemset(sc, 0, 1024);
    for (int C : {192, 128, 256, 64}) {
        for (int grid : {2048, 8192, 32768}) {
            run<0>("BatchNorm map (row lane, quad)", x, y, rows, C, sc, grid);
            run<1>("flat float4 walk", x, y, rows, C, sc, grid);
            run<2>("one row per wave", x, y, rows, C, sc, grid);
        }
    }
    unsigned short* dyb; float* cst; double* sums;
    hipMalloc(&dyb, rows * 256 * 2); hipMalloc(&cst, 4096); hipMalloc(&sums, 4096);
    hipMemset(dyb, 0, rows * 256 * 2); hipMemset(cst, 0, 4096); hipMemset(sums, 0, 4096);
    for (int grid : {8192, 32768}) {
        runb<1>("apply step 1: x + dy -> dx", x, dyb, y, rows, 192, cst, sums, grid);
        runb<2>("apply step 2: + mask, formula, rne", x, dyb, y, rows, 192, cst, sums, grid);
        runb<3>("apply step 3: + prologue", x, dyb, y, rows, 192, cst, sums, grid);
        runb<4>("apply step 4: + runtime pitches", x, dyb, y, rows, 192, cst, sums, grid);
    }
    return 0;
}
