// Round 5 probe: the forward residual GEMM of cfg-2 (x_out = x_mid + h @ W2^T + b: M = 1664 rows, N = 384, K = 1536, split bf16 = three MFMAs
// per product) WITHOUT LDS: both operands are k-contiguous (NT), so a 16 x 32-k MFMA fragment is 16 rows x 64 bytes of global memory -- every
// wave loads its fragments straight into registers, the four waves of a 32 x 32 tile take every fourth k-step and add their accumulators
// through LDS at the end.  No LDS-DMA (the per-CU rate that paces gemm_nt_dma_kernel<.., 32, 32>: DESIGN section 6.11), no barrier in the k-loop.
// Question: is the vector-memory path faster than ~1 KB of LDS-DMA per 45 cycles per CU for this access pattern (16 rows per instruction)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/resid_direct_probe tools/probes/resid_direct_probe.hip && tools/probes/resid_direct_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
    const bf16_t *A_hi, *A_lo, *B_hi, *B_lo;
    const float *bias, *R;
    float* C;
    int M, N, K;
};

// TM x TN tile per workgroup (TM = 16 FM, TN = 16 FN), four waves split the k-steps; PD = k-steps of loads in flight per wave
template <int FM, int FN, int PD>
__global__ __launch_bounds__(256) void resid_direct(const Args p) {
    __shared__ f32x4 red[3][FM * FN][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int ntx = p.N / (16 * FN);
    const int m0 = (blockIdx.x / ntx) * 16 * FM, n0 = (blockIdx.x % ntx) * 16 * FN;
    const int nsteps = p.K / 32;                                         // k-steps of 32; wave w takes w, w + 4, ..
    const bf16_t* pa[FM][2];
    const bf16_t* pb[FN][2];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const long row = min(m0 + i * 16 + r16, p.M - 1);
        pa[i][0] = p.A_hi + row * p.K + kg * 8; pa[i][1] = p.A_lo + row * p.K + kg * 8;
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const long row = n0 + j * 16 + r16;
        pb[j][0] = p.B_hi + row * p.K + kg * 8; pb[j][1] = p.B_lo + row * p.K + kg * 8;
    }
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 ra[PD][FM][2], rb[PD][FN][2];
    auto load = [&](int slot, int s) {
        const long o = (long)s * 32;
#pragma unroll
        for (int i = 0; i < FM; ++i) { ra[slot][i][0] = *reinterpret_cast<const u32x4*>(pa[i][0] + o); ra[slot][i][1] = *reinterpret_cast<const u32x4*>(pa[i][1] + o); }
#pragma unroll
        for (int j = 0; j < FN; ++j) { rb[slot][j][0] = *reinterpret_cast<const u32x4*>(pb[j][0] + o); rb[slot][j][1] = *reinterpret_cast<const u32x4*>(pb[j][1] + o); }
    };
    const int mine = (nsteps - wave + 3) / 4;                            // this wave's k-steps
#pragma unroll
    for (int u = 0; u < PD - 1; ++u)
        if (u < mine) load(u, wave + 4 * u);
#pragma unroll 1
    for (int t0 = 0; t0 < mine; t0 += PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int t = t0 + u;
            if (t >= mine) break;
            if (t + PD - 1 < mine) load((u + PD - 1) % PD, wave + 4 * (t + PD - 1));
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, ra[u][i][0]), al = __builtin_bit_cast(bf16x8, ra[u][i][1]);
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, rb[u][j][0]), bl = __builtin_bit_cast(bf16x8, rb[u][j][1]);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, acc[i][j], 0, 0, 0);
                }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) red[wave - 1][i * FN + j][lane] = acc[i][j];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                f32x4 v = acc[i][j] + red[0][i * FN + j][lane] + red[1][i * FN + j][lane] + red[2][i * FN + j][lane];
                const int m = m0 + i * 16 + r16, n = n0 + j * 16 + kg * 4;          // lane = row, four consecutive columns
                if (m < p.M) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n), r = *reinterpret_cast<const f32x4*>(p.R + (long)m * p.N + n);
                    *reinterpret_cast<f32x4*>(p.C + (long)m * p.N + n) = v + b + r;
                }
            }
    }
}

static bf16_t f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
static float bf2f(bf16_t h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <int FM, int FN, int PD>
static float run(const Args& a, int reps) {
    const unsigned grid = (unsigned)(((a.M + 16 * FM - 1) / (16 * FM)) * (a.N / (16 * FN)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((resid_direct<FM, FN, PD>), dim3(grid), dim3(256), 0, 0, a);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((resid_direct<FM, FN, PD>), dim3(grid), dim3(256), 0, 0, a);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps * 1e3f;
}

int main() {
    const int M = 1664, N = 384;
    for (int K : {1536, 384}) {
        std::vector<float> A((size_t)M * K), B((size_t)N * K), bias(N), R((size_t)M * N);
        srand(3);
        for (auto& v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
        for (auto& v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
        for (auto& v : bias) v = rand() / (float)RAND_MAX;
        for (auto& v : R) v = rand() / (float)RAND_MAX;
        std::vector<bf16_t> Ah(A.size()), Al(A.size()), Bh(B.size()), Bl(B.size());
        for (size_t i = 0; i < A.size(); ++i) { Ah[i] = f2bf(A[i]); Al[i] = f2bf(A[i] - bf2f(Ah[i])); }
        for (size_t i = 0; i < B.size(); ++i) { Bh[i] = f2bf(B[i]); Bl[i] = f2bf(B[i] - bf2f(Bh[i])); }
        Args a;
        bf16_t *dAh, *dAl, *dBh, *dBl; float *db, *dR, *dC;
        CK(hipMalloc(&dAh, Ah.size() * 2)); CK(hipMalloc(&dAl, Al.size() * 2)); CK(hipMalloc(&dBh, Bh.size() * 2)); CK(hipMalloc(&dBl, Bl.size() * 2));
        CK(hipMalloc(&db, N * 4)); CK(hipMalloc(&dR, R.size() * 4)); CK(hipMalloc(&dC, R.size() * 4));
        CK(hipMemcpy(dAh, Ah.data(), Ah.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dAl, Al.data(), Al.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dBh, Bh.data(), Bh.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dBl, Bl.data(), Bl.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, bias.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dR, R.data(), R.size() * 4, hipMemcpyHostToDevice));
        a.A_hi = dAh; a.A_lo = dAl; a.B_hi = dBh; a.B_lo = dBl; a.bias = db; a.R = dR; a.C = dC; a.M = M; a.N = N; a.K = K;
        // correctness of one variant against a double-precision product of the same split operands (sampled)
        hipLaunchKernelGGL((resid_direct<2, 2, 2>), dim3((M / 32) * (N / 32)), dim3(256), 0, 0, a);
        std::vector<float> C(R.size());
        CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int s = 0; s < 400; ++s) {
            const int m = rand() % M, n = rand() % N;
            double ref = bias[n] + R[(size_t)m * N + n];
            for (int k = 0; k < K; ++k) {
                const double ah = bf2f(Ah[(size_t)m * K + k]), al = bf2f(Al[(size_t)m * K + k]), bh = bf2f(Bh[(size_t)n * K + k]), bl = bf2f(Bl[(size_t)n * K + k]);
                ref += ah * bh + ah * bl + al * bh;
            }
            worst = fmax(worst, fabs(C[(size_t)m * N + n] - ref));
        }
        printf("K = %4d: max abs error of 400 sampled outputs %.2e\n", K, worst);
        printf("  32 x 32 tiles (624 workgroups), prefetch 2 / 3 / 4 k-steps: %6.2f %6.2f %6.2f us\n", run<2, 2, 2>(a, 300), run<2, 2, 3>(a, 300), run<2, 2, 4>(a, 300));
        printf("  32 x 64 tiles (312 workgroups), prefetch 2 / 3:             %6.2f %6.2f us\n", run<2, 4, 2>(a, 300), run<2, 4, 3>(a, 300));
        printf("  64 x 32 tiles (312 workgroups), prefetch 2 / 3:             %6.2f %6.2f us\n", run<4, 2, 2>(a, 300), run<4, 2, 3>(a, 300));
        printf("  16 x 32 tiles (1248 workgroups), prefetch 2 / 4:            %6.2f %6.2f us\n", run<1, 2, 2>(a, 300), run<1, 2, 4>(a, 300));
    }
    return 0;
}
