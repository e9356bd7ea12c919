// Probe: how fast can a CU pull GEMM-tile-shaped operand traffic out of L2 on gfx950?  The cfg-2 forward GEMMs spend their k-loop
// at ~30-43 B/clk/CU (tools/timeline_probe.py); this strips the kernel down to its loads -- same tile -> address map, same 1 KB
// DMA pieces (8 rows x 128 B), same stage ring / counted vmcnt / one barrier per k-tile, NO ds_read, NO MFMA -- and sweeps tile
// shape, ring depth and the load flavour (LDS-DMA vs. global_load into registers).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/dma_bw_probe tools/probes/dma_bw_probe.hip && tools/probes/dma_bw_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
struct Rsrc { unsigned int w[4]; };
__device__ __forceinline__ void bdma16(const void* base, int nbytes, unsigned char* lds_dst, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, nbytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds_dst, 16, voff, soff, 0, 0);
#endif
}

struct Args { const unsigned short *A, *B; long ld; int M, N, K, xcd_map; unsigned long long* cyc; u32x4* sink; };

// MODE 0: LDS-DMA ring; MODE 1: global_load_dwordx4 into registers (one stage prefetched), xor-folded
template <int BM, int BN, int NS, int NPL, int MODE>
__global__ __launch_bounds__(256) void stream_kernel(const Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int STAGE = NPL * (BM + BN) * 128, PPW = STAGE / 4096;
    static_assert(STAGE % 4096 == 0, "whole pieces per wave");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile = blockIdx.x;
    const int ntx = p.N / BN, nty = (p.M + BM - 1) / BM, ntile = ntx * nty;
    if (p.xcd_map) { const int q = ntile >> 3, r = ntile & 7, x = tile & 7, i = tile >> 3; tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i; }
    const int m0 = (tile / ntx) * BM, n0 = (tile % ntx) * BN;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem);
    const unsigned short* gp[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int piece = wave * PPW + j;
        constexpr int PA = BM / 8, PB = BN / 8;
        int q = piece;
        const bool isB = q >= NPL * PA;
        if (isB) q -= NPL * PA;
        const int pl = q / (isB ? PB : PA), rb = q % (isB ? PB : PA);
        const int r = rb * 8 + (lane >> 3), c = lane & 7;
        const int sw = (r ^ (r >> 3)) & 7;
        const long plane = (long)pl * (isB ? (long)p.N : (long)p.M) * p.ld;
        const int row = isB ? min(n0 + r, p.N - 1) : min(m0 + r, p.M - 1);
        gp[j] = (isB ? p.B : p.A) + plane + (long)row * p.ld + ((c ^ sw) << 3);
    }
    const int ntiles = p.K / 64;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (MODE == 2) {
        // LDS-DMA through BUFFER loads: resource descriptor (SGPRs) + 32-bit per-lane byte offset + scalar k offset, instead of a
        // 64-bit per-lane address (global_load_lds)
        const int nA = (int)((long)NPL * p.M * p.ld * 2), nB = (int)((long)NPL * p.N * p.ld * 2);
        int voff[PPW];
        bool isb[PPW];
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            isb[j] = (wave * PPW + j) >= NPL * (BM / 8);
            voff[j] = (int)((const unsigned char*)gp[j] - (const unsigned char*)(isb[j] ? p.B : p.A));
        }
        auto issue = [&](int t) {
#pragma unroll
            for (int j = 0; j < PPW; ++j) {
                unsigned char* dst = smem + (t % NS) * STAGE + (wave * PPW + j) * 1024;
                if (isb[j]) bdma16(p.B, nB, dst, voff[j], t * 128);
                else bdma16(p.A, nA, dst, voff[j], t * 128);
            }
        };
#pragma unroll
        for (int u = 0; u < NS - 1; ++u) if (u < ntiles) issue(u);
        for (int t = 0; t < ntiles; ++t) {
            if (t + NS - 1 <= ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t + NS - 1 < ntiles) issue(t + NS - 1);
        }
    } else if constexpr (MODE == 0) {
        auto issue = [&](int t) {
            const unsigned dst = lds0 + (unsigned)((t % NS) * STAGE + wave * PPW * 1024);
#pragma unroll
            for (int j = 0; j < PPW; ++j) glds16(gp[j] + (long)t * 64, dst + j * 1024);
        };
#pragma unroll
        for (int u = 0; u < NS - 1; ++u) if (u < ntiles) issue(u);
        for (int t = 0; t < ntiles; ++t) {
            if (t + NS - 1 <= ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t + NS - 1 < ntiles) issue(t + NS - 1);
        }
    } else {
        u32x4 acc = {0u, 0u, 0u, 0u}, cur[PPW], nxt[PPW];
#pragma unroll
        for (int j = 0; j < PPW; ++j) cur[j] = *reinterpret_cast<const u32x4*>(gp[j]);
        for (int t = 0; t < ntiles; ++t) {
            const int tn = min(t + 1, ntiles - 1);
#pragma unroll
            for (int j = 0; j < PPW; ++j) nxt[j] = *reinterpret_cast<const u32x4*>(gp[j] + (long)tn * 64);
#pragma unroll
            for (int j = 0; j < PPW; ++j) { acc ^= cur[j]; cur[j] = nxt[j]; }
            __syncthreads();
        }
        if (acc[0] == 0x12345678u && acc[1] == 7u) p.sink[blockIdx.x * 256 + tid] = acc;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) p.cyc[blockIdx.x] = t1 - t0;
}

template <int BM, int BN, int NS, int NPL, int MODE>
void run(const char* name, Args a, int reps = 30) {
    constexpr int STAGE = NPL * (BM + BN) * 128;
    const int lds = MODE != 1 ? NS * STAGE : 0;
    auto kern = stream_kernel<BM, BN, NS, NPL, MODE>;
    if (lds > 160 * 1024) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int nwg = (a.N / BN) * ((a.M + BM - 1) / BM);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, a);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, a);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> cyc(nwg);
    hipMemcpy(cyc.data(), a.cyc, nwg * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0; for (auto c : cyc) mean += (double)c; mean /= nwg;
    const double bytes = (double)nwg * STAGE * (a.K / 64);
    const double us = ms * 1e3 / reps;
    const double wg_per_cu = nwg / 256.0;
    printf("%-4s tile %3dx%-3d NS=%d planes=%d mode=%s map=%d  N=%4d K=%4d  %4d WGs  %7.2f us/launch  %6.1f MB  %6.2f TB/s  | in-WG: %7.0f cyc, %5.1f B/clk/WG -> x%.2f WG/CU = %5.1f B/clk/CU\n",
           name, BM, BN, NS, NPL, MODE == 0 ? "dma " : MODE == 2 ? "bdma" : "regs", a.xcd_map, a.N, a.K, nwg, us, bytes / 1e6, bytes / us / 1e6, mean,
           (double)STAGE * (a.K / 64) / mean, wg_per_cu, (double)STAGE * (a.K / 64) / mean * wg_per_cu);
}

int main() {
    const int M = 1664, NMAX = 1536, KMAX = 1536;
    unsigned short *A, *B; unsigned long long* cyc; u32x4* sink;
    hipMalloc(&A, (size_t)2 * M * KMAX * 2); hipMalloc(&B, (size_t)2 * NMAX * KMAX * 2);
    hipMemset(A, 1, (size_t)2 * M * KMAX * 2); hipMemset(B, 2, (size_t)2 * NMAX * KMAX * 2);
    hipMalloc(&cyc, 8192 * 8); hipMalloc(&sink, 8192 * 256 * 16);
    const bool full = getenv("PROBE_FULL") != nullptr;
    for (int map = 1; map >= (full ? 0 : 1); --map) {
        Args fc2{A, B, 1536, M, 384, 1536, map, cyc, sink}, fc1{A, B, 384, M, 1536, 384, map, cyc, sink}, pr{A, B, 384, M, 384, 384, map, cyc, sink};
        printf("--- global_load_lds (dma) vs buffer_load ... lds (bdma): fc2 shape (N=384, K=1536), split planes, xcd_map=%d\n", map);
        run<32, 32, 2, 2, 0>("fc2", fc2); run<32, 32, 2, 2, 2>("fc2", fc2); run<32, 32, 4, 2, 2>("fc2", fc2);
        run<64, 64, 2, 2, 0>("fc2", fc2); run<64, 64, 2, 2, 2>("fc2", fc2); run<64, 64, 3, 2, 0>("fc2", fc2); run<64, 64, 3, 2, 2>("fc2", fc2);
        run<128, 128, 2, 2, 0>("fc2", fc2); run<128, 128, 2, 2, 2>("fc2", fc2);
        printf("--- fc1 shape (N=1536, K=384)\n");
        run<32, 64, 2, 2, 0>("fc1", fc1); run<32, 64, 2, 2, 2>("fc1", fc1);
        run<64, 64, 2, 2, 0>("fc1", fc1); run<64, 64, 2, 2, 2>("fc1", fc1);
        run<128, 128, 2, 2, 0>("fc1", fc1); run<128, 128, 2, 2, 2>("fc1", fc1);
        if (!full) continue;
        printf("--- fc2 shape, ring depth and register loads\n");
        run<32, 32, 3, 2, 0>("fc2", fc2); run<32, 32, 4, 2, 0>("fc2", fc2); run<32, 32, 2, 2, 1>("fc2", fc2);
        run<32, 64, 2, 2, 0>("fc2", fc2); run<32, 64, 3, 2, 0>("fc2", fc2); run<64, 64, 2, 2, 1>("fc2", fc2);
        run<64, 128, 2, 2, 0>("fc2", fc2);
        printf("--- fc1 shape\n");
        run<32, 64, 3, 2, 0>("fc1", fc1); run<32, 64, 2, 2, 1>("fc1", fc1); run<64, 64, 3, 2, 0>("fc1", fc1); run<64, 128, 2, 2, 0>("fc1", fc1);
        printf("--- proj shape (N=384, K=384)\n");
        run<32, 32, 2, 2, 0>("proj", pr); run<32, 32, 4, 2, 0>("proj", pr); run<32, 32, 2, 2, 1>("proj", pr); run<64, 64, 2, 2, 0>("proj", pr);
        printf("--- plain bf16 (one plane) fc2 shape\n");
        run<32, 32, 2, 1, 0>("fc2", fc2); run<64, 64, 2, 1, 0>("fc2", fc2); run<64, 64, 3, 1, 0>("fc2", fc2); run<64, 64, 4, 1, 0>("fc2", fc2);
    }
    return 0;
}
