// Probe: what does ds_read_b64_tr_b16 return?  LDS holds e at 16-bit element e; candidate per-lane address maps are tried.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void probe(unsigned short* out, int mode, int pitch) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x, t = l & 15, g = l >> 4;
    int elem;                                   // element index this lane's address points at
    if (mode == 0) elem = l * 4;                                    // lane-linear 8-byte chunks
    else if (mode == 1) elem = (t >> 2) * pitch + (t & 3) * 4 + g * 16;   // [4 rows][16 cols] block per group, lane -> (row t/4, chunk t%4)
    else elem = (t & 3) * pitch + (t >> 2) * 4 + g * 16;            // lane -> (row t%4, chunk t/4)
    const unsigned addr = (unsigned)(uintptr_t)(lds + elem);        // LDS byte address
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)(v >> (16 * j));
}

int main() {
    unsigned short* d;
    hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 3; ++mode)
        for (int pitch : {16, 64}) {
            if (mode == 0 && pitch != 16) continue;
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode, pitch);
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            printf("mode %d pitch %d\n", mode, pitch);
            for (int l = 0; l < 64; ++l) {
                if (l % 16 < 6 || l % 16 == 15) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
            }
        }
    return 0;
}
