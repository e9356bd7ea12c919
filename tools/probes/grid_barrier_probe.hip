// What does a phase boundary cost inside ONE persistent launch (grid barrier: release fence + one agent-scope atomic + spin + acquire
// fence) against the same boundary as a kernel boundary inside a HIP graph?  Decides whether a persistent block kernel (DESIGN.md
// section 11) can beat the >= 5 us floor of a dependent launch.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/gb tools/probes/grid_barrier_probe.hip && /tmp/gb
// Each phase: every workgroup writes `chunk` bytes (its id and the phase) and, after the boundary, reads the chunk another workgroup
// (on another XCD: +37) wrote in the previous phase and checks it -- so the boundary is also checked for cross-XCD visibility.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// the work of one phase: write my chunk for `phase`, verify the neighbour's chunk of `phase - 1`
__device__ __forceinline__ void phase_work(unsigned* buf, int words, int phase, int bid, int nwg, unsigned* errors) {
    const int other = (bid + 37) % nwg;
    if (phase > 0)
        for (int i = threadIdx.x; i < words; i += blockDim.x) {
            const unsigned v = buf[(size_t)((phase - 1) & 1) * nwg * words + (size_t)other * words + i];
            if (v != (unsigned)((phase - 1) * 100000 + other)) atomicAdd(errors, 1u);
        }
    for (int i = threadIdx.x; i < words; i += blockDim.x) buf[(size_t)(phase & 1) * nwg * words + (size_t)bid * words + i] = (unsigned)(phase * 100000 + bid);
}

__global__ __launch_bounds__(512) void persistent(unsigned* buf, int words, int phases, unsigned* counter, unsigned* errors) {
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 0) smem[0] = 1;                      // keep the allocation
    const int nwg = gridDim.x, bid = blockIdx.x;
    for (int ph = 0; ph < phases; ++ph) {
        phase_work(buf, words, ph, bid, nwg, errors);
        grid_barrier(counter, (unsigned)(ph + 1) * nwg);
    }
}

__global__ __launch_bounds__(512) void one_phase(unsigned* buf, int words, int phase, unsigned* errors) {
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 0) smem[0] = 1;
    phase_work(buf, words, phase, blockIdx.x, gridDim.x, errors);
}

int main() {
    const int phases = 200;
    unsigned *buf, *counter, *errors;
    CK(hipMalloc(&buf, 2 * 1024 * 65536 * 4));
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&errors, 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(persistent), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(one_phase), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    const int grids[] = {64, 256, 256, 512}, lds[] = {1024, 1024, 134 * 1024, 64 * 1024};
    const int chunks[] = {0, 4096, 65536};
    for (int g = 0; g < 4; ++g)
        for (int c = 0; c < 3; ++c) {
            const int nwg = grids[g], words = chunks[c] / 4;
            // ---- persistent launch with grid barriers
            float best_p = 1e9f;
            unsigned err_p = 0;
            for (int it = 0; it < 5; ++it) {
                CK(hipMemsetAsync(counter, 0, 4, s)); CK(hipMemsetAsync(errors, 0, 4, s));
                CK(hipEventRecord(e0, s));
                hipLaunchKernelGGL(persistent, dim3(nwg), dim3(512), lds[g], s, buf, words, phases, counter, errors);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best_p) best_p = ms;
                unsigned e; CK(hipMemcpy(&e, errors, 4, hipMemcpyDeviceToHost)); err_p += e;
            }
            // ---- the same phases as dependent kernels of one graph
            CK(hipMemsetAsync(errors, 0, 4, s));
            hipGraph_t graph; hipGraphExec_t exec;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int ph = 0; ph < phases; ++ph) hipLaunchKernelGGL(one_phase, dim3(nwg), dim3(512), lds[g], s, buf, words, ph, errors);
            CK(hipStreamEndCapture(s, &graph));
            CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            float best_g = 1e9f;
            for (int it = 0; it < 5; ++it) {
                CK(hipEventRecord(e0, s));
                CK(hipGraphLaunch(exec, s));
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best_g) best_g = ms;
            }
            unsigned err_g; CK(hipMemcpy(&err_g, errors, 4, hipMemcpyDeviceToHost));
            CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
            printf("grid %3d x 512 thr, %3d KB LDS, %5d B written + read per workgroup and phase: grid barrier %6.2f us / phase (errors %u)   "
                   "graph kernel boundary %6.2f us / phase (errors %u)\n",
                   nwg, lds[g] / 1024, chunks[c], best_p * 1e3 / phases, err_p, best_g * 1e3 / phases, err_g);
        }
    return 0;
}
