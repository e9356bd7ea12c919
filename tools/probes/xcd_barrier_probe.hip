// what does a barrier among the 32 workgroups of ONE XCD cost when nothing has to leave that XCD's L2?
// (The chip-wide grid barrier costs 7.3 us per phase, a graph's kernel boundary 1.6 us: profiles/r04_grid_barrier_probe.txt.  64 samples / 8
// XCDs = 8 samples per XCD, and every layer of a block except wgrad / Adam is row-local, so a persistent per-block launch on XCD-pinned
// sample groups would only ever need XCD-local barriers -- if they are cheap.)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/xb tools/probes/xcd_barrier_probe.hip && /tmp/xb
// 256 workgroups x 256 threads, one per CU; a workgroup's group = the XCC it really runs on (HW_REG_XCC_ID).  Each phase: write `chunk` bytes
// (phase, id), barrier among the workgroups of the same XCC, read the chunk of the NEXT workgroup of the same XCC and check it.  Variants:
//   0  plain stores, s_waitcnt vmcnt(0), relaxed agent-scope atomic arrive, sc1-load poll + s_sleep, then fence(acquire, agent) = buffer_inv sc1
//   1  the same without any invalidate on the reader (counts the stale reads: a CU's L1 is not refreshed by other CUs' stores)
//   2  sc1 (write-through) stores + sc1 loads of the payload, no fence at all (the guide's granule transport, MI355X_MICROARCH.md)
//   4  variant 0 with the arrive as a WORKGROUP-scope atomic (performed in the XCD's own L2 instead of at the memory side) and sc1-load polling
//   3  variant 0 chip-wide (one counter for all 256 workgroups, release fence before the arrive): the round-4 grid barrier, for reference
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

template <int VAR>
__global__ __launch_bounds__(256) void probe(unsigned* buf, int words, int phases, unsigned* counters, unsigned* slots, unsigned* errors, unsigned* xcc_of) {
    __shared__ unsigned s_me, s_n, s_next;
    const int bid = blockIdx.x, nwg = gridDim.x;
    const unsigned xcc = VAR == 3 ? 0u : xcc_id();   // (variant 3: one group)
    // ---- roster: every workgroup takes a slot in its XCC's list (one-off, chip-wide barrier semantics via the grid counter)
    if (threadIdx.x == 0) {
        s_me = atomicAdd(&slots[xcc], 1u);
        xcc_of[bid] = xcc;
        __hip_atomic_store(&buf[(size_t)2 * nwg * words + xcc * 64 + s_me], (unsigned)bid + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(&counters[15], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(&counters[15], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nwg) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_n = __hip_atomic_load(&slots[xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned nx = 0;
        while (nx == 0) nx = __hip_atomic_load(&buf[(size_t)2 * nwg * words + xcc * 64 + (s_me + 1) % s_n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_next = nx - 1u;
    }
    __syncthreads();
    const unsigned n = s_n, other = s_next;
    unsigned* cnt = &counters[xcc];
    for (int ph = 0; ph < phases; ++ph) {
        unsigned* mine = buf + (size_t)(ph & 1) * nwg * words + (size_t)bid * words;
        const unsigned val = (unsigned)(ph * 100000 + bid);
        for (int i = threadIdx.x * 4; i < words; i += blockDim.x * 4) {
            const u32x4 vv = {val, val, val, val};
            if (VAR == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(mine + i), "v"(vv) : "memory");
            else *reinterpret_cast<u32x4*>(mine + i) = vv;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (VAR == 3) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            if (VAR == 4) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(ph + 1) * n;
            if (VAR == 4) {
                unsigned seen = 0;
                while (seen < target) { asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(cnt) : "memory"); }
            } else {
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            }
            if (VAR == 0 || VAR == 3 || VAR == 4) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        const unsigned* theirs = buf + (size_t)(ph & 1) * nwg * words + (size_t)other * words;
        const unsigned want = (unsigned)(ph * 100000 + other);
        for (int i = threadIdx.x * 4; i < words; i += blockDim.x * 4) {
            u32x4 v;
            if (VAR == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(theirs + i) : "memory");
            else v = *reinterpret_cast<const u32x4*>(theirs + i);
            if (v[0] != want || v[3] != want) atomicAdd(errors, 1u);
        }
    }
}

int main() {
    const int phases = 400, nwg = 256;
    unsigned *buf, *counters, *slots, *errors, *xcc_of;
    CK(hipMalloc(&buf, (size_t)(2 * nwg * 16384 + 16 * 64) * 4));
    CK(hipMalloc(&counters, 64)); CK(hipMalloc(&slots, 64)); CK(hipMalloc(&errors, 4)); CK(hipMalloc(&xcc_of, nwg * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int chunks[] = {0, 4096, 65536};
    const char* names[] = {"XCD-local, acquire = buffer_inv sc1 ", "XCD-local, NO invalidate (stale?)    ", "XCD-local, sc1 stores + sc1 loads    ", "chip-wide, release + acquire (r04)   ", "XCD-local, L2 (wg-scope) atomic + inv"};
    for (int var = 0; var < 5; ++var)
        for (int c = 0; c < 3; ++c) {
            const int words = chunks[c] / 4;
            float best = 1e9f; unsigned err = 0;
            for (int it = 0; it < 4; ++it) {
                CK(hipMemsetAsync(counters, 0, 64, s)); CK(hipMemsetAsync(slots, 0, 64, s)); CK(hipMemsetAsync(errors, 0, 4, s));
                CK(hipMemsetAsync(buf + (size_t)2 * nwg * words, 0, 16 * 64 * 4, s));
                CK(hipEventRecord(e0, s));
                switch (var) {
                    case 0: hipLaunchKernelGGL(probe<0>, dim3(nwg), dim3(256), 0, s, buf, words, phases, counters, slots, errors, xcc_of); break;
                    case 1: hipLaunchKernelGGL(probe<1>, dim3(nwg), dim3(256), 0, s, buf, words, phases, counters, slots, errors, xcc_of); break;
                    case 2: hipLaunchKernelGGL(probe<2>, dim3(nwg), dim3(256), 0, s, buf, words, phases, counters, slots, errors, xcc_of); break;
                    case 4: hipLaunchKernelGGL(probe<4>, dim3(nwg), dim3(256), 0, s, buf, words, phases, counters, slots, errors, xcc_of); break;
                    default: hipLaunchKernelGGL(probe<3>, dim3(nwg), dim3(256), 0, s, buf, words, phases, counters, slots, errors, xcc_of); break;
                }
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                unsigned e; CK(hipMemcpy(&e, errors, 4, hipMemcpyDeviceToHost)); err += e;
            }
            unsigned h[256]; CK(hipMemcpy(h, xcc_of, sizeof(h), hipMemcpyDeviceToHost));
            int per[16] = {0}; for (int i = 0; i < nwg; ++i) per[h[i] & 15]++;
            printf("%s %5d B per workgroup and phase: %6.2f us / phase   wrong 16-byte reads %u   (workgroups per XCC: %d %d %d %d %d %d %d %d)\n", names[var], chunks[c],
                   best * 1e3 / phases, err, per[0], per[1], per[2], per[3], per[4], per[5], per[6], per[7]);
        }
    return 0;
}
