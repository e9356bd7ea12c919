// gfx950 v_permlane32_swap: what __builtin_amdgcn_permlane32_swap(v, v, false, false) returns per lane.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pp tools/probes/permlane_probe.hip && /tmp/pp
// Result (MI355X): r[0][l] = v[l & 31] (the lower half, in both halves), r[1][l] = v[32 + (l & 31)] (the upper half) -- so
// op(r[0], r[1]) is the reduction across the two halves, in every lane, with ONE VALU instruction instead of a ds_bpermute.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* o) {
    const int v = threadIdx.x * 10;
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
}
int main() {
    int* d; int h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l : {0, 1, 31, 32, 33, 63}) printf("lane %2d: r0 %4d r1 %4d\n", l, h[l], h[64 + l]);
    return 0;
}
