// Evaluation reductions on the device ("next" row f3) and the bit-packed voxel input ("next" row f2).
//   cls_eval      argmax + total / per-class correct counts              (train_cls_voxel.py:315-329, train_cls.py:22-41)
//   partseg_eval  argmax restricted to the parts of the shape's own category, per-class seen/correct counts and the
//                 per-shape mean part IoU                                  (train_partseg.py:181-206)
//   unpack_bits   1 bit/voxel (z fastest, LSB first) -> fp32 occupancy grid [B,1,V,V,V]  (the .binvox payload after host RLE
//                 decode, utils/binvox_rw.py:117-151; 32x less H2D traffic than the int32 grids of data/modelnet40.py:40)
#include "kernels.h"

namespace {

__global__ __launch_bounds__(256) void cls_eval_kernel(const float* __restrict__ logits, int ld, const long long* __restrict__ target,
                                                       long rows, int C, int* __restrict__ pred, long long* __restrict__ counts) {
    // counts: [0] total correct, [1 .. C] per-class correct, [1 + C .. 2C] per-class total
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* l = logits + row * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < C; c += 64)
        if (l[c] > best) { best = l[c]; bi = c; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }       // first maximum, like torch.max / np.argmax
    }
    if (lane == 0) {
        const int t = (int)target[row];
        if (pred) pred[row] = bi;
        atomicAdd(reinterpret_cast<unsigned long long*>(counts + 1 + C + t), 1ull);
        if (bi == t) {
            atomicAdd(reinterpret_cast<unsigned long long*>(counts), 1ull);
            atomicAdd(reinterpret_cast<unsigned long long*>(counts + 1 + t), 1ull);
        }
    }
}

constexpr int MAX_PARTS = 16;
// one workgroup per shape.  part_range[l] = {first part id, number of parts} of the category that part label l belongs to.
__global__ __launch_bounds__(256) void partseg_eval_kernel(const float* __restrict__ logits, int ld, const long long* __restrict__ target,
                                                           int N, int num_part, const int* __restrict__ part_range,
                                                           int* __restrict__ pred, double* __restrict__ shape_iou,
                                                           int* __restrict__ shape_first, long long* __restrict__ counts) {
    // counts: [0] total correct, [1 .. P] per-part correct, [1 + P .. 2P] per-part seen
    __shared__ int inter[MAX_PARTS], uni[MAX_PARTS], seen[MAX_PARTS], corr[MAX_PARTS];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < MAX_PARTS) inter[tid] = uni[tid] = seen[tid] = corr[tid] = 0;
    __syncthreads();
    const int l0 = (int)target[(long)b * N];
    const int first = part_range[2 * l0], cnt = part_range[2 * l0 + 1];
    for (int n = tid; n < N; n += 256) {
        const float* l = logits + ((long)b * N + n) * ld + first;
        float best = l[0];
        int bi = 0;
        for (int c = 1; c < cnt; ++c)
            if (l[c] > best) { best = l[c]; bi = c; }
        const int p = bi + first;
        const int t = (int)target[(long)b * N + n];
        if (pred) pred[(long)b * N + n] = p;
        const int tl = t - first;
        if (tl >= 0 && tl < cnt) { atomicAdd(&seen[tl], 1); if (p == t) atomicAdd(&corr[tl], 1); }
        else if (t >= 0 && t < num_part)                      // a label outside the shape's category is still "seen" (:190-192)
            atomicAdd(reinterpret_cast<unsigned long long*>(counts + 1 + num_part + t), 1ull);
        if (p == t) { atomicAdd(&inter[bi], 1); atomicAdd(&uni[bi], 1); }
        else { atomicAdd(&uni[bi], 1); if (tl >= 0 && tl < cnt) atomicAdd(&uni[tl], 1); }
    }
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;                                       // fp64 like numpy: the per-shape IoU is bit-exact
        long long tc = 0;
        for (int c = 0; c < cnt; ++c) {
            s += (uni[c] == 0) ? 1.0 : (double)inter[c] / (double)uni[c];
            tc += corr[c];
            atomicAdd(reinterpret_cast<unsigned long long*>(counts + 1 + first + c), (unsigned long long)corr[c]);
            atomicAdd(reinterpret_cast<unsigned long long*>(counts + 1 + num_part + first + c), (unsigned long long)seen[c]);
        }
        atomicAdd(reinterpret_cast<unsigned long long*>(counts), (unsigned long long)tc);
        shape_iou[b] = s / cnt;
        shape_first[b] = first;
    }
}

__global__ void unpack_bits_kernel(const unsigned int* __restrict__ bits, float* __restrict__ out, long nwords) {
    for (long w = (long)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (long)gridDim.x * blockDim.x) {
        const unsigned int v = bits[w];
        f32x4* o = reinterpret_cast<f32x4*>(out + w * 32);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            f32x4 t = {(float)((v >> (4 * q)) & 1u), (float)((v >> (4 * q + 1)) & 1u), (float)((v >> (4 * q + 2)) & 1u),
                       (float)((v >> (4 * q + 3)) & 1u)};
            o[q] = t;
        }
    }
}

}  // namespace

int s3d_launch_cls_eval(const float* logits, int ld, const long long* target, long rows, int C, int* pred, long long* counts, hipStream_t s) {
    hipLaunchKernelGGL(cls_eval_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits, ld > 0 ? ld : C, target, rows, C, pred, counts);
    S3D_CHECK_LAUNCH("cls_eval");
    return 0;
}
int s3d_launch_partseg_eval(const float* logits, int ld, const long long* target, int B, int N, int num_part, const int* part_range,
                            int* pred, double* shape_iou, int* shape_first, long long* counts, hipStream_t s) {
    hipLaunchKernelGGL(partseg_eval_kernel, dim3(B), dim3(256), 0, s, logits, ld > 0 ? ld : num_part, target, N, num_part, part_range,
                       pred, shape_iou, shape_first, counts);
    S3D_CHECK_LAUNCH("partseg_eval");
    return 0;
}
int s3d_launch_unpack_bits(const unsigned int* bits, float* out, long nwords, hipStream_t s) {
    long blocks = (nwords + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(unpack_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, s, bits, out, nwords);
    S3D_CHECK_LAUNCH("unpack_bits");
    return 0;
}
