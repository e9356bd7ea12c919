// bf16 / split-bf16 MFMA GEMM for gfx950: C[M,N] = op(A) * op(B) with fused epilogues (see gemm.hip).
// The argument block is the public C-ABI struct (include/s3d_hip.h).
#pragma once
#include "common.h"
#include "s3d_hip.h"

enum {
    EPI_BF16_BIAS = S3D_EPI_BF16_BIAS,  // O_hi/O_lo = split(acc*alpha + bias[n])                     (qkv, generic linear)
    EPI_GELU = S3D_EPI_GELU,            // pre = acc + bias; aux = bf16(pre); O = split(gelu(pre))    (mlp.fc1)
    EPI_RESID = S3D_EPI_RESID,          // C = acc (+ bias[n]) + R[m][n]; optional bf16 copy in O_hi  (attn.proj, mlp.fc2, residual dgrads)
    EPI_TOKEN = S3D_EPI_TOKEN,          // C = acc*alpha + (m%ntok==0 ? cls[n] : bias[n]) + pos[(m%ntok)*N+n]
    EPI_F32 = S3D_EPI_F32,              // C = acc*alpha (+ bias[n])                                  (dgrad -> LayerNorm bwd)
    EPI_DGELU = S3D_EPI_DGELU,          // O_hi = bf16(acc * gelu'(aux[m][n]))                        (mlp.fc2 dgrad)
    EPI_ATOMIC = S3D_EPI_ATOMIC,        // atomicAdd(C[m][n], acc*alpha); optional bias_grad          (wgrad, split-K)
    EPI_RELU = S3D_EPI_RELU,            // O = split(relu(acc + bias)), aux = bf16(pre)               (encoder linear1)
    EPI_DRELU = S3D_EPI_DRELU,          // O_hi = bf16(aux > 0 ? acc : 0)                             (encoder linear2 dgrad)
};
typedef S3dGemmArgs GemmArgs;

// ta/tb: operand stored k-major.  splitk <= 0: automatic (wgrad only).  Returns 0 on success.
int s3d_launch_gemm(bool ta, bool tb, bool split, int epi, const GemmArgs& a, int splitk, hipStream_t stream);

// row-stream kernel (rowstream_gemm.hip): forward NT, split precision, F32 epilogue, K <= 96, N <= 192, >= 32768 rows
bool s3d_rowstream_gemm_ok(bool split, int epi, const GemmArgs& a);
int s3d_launch_rowstream_gemm(const GemmArgs& a, hipStream_t stream);

// forward NT RESID launch that would run the fused LayerNorm epilogue (GemmArgs::ln_tickets), see gemm.hip
bool s3d_gemm_ln_fusable(bool split, const GemmArgs& a);
int s3d_gemm_pick_tile(int M, int N, int splitk, bool split);

// dgrad (k-major B, epilogue epi_a) and the wgrad that consumes the same dy, fused into one launch (see gemm.hip)
struct AdamFillQueue;        // adam_fill.h: optimizer shares riding on the launch as filler workgroups (LDS-DMA pair kernel only)
int s3d_launch_gemm_pair(int epi_a, const GemmArgs& dgrad, const GemmArgs& wgrad, hipStream_t stream, AdamFillQueue* fill = nullptr,
                         const GemmArgs* wgrad2 = nullptr);

// bench-only timing of every GEMM launch with HIP events on the launch stream (see gemm.hip)
void s3d_gemm_prof_enable(bool on);
int s3d_gemm_prof_collect(double* rows, int cap);
void s3d_gemm_prof_skip(long long key);
long long s3d_gemm_prof_skip_get();
// ... and for launches outside gemm.hip: skipped? / events around the launch when profiling is enabled
bool s3d_prof_skipped(long long key);
void s3d_prof_begin(long long key, double flops, hipStream_t s);
void s3d_prof_end(hipStream_t s);
