// Fused launches of the timm Block forward (fused_block.hip): norm1 + qkv + attention, and norm2 + fc1 + GELU.
#pragma once
#include "common.h"

struct FusedAttnArgs {
    const float* x;                       // residual stream into the block [Bb*N][D]
    const float* gamma; const float* beta; float eps;      // norm1
    const bf16_t* w_hi; const bf16_t* w_lo; const float* bias;   // attn.qkv: weight planes [3D][D], bias [3D]
    bf16_t* xn_hi; bf16_t* xn_lo;         // out: norm1(x) planes [M][D]
    float* mean; float* rstd;             // out: [M]
    bf16_t* qkv_hi;                       // out: [M][3D] (bf16 of q | k | v; the lo plane is not needed by anything downstream)
    bf16_t* att_hi; bf16_t* att_lo;       // out: softmax(q k^T * scale) v, heads concatenated, [M][D]
    float* lse;                           // out: [Bb*H][N]; lse_packed: [(b / 2) * H + h][(b & 1) * N + t], what the backward expects
                                          // when it packs two short sequences into one tile (s3d_attention_pairs_packed)
    int Bb, N, H; float scale; int lse_packed;
};
struct FusedMlpArgs {
    const float* x;                       // residual stream after the attention branch [M][D]
    const float* gamma; const float* beta; float eps;      // norm2
    const bf16_t* w_hi; const bf16_t* w_lo; const float* bias;   // mlp.fc1: weight planes [hidden][D], bias [hidden]
    bf16_t* xn_hi; bf16_t* xn_lo;         // out: norm2(x) planes [M][D]
    float* mean; float* rstd;             // out: [M]
    bf16_t* hpre; bf16_t* hact_hi; bf16_t* hact_lo;        // out: [M][hidden]
    long M; int hidden, nslice;           // nslice = hidden / 192
    int band_rows;                        // filled by the launcher: token rows per workgroup (<= 64)
};
// Backward counterpart of the attention half (round 4): attn.proj dgrad (this head's 64 input columns) -> attention backward, per
// (pair of samples, head).  Replaces the dgrad half of the proj pair launch and the attention-backward launch; the proj wgrad rides on
// the qkv pair launch as a third problem (s3d_launch_gemm_pair).
struct FusedAttnBwdArgs {
    const bf16_t* dxm; long lddxm;        // d(x_mid) as bf16 [M][D] (what LayerNorm-2 backward writes)
    const bf16_t* w_hi;                   // attn.proj weight, high plane [D out][D in]
    const bf16_t* qkv_hi;                 // saved q | k | v [M][3D]
    const float* lse;                     // as written by the forward (lse_packed: see FusedAttnArgs)
    bf16_t* dqkv;                         // out: d(q | k | v) [M][3D]
    int Bb, N, H; float scale; int lse_packed;
    // optional (round 5, bwd_gemm.hip "Row statistics"): the LayerNorm-1 backward's per-row dots of dqkv, st_s1[row] += sum_k dqkv[row][k] u[k],
    // st_s2[row] += sum_k dqkv[row][k] (qkv[row][k] - c[k]) over this head's 3 x 64 columns (fp32 atomics; u, c: [3D]).  nullptr: not computed.
    const float* st_u; const float* st_c; float* st_s1; float* st_s2;
};
bool s3d_fused_attn_bwd_ok(int Bb, int N, int D, int H);
int s3d_launch_fused_attn_bwd(const FusedAttnBwdArgs& a, int D, hipStream_t s);
bool s3d_fused_attn_ok(int Bb, int N, int D, int H);
bool s3d_fused_mlp1_ok(long M, int D, int hidden);
int s3d_launch_fused_attn(const FusedAttnArgs& a, int D, hipStream_t s);
int s3d_launch_fused_mlp1(const FusedMlpArgs& a, int D, hipStream_t s);
