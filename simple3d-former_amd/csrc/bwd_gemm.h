// Backward GEMMs of the small-batch block stack (bwd_gemm.hip): split-K dgrad with consumer-side reduction, grouped full-k wgrad.
#pragma once
#include "common.h"
#include "s3d_hip.h"

// dx planes = dy @ W: C[slice] (slice s at a.C + s * slice_stride) = A[:, k-slice s] @ B[k-slice s, :]; A = a.A_hi [M][K] k-contiguous,
// B = a.B_hi [K][N] k-major, fp32 output, no bias.  The caller's consumer adds the nslice planes (S3dLnBwdArgs::dy_parts).
bool s3d_dgrad_splitk_ok(const S3dGemmArgs& a);
int s3d_dgrad_splitk_slices(int K, int want);      // slices a request for `want` (1 .. 4) really gives (whole 64-tiles per slice)
int s3d_launch_dgrad_splitk(const S3dGemmArgs& a, int nslice, long slice_stride, hipStream_t s);
// dW_i (+)= alpha * dy_i^T x_i, db_i (+)= alpha * colsum(dy_i) for n <= 24 layers that share the row count K, one launch
// fills (optional): up to 6 optimizer shares (adam_fill.h) that ride on the launch as filler workgroups behind the tiles
struct AdamFill;
int s3d_launch_wgrad_group(const S3dWgradItem* items, int n, int K, float alpha, int accumulate, hipStream_t s, const AdamFill* fills = nullptr,
                           int nfill = 0);
// the fused backward chain (see bwd_gemm.hip, "Row statistics"): fc2 dgrad * gelu' + row statistics; dgrad + LayerNorm backward epilogue;
// the weights-only vectors u / c of the row statistics
int s3d_launch_dgrad_dgelu(const S3dGemmArgs& a, const S3dRowStats* st, hipStream_t s);
int s3d_launch_dgrad_lnbwd(const S3dGemmArgs& a, const S3dLnBwdArgs& ln, const S3dRowStats* st, hipStream_t s);
int s3d_launch_ln_aux(const S3dLnAuxLayer* layers, int n, int D, hipStream_t s);
// the same for 192-wide layers with WHOLE rows per workgroup (64 x 192 tiles): the row statistics come from the tile itself
bool s3d_dgrad_lnrows_ok(const S3dGemmArgs& a, const S3dLnBwdArgs& ln);
int s3d_launch_dgrad_lnrows(const S3dGemmArgs& a, const S3dLnBwdArgs& ln, hipStream_t s);
