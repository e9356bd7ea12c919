// Argument blocks + launchers of the non-GEMM kernels (ln.hip, misc.hip).
#pragma once
#include "common.h"
#include "s3d_hip.h"

typedef S3dLnArgs LnArgs;
typedef S3dLnBwdArgs LnBwdArgs;
int s3d_launch_ln_fwd(const LnArgs& a, hipStream_t s);
int s3d_launch_ln_grad_reduce(const float* const* partial, float* const* dgamma, float* const* dbeta, int n_ln, int nblk, int D,
                              hipStream_t s);
int s3d_launch_ln_grad_reduce_rows(const float* const* partial, float* const* dgamma, float* const* dbeta, const int* rows, int n_ln, int D,
                                   hipStream_t s);
struct AdamFillQueue;        // adam_fill.h: optimizer shares that ride on this launch as filler workgroups (nullptr: none)
int s3d_launch_ln_bwd(const LnBwdArgs& a, hipStream_t s, AdamFillQueue* fill = nullptr);

// ---- tokenizer patch gather ("fold"): voxel grid -> GEMM A operand (split-bf16 planes) ----
enum { FOLD_ZMEAN = 0, FOLD_NAIVE = 1, FOLD_PATCH = 2, FOLD_PATCH_GROUP = 3 };
typedef S3dFoldArgs FoldArgs;
int s3d_launch_fold(const FoldArgs& a, hipStream_t s);

// d(pos_embed)[t][:] += sum_g dx[g*ntok+t][:], d(cls) += rows t==0, d(conv bias) += rows t>=1
typedef S3dPosGradArgs PosGradArgs;
int s3d_launch_patchify(const float* img, bf16_t* a_hi, bf16_t* a_lo, long lda, int B, int C, int H, int W, int p, hipStream_t s);
int s3d_launch_posgrad(const PosGradArgs& a, hipStream_t s);

// pass-2 token assembly of group_embed: out = cat(cls, src) + pos  (and the gather of its backward)
int s3d_launch_assemble(const float* src, const float* cls, const float* pos, float* out, long B, int n, int D, hipStream_t s);
int s3d_launch_assemble_bwd(const float* dout, float* dsrc, long B, int n, int D, hipStream_t s);

// fp32 [rows][cols] -> split-bf16 planes with row pitch ld_out (pad columns untouched)
int s3d_launch_split(const float* src, bf16_t* hi, bf16_t* lo, long rows, long cols, long ld_out, hipStream_t s);

// ---- classification head (fp32 VALU; tiny) ----
typedef S3dHeadArgs HeadArgs;
int s3d_launch_head_fwd(const HeadArgs& a, hipStream_t s);
int s3d_launch_head_bwd(const HeadArgs& a, hipStream_t s);

// mean cross-entropy (optionally class-weighted) forward + d(logits)
typedef S3dCeArgs CeArgs;
int s3d_launch_ce(const CeArgs& a, hipStream_t s);

int s3d_launch_head_loss(const S3dHeadLossArgs& a, hipStream_t s);

// ---- fused Adam over a flat fp32 arena (+ split-bf16 shadow planes) ----
typedef S3dAdamState AdamState;
int s3d_launch_adam_begin(AdamState* st, hipStream_t s);
int s3d_launch_adam_apply(float* p, float* g, float* m, float* v, bf16_t* hi, bf16_t* lo, long n, const AdamState* st,
                          int zero_grad, const bf16_t* g_wire, int max_blocks, hipStream_t s);
int s3d_launch_adam(float* p, float* g, float* m, float* v, bf16_t* hi, bf16_t* lo, long n, AdamState* st,
                    int zero_grad, const bf16_t* g_wire, hipStream_t s);
int s3d_launch_adam_ranges(float* p, float* g, float* m, float* v, bf16_t* hi, bf16_t* lo, const long* ranges, int n, const AdamState* st,
                           int zero_grad, hipStream_t s);
int s3d_launch_pack_bf16(const float* src, bf16_t* dst, long n, hipStream_t s);
int s3d_launch_l2norm_rows_fwd(const float* x, long ldx, long rows, int D, float* inv_norm, bf16_t* hi, bf16_t* lo, long ldo, hipStream_t s);
int s3d_launch_l2norm_rows_bwd(const float* dxn, long lddxn, const float* x, long ldx, const float* inv_norm, long rows, int D, float* dx, long lddx,
                               hipStream_t s);
int s3d_launch_am_weight_fwd(const float* W, int D, int C, float scale, float* Wl, int ldw, float* inv_w, hipStream_t s);
int s3d_launch_am_weight_bwd(const float* dWl, int ldw, const float* W, const float* inv_w, int D, int C, float scale, float* dW, hipStream_t s);

// ---- point-cloud operators (points.hip) ----
int s3d_launch_fps(const float* xyz, long xyz_ld, const long long* start, int B, int N, int npoint, int* out_idx,
                   float* new_xyz, hipStream_t s);
int s3d_launch_knn(const float* query, const float* ref, int B, int S, int N, int K, int* out_idx, float* out_w, hipStream_t s);
int s3d_launch_group_gather(const float* xyz, const float* new_xyz, const float* feats, const int* idx, int B, int N, int S,
                            int K, int C, bf16_t* a_hi, bf16_t* a_lo, int lda, hipStream_t s);
int s3d_launch_group_scatter(const float* dA, int ldd, const int* idx, int B, int N, int S, int K, int C, float* dfeats, hipStream_t s);
int s3d_launch_group_project_fwd(const S3dGroupProjArgs& a, hipStream_t s);
int s3d_launch_group_project_bwd(const S3dGroupProjArgs& a, hipStream_t s);
int s3d_launch_neighbor_csr(const int* idx, int B, int N, int S, int K, int* inv_off, int* inv_rows, hipStream_t s);
int s3d_launch_bn_fwd(const S3dBnArgs& a, hipStream_t s);
int s3d_launch_bn_bwd(const S3dBnArgs& a, hipStream_t s);
int s3d_launch_interp3(const float* f1, int S, const float* f2, const int* idx, const float* w, int B, int N, int C, float* out,
                       hipStream_t s);
int s3d_launch_interp3_bwd(const float* dout, const int* idx, const float* w, int B, int S, int N, int C, float* df1, hipStream_t s);
int s3d_launch_mean_points(const float* x, int B, int N, int C, float* out, hipStream_t s);
int s3d_launch_bcast_rows(const float* x, int N, int C, long rows, float scale, float* y, hipStream_t s);
int s3d_launch_pack_rows(const float* x, int C, int ldx, long rows, bf16_t* hi, bf16_t* lo, int ldo, hipStream_t s);
int s3d_launch_add_inplace(float* a, const float* b, long n, hipStream_t s);
int s3d_launch_sgd(float* p, float* g, float* buf, bf16_t* hi, bf16_t* lo, long n, float lr, float momentum, float grad_scale,
                   int* step_counter, const float* hyper, hipStream_t s);

// ---- evaluation metrics + bit-packed voxel input (metrics.hip) ----
int s3d_launch_cls_eval(const float* logits, int ld, const long long* target, long rows, int C, int* pred, long long* counts, hipStream_t s);
int s3d_launch_partseg_eval(const float* logits, int ld, const long long* target, int B, int N, int num_part, const int* part_range,
                            int* pred, double* shape_iou, int* shape_first, long long* counts, hipStream_t s);
int s3d_launch_unpack_bits(const unsigned int* bits, float* out, long nwords, hipStream_t s);
