/* s3d_hip.h -- C ABI of libs3d_hip.so: the MI355X (gfx950) kernels of the Simple3D-Former training hot path.
 *
 * The reference (VITA-Group/Simple3D-Former) is pure Python on PyTorch + timm and has no FFI of its own; each entry
 * point below names the reference operator it replaces (paths relative to the reference repository root) and is what
 * a ctypes / cffi binding on the reference side would bind (INTEGRATION.md shows that stub).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer owned by the caller (PyTorch's allocator in the
 *     shipped host code).  The library never allocates or frees device memory and keeps no global state besides a
 *     per-process error string.
 *   - every call only ENQUEUES work on `stream` (no synchronisation, graph-capture safe).
 *   - return value 0 = success; non-zero = failure, text via s3d_last_error_string().  Never aborts.
 *   - bf16 tensors are passed as uint16_t*.  "hi/lo" pairs are split-bf16 planes: x ~= hi + lo;
 *     lo may be NULL wherever documented (plain-bf16 mode).
 *   - matrices are row-major; ld* are row pitches in ELEMENTS.
 */
#ifndef S3D_HIP_H
#define S3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* s3d_stream_t; /* hipStream_t */

/* ------------------------------------------------------------------------------------------------ library */
int s3d_version(void);                       /* 10000*major + 100*minor + patch */
const char* s3d_last_error_string(void);
size_t s3d_sizeof(const char* struct_name);  /* ABI self-check for foreign-language mirrors; 0 if unknown */

/* ------------------------------------------------------------------------------------------------ GEMM
 * Replaces every nn.Linear / Conv-as-GEMM contraction of the path: timm Attention.qkv / .proj and Mlp.fc1 / .fc2
 * (timm==0.3.2, called from models/vit_3d_2d_pretrain.py:466-468), Conv3d patchify of
 * models/embed_layer_3d_modality.py:160-161,175 and their autograd backward (train_cls_voxel.py:287). */
enum S3dGemmEpi {
    S3D_EPI_BF16_BIAS = 0, S3D_EPI_GELU = 1, S3D_EPI_RESID = 2, S3D_EPI_TOKEN = 3, S3D_EPI_F32 = 4,
    S3D_EPI_DGELU = 5, S3D_EPI_ATOMIC = 6, S3D_EPI_RELU = 7, S3D_EPI_DRELU = 8
};
typedef struct S3dGemmArgs {
    const uint16_t* A_hi; const uint16_t* A_lo; long lda;
    const uint16_t* B_hi; const uint16_t* B_lo; long ldb;
    int M, N, K;
    int kchunk;                       /* filled by the library */
    const float* bias;
    const float* R; long ldr;
    float* C; long ldc;
    uint16_t* O_hi; uint16_t* O_lo; long ldo;
    uint16_t* aux; long ldaux;
    float alpha;
    const float* cls; const float* pos; int ntok;
    float* bias_grad;
    /* optional dropout on the epilogue value (RESID: on acc+bias before the residual add; RELU: on the activation; DRELU: on
     * the gradient): keep(m*N + n) = hash(*drop_seed, drop_site, index) >= drop_thr, kept values scaled by drop_scale.
     * drop_thr = 0 disables it.  The seed lives in device memory so that HIP-graph replays see a fresh value. */
    const unsigned long long* drop_seed; int drop_site; unsigned int drop_thr; float drop_scale;
    /* optional fused LayerNorm of the RESID output (forward NT GEMMs whose N is the whole model dimension: attn.proj -> norm2,
     * mlp.fc2 -> the next block's norm1).  ln_tickets = one zero-initialised int per row band of the launch (>= ceil(M / 32)
     * entries; the library leaves them zero): the workgroup whose tile completes a band LAST normalises the band's rows of C
     * (nn.LayerNorm(eps), gamma / beta) into the split planes ln_hi / ln_lo (row pitch ld_ln) and writes ln_mean / ln_rstd [M].
     * NULL disables it.  s3d_gemm_ln_fusable() tells whether a launch will take the fused path (else call s3d_layernorm_fwd). */
    int* ln_tickets;
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    uint16_t* ln_hi; uint16_t* ln_lo; long ld_ln;
    float* ln_mean; float* ln_rstd;
    /* optional (forward NT GEMM, epilogue F32, N % 8 == 0): col_sums[0 .. N) += column sums of the output, col_sums[N .. 2N) += column
     * sums of its squares over the M rows (fp64 atomics; the caller zeroes it).  They are the batch statistics of the train-mode
     * BatchNorm that follows a point-path convolution (S3dBnArgs::have_sums), so that tensor is not read a second time to get them. */
    double* col_sums;
    /* optional low plane of `aux` (the saved pre-activation): GELU / RELU epilogues then store aux_lo = bf16(pre - aux), DGELU / DRELU read
     * pre = aux + aux_lo.  Only the split-precision backward mode (S3dBlockScratch::dx_a_lo) sets it. */
    uint16_t* aux_lo;
} S3dGemmArgs;
/* ta / tb: operand stored k-major.  (0,0) forward "x @ W^T"; (0,1) dgrad "dy @ W"; (1,1) wgrad "dy^T @ x" (split-K,
 * fp32 atomics into C, optional bias_grad = column sums of dy).  split: three-MFMA split-bf16 product -- the forward's precision;
 * for (0,1) / (1,1) it is the split-precision BACKWARD of the parity mode (both operands need their lo planes, no split-K). */
int s3d_gemm(int ta, int tb, int split, int epi, const S3dGemmArgs* args, int splitk, s3d_stream_t stream);
/* One launch = the dgrad "dx = dy @ W" (NN, epilogue epi_dgrad: F32 / DGELU / DRELU / RESID / BF16_BIAS) of a Linear layer AND the wgrad
 * "dW += dy^T x, db += colsum(dy)" (TN, split-K fp32 atomics) that consumes the same dy: workgroups [0, nA) of the grid run the first
 * problem, the rest the second.  This is how s3d_block_bwd issues every Linear backward of the small-batch configurations (each half
 * alone under-fills the chip at 1664 token rows); shapes that want 128x128 tiles fall back to two launches.  The library picks the
 * wgrad's k-split (kchunk of both argument structs is ignored). */
int s3d_gemm_pair(int epi_dgrad, const S3dGemmArgs* dgrad, const S3dGemmArgs* wgrad, s3d_stream_t stream);
/* the same launch carrying a SECOND wgrad (any layer whose dy and x are ready: s3d_block_bwd puts attn.proj's wgrad on the qkv pair
 * launch once the proj dgrad has moved into the fused attention backward) */
int s3d_gemm_pair3(int epi_dgrad, const S3dGemmArgs* dgrad, const S3dGemmArgs* wgrad, const S3dGemmArgs* wgrad2, s3d_stream_t stream);
/* Round 5 -- the same two halves as launches of their own, for callers that keep dy until several layers' wgrads can share one launch
 * (s3d_blocks_bwd with S3dBlockScratch::wg_ring does):
 *   s3d_gemm_dgrad_splitk: "dx = dy @ W" (A = dy [M][K] k-contiguous, B = W [K][N] k-major, fp32 C, alpha; no bias) on 64 x 64 tiles with
 *   the k range cut into nslice slices of whole 64-tiles; slice s STORES its partial product at C + s * slice_stride (elements), and the
 *   consumer adds the planes (s3d_layernorm_bwd: S3dLnBwdArgs::dy_parts / dy_part_stride).  No atomics: bitwise reproducible.
 *   s3d_gemm_wgrad_group: for n <= 48 layers with the same row count K: dW_i[out][in] (+)= alpha * dy_i[K][out]^T x_i[K][in] and, when db is
 *   set, db_i[out] (+)= alpha * colsum(dy_i); 128 x 128 output tiles over the full K, one workgroup per tile, plain read-modify-write
 *   (accumulate = 1) or overwrite (accumulate = 0): no split-K, no atomics, bitwise reproducible.  out, in, ld_dy, ld_x multiples of 8.
 *   s3d_gemm_dgrad_splitk_slices: the slices a request for `want` (1 .. 4) really gives for this K -- pass THAT count to both calls. */
typedef struct S3dWgradItem {
    const uint16_t* dy; long ld_dy; int out;
    const uint16_t* x; long ld_x; int in;
    float* dW; long ldw;
    float* db;                        /* optional */
} S3dWgradItem;
/* Round 5 -- the LayerNorm backward as an EPILOGUE of the dgrad that produces its input (s3d_blocks_bwd with S3dBlockScratch::ln_aux does):
 * the two per-row reductions of nn.LayerNorm's backward, s1 = mean_n(dy gamma) and s2 = mean_n(dy gamma xh), are linear in the gradient dz
 * the dgrad multiplies (dy = dz @ W):  s1 = sum_k dz[k] u[k],  s2 = (1/D) sum_k dz[k] (pre[k] - c[k])  with the weights-only vectors
 * u[k] = mean_n(W[k][n] gamma[n]), c[k] = b[k] + sum_n W[k][n] beta[n] (s3d_ln_aux) and the layer's own saved pre-activation `pre`.  The
 * producer of dz accumulates both dots per row (rs1, rs2: fp32 atomics; s3d_gemm_dgrad_dgelu for mlp.fc2's dgrad, the fused attention
 * backward for dqkv), and s3d_gemm_dgrad_lnbwd applies  dx = rstd (dy gamma - s1 - xh s2) + dres  element-wise to its own output tile
 * (+ bf16 copy, + column partials of dgamma / dbeta: partial[row tile][2][D], >= ceil(M / 64) rows, or atomics).  Two launches fewer per block. */
typedef struct S3dRowStats {
    const float* u; const float* c;    /* producer: weights-only vectors of the Linear whose dgrad FOLLOWS, indexed by this launch's output column */
    float* rs1; float* rs2;            /* [M]: accumulated by the producer (zero before it), read by the consumer (rs2 is divided by D there) */
    float* zero_buf; int zero_n;       /* optional: zero_n floats the launch clears (a statistics buffer that is idle while it runs) */
} S3dRowStats;
typedef struct S3dLnAuxLayer {
    const uint16_t* w_hi; const uint16_t* w_lo; const float* bias;     /* Linear weight [K][D] as split planes, bias [K] (may be NULL) */
    const float* gamma; const float* beta;                             /* the LayerNorm in front of it */
    float* u; float* c; int K;                                         /* out: [K] each */
} S3dLnAuxLayer;
int s3d_ln_aux(const S3dLnAuxLayer* layers, int n, int D, s3d_stream_t stream);       /* n <= 32 layers, one launch */
struct S3dLnBwdArgs;
int s3d_gemm_dgrad_dgelu(const S3dGemmArgs* args, const S3dRowStats* stats, s3d_stream_t stream);   /* O_hi = bf16(A @ B * gelu'(aux)); stats may be NULL */
int s3d_gemm_dgrad_lnbwd(const S3dGemmArgs* args, const struct S3dLnBwdArgs* ln, const S3dRowStats* stats, s3d_stream_t stream);
/* The same for 192-wide layers at many rows (deit_tiny: the point path's 257 / 513-token sequences): 64 x 192 tiles hold WHOLE rows, so the row
 * statistics come from the tile itself -- no producer, any dgrad that feeds a LayerNorm backward (fc1 -> norm2, qkv -> norm1).  N == 192,
 * K % 32 == 0; ln->partial (if set) receives one row of [2][192] column sums per 64-row tile (ln->partial_blocks >= ceil(M / 64)).
 * Replaces a dgrad with the F32 epilogue + s3d_layernorm_bwd (models/3DViT/model.py:318-320 backward). */
int s3d_gemm_dgrad_lnrows(const S3dGemmArgs* args, const struct S3dLnBwdArgs* ln, s3d_stream_t stream);
int s3d_gemm_dgrad_splitk_slices(int K, int want);
int s3d_gemm_dgrad_splitk(const S3dGemmArgs* args, int nslice, long slice_stride, s3d_stream_t stream);
int s3d_gemm_wgrad_group(const S3dWgradItem* items, int n, int K, float alpha, int accumulate, s3d_stream_t stream);
/* 1 if s3d_gemm(0, 0, split, S3D_EPI_RESID, args, ...) with args->ln_tickets set would run the fused LayerNorm epilogue */
int s3d_gemm_ln_fusable(int split, const S3dGemmArgs* args);
/* 1 if a forward (0,0) F32-epilogue launch of this shape accumulates S3dGemmArgs::col_sums (128x128 tiles, N % 8 == 0); the caller
 * otherwise leaves col_sums NULL and lets s3d_batchnorm_fwd compute its own statistics. */
int s3d_gemm_col_sums_ok(int split, int M, int N);

/* ---- NOT part of the operator ABI: measurement / test aids (s3d_prof_*, s3d_cov_*, s3d_graph_marker / s3d_graph_events_at_markers,
 * s3d_debug_*).  They exist for bench.py and the test suite, may change between versions, and a foreign-language binding should not bind them.
 * Measurement aid (bench.py roofline leg): when enabled, every GEMM launch is bracketed by HIP events recorded on the
 * launch stream.  s3d_prof_collect synchronises those events and fills rows of 4 doubles
 * {kernel key, launches, total ms, total algorithmic flops (2*M*N*K)}; key digits = 1|BM|BN|ta|tb|split|epilogue for a
 * single problem, 2|BM dgrad|BM wgrad|000|epilogue for a fused dgrad + wgrad launch (decoded in bench.py).  Returns the number
 * of distinct kernels.  Must be off during graph capture.  s3d_prof_event_overhead measures what the bracket itself adds:
 * the median time between two events recorded back-to-back on the stream with nothing in between (microseconds). */
int s3d_prof_enable(int on);
int s3d_prof_collect(double* rows, int cap);
/* Launch coverage (test aid): while enabled, every kernel launch notes "family:instantiation key".  s3d_cov_enable(1) clears and starts,
 * s3d_cov_enable(0) stops; s3d_cov_collect copies the newline-separated list "family:key:launches" (NUL-terminated, truncated to cap)
 * and returns the buffer size the full list needs.  tests/test_gpu_zz_coverage.py uses it to prove that every kernel instantiation a
 * benched training step dispatches has also run inside a test that compares with the oracle. */
int s3d_cov_enable(int on);
long s3d_cov_collect(char* buf, long cap);
/* Deterministic mode (default: environment S3D_DETERMINISTIC=1, else off).  On: no reduction combines partial sums from several
 * workgroups with fp32 atomics -- wgrads run without split-K, the token / conv-bias gradients, the loss and the final-norm
 * gamma / beta gradients take single-writer kernels -- so a training step (train_cls_voxel.py:277-288) is bitwise reproducible
 * run to run.  Slower; meant for parity and trajectory tests.  Process-wide; set it before capturing graphs.  The point path's training
 * step (train_cls.py / train_partseg.py) is NOT covered: its BatchNorm batch statistics and scatter / interpolation gradients are summed
 * with atomics in either mode (profiles/r06_repro_audit.txt: forward bit-reproducible, gradients to 2.5e-3 of the largest). */
int s3d_set_deterministic(int on);
int s3d_get_deterministic(void);
int s3d_prof_event_overhead(s3d_stream_t stream, double* microseconds);
/* Difference timing: while key != 0, launches of the GEMM instantiation with that key (as reported by s3d_prof_collect) are
 * suppressed (they return 0 without enqueuing anything).  bench.py captures the training step once with and once without the
 * dominant kernel and takes (t_full - t_without) / launches as that kernel's duration inside the busy, gap-free graph. */
int s3d_prof_skip(double key);
/* the key currently suppressed (0 = none): bench.py reads it right behind its timed loop and reports "kernels_suppressed" */
double s3d_prof_skip_get(void);

/* ------------------------------------------------------------------------------------------------ LayerNorm
 * nn.LayerNorm(eps=1e-6) of timm Block.norm1/.norm2 and VisionTransformer.norm (vit_3d_2d_pretrain.py:287,469). */
typedef struct S3dLnArgs {
    const float* x; long ldx;
    long rows; int D; float eps;
    const float* gamma; const float* beta;
    uint16_t* out_hi; uint16_t* out_lo; float* out_f32; long ldo;
    float* mean; float* rstd;
} S3dLnArgs;
typedef struct S3dLnBwdArgs {
    const float* dy; long lddy;
    const float* x; long ldx;
    const float* mean; const float* rstd; const float* gamma;
    const float* dres; long lddres;
    float* dx; long lddx;
    uint16_t* dx_bf; long lddxbf;
    float* dgamma; float* dbeta;
    long rows; int D;
    /* partial != NULL: instead of atomically adding its column sums to dgamma / dbeta, workgroup b of the partial_blocks
     * launched STORES them at partial[b][0][0..D) (gamma) and partial[b][1][0..D) (beta); s3d_layernorm_grad_reduce adds the
     * partial_blocks rows to dgamma / dbeta afterwards (same-address fp32 atomics from ~100 workgroups cost a third of the
     * kernel at cfg-2; without them the grid can also be finer) */
    float* partial; int partial_blocks;
    /* optional dropout mask applied to the bf16 copy only (the branch gradient of a post-norm residual), see S3dGemmArgs */
    const unsigned long long* drop_seed; int drop_site; unsigned int drop_thr; float drop_scale;
    uint16_t* dx_bf_lo;            /* optional: low plane of dx_bf (dx ~= dx_bf + dx_bf_lo), same pitch -- split-precision backward mode */
    /* dy arrives as dy_parts partial planes (the k-slices of s3d_gemm_dgrad_splitk): plane k at dy + k * dy_part_stride (elements), same
     * pitch; they are added in plane order while they are loaded (at most 4).  0 or 1: dy is the whole gradient. */
    int dy_parts; long dy_part_stride;
} S3dLnBwdArgs;
int s3d_layernorm_fwd(const S3dLnArgs* args, s3d_stream_t stream);
int s3d_layernorm_bwd(const S3dLnBwdArgs* args, s3d_stream_t stream);
/* dgamma[c] += sum_b partial[b][0][c], dbeta[c] += sum_b partial[b][1][c] for n_ln independent LayerNorms in one launch
 * (n_ln <= 64; partial[i] is [nblk][2][D]) */
int s3d_layernorm_grad_reduce(const float* const* partial, float* const* dgamma, float* const* dbeta, int n_ln, int nblk, int D,
                              s3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------ attention
 * softmax(q k^T * hd^-0.5) v of timm Attention.forward (math restated at visualize_attention_map_voxel.py:125-138)
 * and of nn.MultiheadAttention inside group_embed (vit_3d_2d_pretrain.py:381,479; seq-first: sb=1, st=Nb). */
typedef struct S3dAttnArgs {
    const uint16_t* qkv_hi; const uint16_t* qkv_lo; long ld;
    uint16_t* out_hi; uint16_t* out_lo; long ldo;
    float* lse;
    int Bb, H, N, D;
    long sb, st;
    float scale;
    const uint16_t* dout; long lddo;
    uint16_t* dqkv; long lddq;
    float* delta;
    /* optional dropout on the attention weights, index ((b*H + h)*N + q)*N + key (one hash per pair of adjacent keys, 16 bits each:
     * csrc/common.h drop_keep_attn), see S3dGemmArgs */
    const unsigned long long* drop_seed; int drop_site; unsigned int drop_thr; float drop_scale;
    /* block-diagonal attention: 0 = off; otherwise a query attends only to the keys of its own segment of `seg` consecutive
     * tokens (needs seg <= N <= 2*seg and N <= 32).  The launchers use it themselves to put TWO short sequences (N <= 16,
     * contiguous in memory: group_embed pass 1 has N = 15) into one 32-row MFMA tile; lse / delta are then laid out for the
     * packed problem (Bb/2, 2N), consistently between forward and backward. */
    int seg;
    /* split-precision backward (parity mode): when dqkv_lo is set, s3d_attention_bwd takes q, k, v, dout and out as hi + lo pairs
     * (qkv_lo, dout_lo, out_lo required), evaluates the backward in fp32 and stores d(qkv) as the pair dqkv / dqkv_lo.  Slow reference
     * kernels, any N / head dim / layout; delta is still written. */
    const uint16_t* dout_lo; uint16_t* dqkv_lo;
    /* optional 1-bit copy of the attention-weight dropout mask for LONG sequences (the cooperative kernels: N >= 192, head dim < 256;
     * ignored otherwise): [Bb*H][ceil(N/32)][ceil(N/32)][32] words, word q of tile (query tile, key tile) = keep bits of the tile's 32
     * keys for query row q.  s3d_attention_fwd writes it while it evaluates the hash, s3d_attention_bwd (given the SAME buffer, seed and
     * shape) reads it instead of evaluating the hash twice more.  NULL: every kernel evaluates the hash. */
    unsigned int* drop_mask;
    /* split forward of LONG sequences (head dim 192, the cooperative kernel): 1 = the probabilities enter the P V product as ONE bf16 plane
     * (P_hi (V_hi + V_lo): two MFMAs per product instead of three; the row sum, the log-sum-exp and the Q K^T product stay as they are).  The
     * rounding of P is 2^-9 relative per weight; through a softmax-weighted average it reaches the layer's output at <= 2^-9 of the spread of V
     * and the logits of the group_embed model at 1 - 3e-6 (tools/r6/pv_plain_probe.py: diffuse and sharpened attention, bar 1e-3).  The seq-first
     * encoder layer of group_embed sets it (vit_3d_2d_pretrain.py:381,479); 0 (default) = the full split product. */
    int p_single_plane;
} S3dAttnArgs;
int s3d_attention_fwd(const S3dAttnArgs* args, int split, s3d_stream_t stream);
int s3d_attention_bwd(const S3dAttnArgs* args, s3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------ tokenizers
 * VoxelEmbed.forward (embed_layer_3d_modality.py:170-177), VoxelNaiveProjection.forward (:202-209),
 * VoxelEmbed_no_average.forward (:63-70): s3d_voxel_fold gathers the grid into the patch-GEMM operand
 * (mode 0: z-folded sum, the mean is applied as alpha=1/P in the GEMM epilogue; 1: clamp(sum_z); 2: plain patches,
 * rows b*(P^3+1)+1+idx; 3: plain patches, rows grouped '(b px py) pz' with a cls slot per group, :474-476).
 * Token assembly (cls concat + positional add, vit_3d_2d_pretrain.py:458-466) is the S3D_EPI_TOKEN GEMM epilogue. */
typedef struct S3dFoldArgs {
    const float* x;
    uint16_t* a_hi; uint16_t* a_lo; long lda;
    int B, V, c, P, mode;
} S3dFoldArgs;
int s3d_voxel_fold(const S3dFoldArgs* args, s3d_stream_t stream);
/* 2-D branch (Feature3D_ViT2D_V2.forward_images, vit_3d_2d_pretrain.py:435-451): timm PatchEmbed's Conv2d(C, D, p, stride p)
 * as a patch-GEMM operand.  img [B,C,H,W] fp32 -> rows b*(np+1)+1+patch of split-bf16 planes, k in (c, i, j) order; the row
 * b*(np+1) (cls slot) is zeroed.  Followed by s3d_gemm(.., S3D_EPI_TOKEN) with pos = pos_embed, ntok = np+1. */
int s3d_image_patchify(const float* img, uint16_t* a_hi, uint16_t* a_lo, long lda, int B, int C, int H, int W, int p,
                       s3d_stream_t stream);

typedef struct S3dPosGradArgs {
    const float* dx; long groups; int ntok, D;
    float* dpos; float* dcls; float* dbias;
} S3dPosGradArgs;
int s3d_token_grads(const S3dPosGradArgs* args, s3d_stream_t stream);

int s3d_split_bf16(const float* src, uint16_t* hi, uint16_t* lo, long rows, long cols, long ld_out, s3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------ head + loss
 * voxel_head = nn.Linear (vit_3d_2d_pretrain.py:364) or AMSoftmaxLayer.forward (:50-56);
 * F.cross_entropy(pred, cls_idx[, weight]) (train_cls_voxel.py:282-285). */
typedef struct S3dHeadArgs {
    const float* feat; int B, D, C;
    const float* W; const float* bias;
    float* logits;
    int am_softmax; float am_scale;
    const float* dlogits; float* dfeat; float* dW; float* dbias;
    float* scratch;
} S3dHeadArgs;
int s3d_head_fwd(const S3dHeadArgs* args, s3d_stream_t stream);
int s3d_head_bwd(const S3dHeadArgs* args, s3d_stream_t stream);

/* AMSoftmaxLayer as a PER-ROW head (models/3DViT/model.py:123-142, selected by cfg.model.head == 'AMSoftmax' at :230-231 / :427-428: the
 * per-point head of PointTransformerSeg, B*N rows): logits = s * (x / max(|x|, 1e-12)) @ (W / max(|W[:, c]|, 1e-12)), W [D][C].  Factored
 * as a row normalisation around the ordinary Linear GEMMs (s3d_gemm forward / dgrad / wgrad) on the weight Wl[c][d] = s * W[d][c] / |W[:, c]|:
 *   s3d_l2norm_rows_fwd   xn = x / |x| as split-bf16 operand planes (lo may be NULL), inv_norm[r] = 1 / max(|x_r|, 1e-12)
 *   s3d_l2norm_rows_bwd   dx = (dxn - xn (xn . dxn)) * inv_norm            (dx may alias dxn)
 *   s3d_am_weight_fwd     Wl fp32 [C][ldw] (pad columns untouched) and inv_w[c] = 1 / max(|W[:, c]|, 1e-12)
 *   s3d_am_weight_bwd     dW[d][c] += s * (dWl[c][d] - wn[d][c] (wn[:, c] . dWl[c][:])) * inv_w[c]
 * (s3d_head_fwd / s3d_head_bwd with am_softmax = 1 are the small-batch form used by the voxel head.) */
int s3d_l2norm_rows_fwd(const float* x, long ldx, long rows, int D, float* inv_norm, uint16_t* hi, uint16_t* lo, long ldo,
                        s3d_stream_t stream);
int s3d_l2norm_rows_bwd(const float* dxn, long lddxn, const float* x, long ldx, const float* inv_norm, long rows, int D, float* dx,
                        long lddx, s3d_stream_t stream);
int s3d_am_weight_fwd(const float* W, int D, int C, float scale, float* Wl, int ldw, float* inv_w, s3d_stream_t stream);
int s3d_am_weight_bwd(const float* dWl, int ldw, const float* W, const float* inv_w, int D, int C, float scale, float* dW,
                      s3d_stream_t stream);

typedef struct S3dCeArgs {
    const float* logits; const long long* target; const float* weight;
    long rows; int C;
    float* loss;
    float* dlogits;
    float grad_scale;
    int ld;             /* row pitch of logits / dlogits (0 = C); pad columns of dlogits are written as 0 */
} S3dCeArgs;
int s3d_cross_entropy(const S3dCeArgs* args, s3d_stream_t stream);

/* The loss end of the training step in two launches instead of six: final norm of the class rows (vit_3d_2d_pretrain.py:469-470)
 * -> voxel_head Linear (:364) -> F.cross_entropy (train_cls_voxel.py:282-285) -> d(logits) -> d(feat) -> backward of the final
 * norm, one workgroup per sample; then the reductions over the batch (head weight / bias gradients, gamma / beta gradients, the
 * loss) with one writer per output -- deterministic, no atomics.  Same arithmetic as s3d_layernorm_fwd + s3d_head_fwd +
 * s3d_cross_entropy + s3d_head_bwd + s3d_layernorm_bwd. */
typedef struct S3dHeadLossArgs {
    const float* x; long ldx;                 /* class row of sample b = x + b*ldx (the last block's output) */
    int B, D, C; float eps;
    const float* gamma; const float* beta;    /* final LayerNorm */
    const float* W; const float* bias;        /* Linear head, W [C][D] */
    const long long* target; const float* weight;   /* class indices [B]; optional class weights [C] */
    float grad_scale;                          /* multiplies d(logits) (1 for plain training) */
    float* feat; float* mean; float* rstd;    /* out: norm(x)[:, 0] [B][D] and its statistics [B] */
    float* logits; float* dlogits;            /* out: [B][C] */
    float* loss;                               /* out: loss[0] = (weighted) mean CE, loss[1] = its denominator */
    float* dx; uint16_t* dx_bf; long lddx;    /* out: d(loss)/d(x) at the class rows, fp32 and bf16, row pitch lddx */
    float* dW; float* dbias; float* dgamma; float* dbeta;   /* accumulated (+=) */
    float* scratch;                            /* B * (2*D + 1) floats */
    int zero_tokens;                           /* > 0: also clear dx / dx_bf of the zero_tokens token rows that FOLLOW each class row (lddx = tokens per
                                                * sample * D: the last block's d(x_out) is zero except at the class rows, and a dense last block reads all of it) */
} S3dHeadLossArgs;
int s3d_head_loss_fused(const S3dHeadLossArgs* args, s3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------ optimizer
 * torch.optim.Adam(model.parameters(), lr) .step() (train_cls_voxel.py:195,288) over a flat parameter arena; also
 * refreshes the split-bf16 weight planes and (optionally) zeroes the gradients (optimizer.zero_grad(), :277). */
typedef struct S3dAdamState {
    float lr, beta1, beta2, eps;
    float grad_scale;
    float step_size, bc2_sqrt;
    int step;
    int pad;
} S3dAdamState;
int s3d_adam_step(float* p, float* g, float* m, float* v, uint16_t* hi, uint16_t* lo, long n, S3dAdamState* state,
                  int zero_grad, s3d_stream_t stream);
/* The same update in pieces, so that a slice of the arena can be updated as soon as ITS gradients are final -- on a second stream,
 * beside the rest of loss.backward() (the update streams 36 bytes per parameter from HBM, the small-batch backward chain next to it is
 * latency-bound): s3d_adam_begin advances the step count / bias corrections once per optimizer.step(); every s3d_adam_apply after
 * it (any stream ordered behind the begin) updates [p, p + n) exactly as s3d_adam_step does.  g_wire may be null (fp32 gradient) or
 * the bf16 wire buffer of the slice; max_workgroups <= 0 selects the default grid. */
int s3d_adam_begin(S3dAdamState* state, s3d_stream_t stream);
int s3d_adam_apply(float* p, float* g, const uint16_t* g_wire, float* m, float* v, uint16_t* hi, uint16_t* lo, long n,
                   const S3dAdamState* state, int zero_grad, int max_workgroups, s3d_stream_t stream);
/* The update of up to 64 disjoint float4-aligned ranges {offset, count} (in floats, host array ranges[2*n]) of the arena in ONE launch
 * (after s3d_adam_begin): what S3dAdamFill below left over. */
int s3d_adam_apply_ranges(float* p, float* g, float* m, float* v, uint16_t* hi, uint16_t* lo, const long* ranges, int n,
                          const S3dAdamState* state, int zero_grad, s3d_stream_t stream);
/* optimizer.step() as FILLER work inside loss.backward() (train_cls_voxel.py:287-288): the Adam update streams 40 bytes per parameter
 * (0.86 GB, 7 % of a cfg-2 step as a launch of its own) while the small-batch backward is a chain of latency-bound launches that leaves
 * the HBM idle.  With S3dBlockScratch::adam_fill set, s3d_blocks_bwd lets the update of the GEMM parameters (qkv / proj / fc1 / fc2
 * weights and biases) of block i+1 -- final once that block's backward has retired -- ride on the launches of block i as extra workgroups
 * behind their main grids (same stream, same kernels; bitwise the arithmetic of s3d_adam_apply).  Call s3d_adam_begin BEFORE the backward.
 * p / g / m / v / hi / lo: the arena BASE pointers (the block gradients of S3dBlockGrads must lie inside g); filled: host array that
 * receives the {offset, count} pairs (floats) of the ranges the call has updated, *n_filled their number (<= filled_cap / 2 pairs);
 * everything else (LayerNorm parameters, the block processed last, parameters outside the blocks) is the caller's to update afterwards,
 * e.g. with one s3d_adam_apply_ranges.  Not for gradients that still have to be all-reduced or accumulated (data parallel, group_embed's
 * two passes). */
typedef struct S3dAdamFill {
    float* p; float* g; float* m; float* v;
    uint16_t* hi; uint16_t* lo;
    const S3dAdamState* state;
    int zero_grad;
    long* filled; int filled_cap; int* n_filled;       /* HOST memory */
} S3dAdamFill;
/* Data-parallel gradient wire format (replaces DDP's fp32 bucket all-reduce, train_cls_voxel.py:155-159,287, by half the bytes
 * on xGMI): s3d_pack_bf16 rounds a finished gradient bucket to bf16 (rne; n % 8 == 0), the bf16 buffer is sum-all-reduced, and
 * s3d_adam_step_wire takes the gradient from it (g is only zeroed).  The averaging stays in S3dAdamState::grad_scale. */
int s3d_pack_bf16(const float* src, uint16_t* dst, long n, s3d_stream_t stream);
int s3d_adam_step_wire(float* p, float* g, const uint16_t* g_wire, float* m, float* v, uint16_t* hi, uint16_t* lo, long n,
                       S3dAdamState* state, int zero_grad, s3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------ mid-graph events for data parallelism
 * DDP (train_cls_voxel.py:155-159, :287) all-reduces gradient buckets while backward is still running.  With HIP graphs that used to
 * mean one graph per backward segment and a host-launched RCCL call in between -- every graph boundary drains the device queue
 * (DESIGN.md section 7).  Here the whole step is captured as ONE flat graph with a marker (s3d_graph_marker) behind every backward
 * segment; s3d_graph_events_at_markers then replaces marker k of the captured hipGraph_t (e.g. torch.cuda.CUDAGraph(keep_graph=True)
 * .raw_cuda_graph(), before its first replay) by an event-record node of events[k], in line.  The host launches the graph once and,
 * per bucket, makes the stream it issues the collective from wait for the event behind "its" segment (s3d_stream_wait_event, called
 * after the graph launch: it sees the record that replay is going to perform).  The collective overlaps the later segments, no
 * collective is inside a graph (no hang risk with real peers), and the compute stream runs the whole backward without a boundary.
 * PyTorch's Event API refuses external events inside its captures (tools/probes/external_event_probe.py) and so does
 * hipEventRecordWithFlags(hipEventRecordExternal) in the runtime it bundles; the edited graph works (tools/probes/step_graph_probe.py). */
int s3d_event_create(void** event_out);
int s3d_event_destroy(void* event);
int s3d_graph_marker(int id, s3d_stream_t capturing_stream);
int s3d_graph_events_at_markers(void* hip_graph, void* const* events, int n);
int s3d_stream_wait_event(s3d_stream_t waiting_stream, void* event);
/* NOT part of the operator ABI -- measurement aid: copies nbytes (16-byte multiple) with 16 workgroups paced to gbps GB/s (a stand-in
 * with the duration and footprint of a ring all-reduce on one GPU; simple3d-former_amd/parallel.py, profiles/r05_dp_branch_tax.txt). */
int s3d_debug_paced_copy(void* dst, const void* src, long nbytes, float gbps, s3d_stream_t stream);
/* NOT part of the operator ABI -- same-process A/B of dispatch alternatives (tools/r6/attn_ab.py): knob ids are private to the library's
 * launchers (0 .. 15), value -1 restores the shipped rule.  The shipped rules quote the measurements these knobs produced.  In use:
 * 0 = 2: the round-5 long-sequence attention forward instead of the pipelined one; 2 = 0: s3d_encoder_layer_fwd keeps the full split in P V
 * (S3dAttnArgs::p_single_plane off); 8 = 0: the per-wave forward instead of the one-tile kernel at head dim 256, N <= 32.  Tuning builds only (make EXP=1; the product library has no kernel that computes wrong results):
 * 1 = 1 .. 5: timing ablations of the pipelined forward; 3 = 1 .. 8: timing ablations of the long-sequence dK / dV kernel; 4 = 0: the
 * pipelined forward with per-tile staging addresses (A/B of the uniform-base form). */
int s3d_debug_knob(int id, int value);

/* ------------------------------------------------------------------------------------------------ timm Block
 * One pre-norm transformer block: x += attn(norm1(x)); x += mlp(norm2(x))  (timm==0.3.2 Block.forward, invoked by
 * `for blk in self.blocks: x = blk(x)` at vit_3d_2d_pretrain.py:467-468,481-482,493-494) and its backward. */
typedef struct S3dBlockShape {
    int Bb, N, D, H, hidden;      /* sequences, tokens per sequence, model dim, heads, MLP hidden */
    float eps;
    int split;                    /* 1: split-bf16 forward (default), 0: plain bf16 */
    int cls_only_block;           /* index + 1 of the block whose OUTPUT is consumed at the class-token rows only (the last block:
                                   * forward_features returns norm(x)[:, 0], vit_3d_2d_pretrain.py:469-470), 0 = none.  Everything
                                   * after that block's attention is row-local (proj, norm2, mlp, residuals), so s3d_blocks_fwd /
                                   * s3d_blocks_bwd run it on the Bb class rows (row pitch N*D) instead of all Bb*N rows -- same
                                   * values at the rows that matter; x_mid / x_out / hact of the other rows are not produced.
                                   * Needs S3dBlockScratch::dx_b_cls / dx_b_bf_cls / datt_cls. */
    int fuse;                     /* 0: the library picks the launch structure -- for the split-bf16 forward of small token counts (N <= 32
                                   * tokens per sequence, head dim 64, D = 192 / 384) norm1 + qkv + attention run as ONE launch per block and
                                   * norm2 + fc1 + GELU as another (four launches per block instead of seven; acts->qkv_lo is then not
                                   * written), and in the backward attn.proj's dgrad runs inside the attention-backward launch with its wgrad
                                   * on the qkv pair launch (six launches per block instead of seven; scratch->datt / delta are then not
                                   * written); 1: the fused forward only; -1: always the seven-launch sequences */
    int* ln_tickets;              /* optional: >= ceil(Bb*N / 32) zero-initialised ints (left zero by every call).  When set, norm2 and
                                   * (in s3d_blocks_fwd) the NEXT block's norm1 run inside the attn.proj / mlp.fc2 GEMM launches
                                   * (S3dGemmArgs::ln_tickets) instead of as LayerNorm kernels of their own */
} S3dBlockShape;
typedef struct S3dBlockParams {
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *qkv_b, *proj_b, *fc1_b, *fc2_b;
    const uint16_t *qkv_w_hi, *qkv_w_lo, *proj_w_hi, *proj_w_lo, *fc1_w_hi, *fc1_w_lo, *fc2_w_hi, *fc2_w_lo;
} S3dBlockParams;
typedef struct S3dBlockGrads {
    float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *qkv_w, *qkv_b, *proj_w, *proj_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} S3dBlockGrads;
typedef struct S3dBlockActs {     /* written by forward, read by backward; M = Bb*N rows */
    float *x_in, *x_mid, *x_out;                  /* fp32 residual stream [M][D] */
    float *mean1, *rstd1, *mean2, *rstd2;         /* [M] */
    float *lse;                                   /* [Bb*H*N] */
    uint16_t *xn1_hi, *xn1_lo, *qkv_hi, *qkv_lo, *att_hi, *att_lo, *xn2_hi, *xn2_lo;
    uint16_t *hpre, *hact_hi, *hact_lo;           /* [M][hidden] */
    uint16_t* hpre_lo;                            /* optional [M][hidden]: low plane of hpre, written by the forward when set (the
                                                   * split-precision backward needs the pre-activation to 16 bits); disables the fused launches */
} S3dBlockActs;
typedef struct S3dBlockScratch {  /* backward scratch shared by all blocks */
    float* dxn;                                   /* [M][D] */
    float *dx_a, *dx_b;                           /* ping-pong residual gradients [M][D] */
    uint16_t *dx_a_bf, *dx_b_bf;                  /* bf16 copies */
    uint16_t* dh;                                 /* [M][hidden] */
    uint16_t* dqkv;                               /* [M][3D] */
    uint16_t* datt;                               /* [M][D] */
    float* delta;                                 /* [Bb*H*N] */
    /* optional: LayerNorm column-sum partials, [n_layers_in_one_s3d_blocks_bwd_call][2][ln_partial_blocks][2][D]; when set,
     * s3d_blocks_bwd / s3d_block_bwd run the LayerNorm backward kernels in partial mode and finish with one
     * s3d_layernorm_grad_reduce over all their LayerNorms (NULL: atomics) */
    float* ln_partial; int ln_partial_blocks;
    /* S3dBlockShape::cls_only_block: d(x_mid) fp32 / bf16 and d(att) of that block, [M][D] each, ZERO-initialised by the caller once;
     * only the class rows are ever written, so the other rows stay zero for the dense attention / norm1 backward that follow */
    float* dx_b_cls; uint16_t* dx_b_bf_cls; uint16_t* datt_cls;
    /* Split-precision backward (parity mode; train_cls_voxel.py:287 checked to ~1e-4 instead of the bf16 noise floor): low planes of every
     * bf16 gradient buffer above.  When dx_a_lo is set (then all of them must be, and acts->hpre_lo / the lo planes of the saved
     * activations), s3d_block(s)_bwd run every dgrad / wgrad as a three-MFMA split product on hi + lo operands without split-K, the
     * attention backward in fp32, and keep every intermediate gradient as a hi + lo pair.  Several times slower; tests only.
     * The forward that precedes it must have written PER-BLOCK low planes (xn1_lo, qkv_lo, xn2_lo, hact_lo of every S3dBlockActs entry
     * distinct buffers) with S3dBlockShape::fuse = -1: the fused launches do not write qkv_lo, and low planes shared between blocks hold
     * the LAST block's data -- the library can only check that the pointers are set. */
    uint16_t *dx_a_lo, *dx_b_lo, *dh_lo, *dqkv_lo, *datt_lo, *dx_b_lo_cls, *datt_lo_cls;
    const S3dAdamFill* adam_fill;                 /* optional (host pointer): the optimizer update rides on the backward launches, see S3dAdamFill */
    /* Round 5 -- dgrad chain + grouped wgrads (small token counts with the fused attention backward, plain-bf16 backward only).  wg_ring =
     * wg_slots x s3d_block_wgrad_slot_bytes(shape) bytes: every block of a s3d_blocks_bwd call keeps its four dy tensors (d(x_out), d(x_mid),
     * dh, dqkv as bf16) in a slot of its own instead of the shared dx_a_bf / dx_b_bf / dh / dqkv, its backward becomes dgrad-only launches,
     * and the wgrads of up to wg_slots blocks run as ONE s3d_gemm_wgrad_group launch (read-modify-write into the gradient arena: no
     * split-K, no atomics) before the last LayerNorm backward of the group / of the call.  NULL / 0: the paired launches of rounds 1 - 4.
     * dgrad_splitk (1 .. 4; 0 = 1): k-slices of the fc1 / qkv dgrads on that path; dxn must then hold dgrad_splitk planes of [M][D]. */
    uint16_t* wg_ring; int wg_slots; int dgrad_splitk;
    /* ... and the LayerNorm backward kernels of that chain folded into the dgrads (see S3dRowStats): ln_aux = per block [u2 | c2 | u1 | c1] =
     * 2 * (hidden + 3 D) floats, block i at i * that (filled by s3d_blocks_bwd itself, one s3d_ln_aux launch per call, or by s3d_blocks_ln_aux), ln_rowstat = 4 * M floats, ZERO when the
     * first backward runs (the chain's own launches clear what they have consumed).  Both NULL: stand-alone LayerNorm backward launches. */
    float* ln_aux; float* ln_rowstat;
    int ln_aux_valid;                             /* 1: the caller has run s3d_blocks_ln_aux for the blocks of this call since the last parameter
                                                   * update (a backward issued as several s3d_blocks_bwd calls computes the vectors once); 0: every call
                                                   * computes the vectors of its own blocks.  ln_aux is indexed by block: depth * 2 * (hidden + 3 D) floats. */
    int wg_overwrite;                             /* 1: the grouped wgrads STORE dW / db instead of adding to them (the caller knows the gradient
                                                   * arena is not accumulating across backward calls: no read of the old values) */
} S3dBlockScratch;
size_t s3d_block_wgrad_slot_bytes(const S3dBlockShape* shape);
/* Workspace layout for callers that do not want to re-derive it (the shipped Python host allocates the same buffers one by one,
 * simple3d-former_amd/engine.py::_BlockWorkspace / _BlockScratch): ONE device allocation holds the saved activations of `depth`
 * consecutive blocks (acts[i].x_out aliases acts[i+1].x_in; low planes of xn1 / qkv / xn2 / hact shared by all blocks -- they only feed the
 * next forward launch) and, with with_backward != 0, the backward scratch incl. the LayerNorm partial sums and, when
 * shape->cls_only_block != 0, the class-row buffers.  s3d_block_workspace_bytes returns the size (0 on a bad shape); s3d_block_workspace_carve
 * fills acts[0 .. depth) and *scratch (may be NULL when with_backward == 0) with pointers into `base` (256-byte aligned pieces; host-side
 * arithmetic only, nothing is enqueued) and reports the sub-range [*zero_offset, *zero_offset + *zero_bytes) that the caller must clear
 * ONCE before the first backward (the class-row buffers: only their class rows are ever written; with_backward = 2: also the row statistics).  Split-precision parity mode (the *_lo
 * gradient planes, hpre_lo) is not laid out here.  with_backward = 2: additionally the dy ring of the dgrad chain (S3dBlockScratch::wg_ring:
 * min(depth, 3) slots, dgrad_splitk = 3 and a three-plane dxn) and the buffers of the fused LayerNorm backward (ln_aux, ln_rowstat). */
size_t s3d_block_workspace_bytes(const S3dBlockShape* shape, int depth, int with_backward);
int s3d_block_workspace_carve(const S3dBlockShape* shape, int depth, int with_backward, void* base, size_t bytes, S3dBlockActs* acts,
                              S3dBlockScratch* scratch, size_t* zero_offset, size_t* zero_bytes);
int s3d_block_fwd(const S3dBlockShape* shape, const S3dBlockParams* params, const S3dBlockActs* acts,
                  s3d_stream_t stream);
/* in: d(x_out) in scratch->dx_a (+ bf16 copy dx_a_bf); out: d(x_in) in scratch->dx_a / dx_a_bf again. */
int s3d_block_bwd(const S3dBlockShape* shape, const S3dBlockParams* params, const S3dBlockGrads* grads,
                  const S3dBlockActs* acts, const S3dBlockScratch* scratch, s3d_stream_t stream);
/* depth consecutive blocks (acts[i].x_out must alias acts[i+1].x_in); backward runs i = depth-1 .. 0. */
int s3d_blocks_fwd(const S3dBlockShape* shape, const S3dBlockParams* params, const S3dBlockActs* acts, int depth,
                   s3d_stream_t stream);
int s3d_blocks_bwd(const S3dBlockShape* shape, const S3dBlockParams* params, const S3dBlockGrads* grads,
                   const S3dBlockActs* acts, const S3dBlockScratch* scratch, int first, int last,
                   s3d_stream_t stream);
/* blocks first .. last (first <= last < depth) of the same stack, the forward's counterpart of s3d_blocks_bwd's range: a data-parallel
 * caller whose parameters arrive bucket by bucket (sharded optimizer: all-gather of block k's parameters overlapped with blocks < k of the next
 * forward, train_cls_voxel.py:287-288's DDP overlap turned around) runs the forward in the same ranges.  Block `last` never reads the
 * parameters of block last + 1 (the fused next-norm1 hand-off of ln_tickets stops at the range end); results are those of s3d_blocks_fwd. */
int s3d_blocks_fwd_range(const S3dBlockShape* shape, const S3dBlockParams* params, const S3dBlockActs* acts, int depth, int first, int last,
                         s3d_stream_t stream);
/* the weights-only row-statistics vectors (S3dBlockScratch::ln_aux) of blocks last .. first in one launch; set scratch->ln_aux_valid = 1 for the
 * s3d_blocks_bwd calls that follow, until the parameters change */
int s3d_blocks_ln_aux(const S3dBlockShape* shape, const S3dBlockParams* params, const S3dBlockScratch* scratch, int first, int last,
                      s3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------ group_embed
 * nn.TransformerEncoderLayer(d_model=D, dim_feedforward=D, nhead=4) exactly as the reference builds and feeds it
 * (models/vit_3d_2d_pretrain.py:381, :479): default batch_first=False and post-norm, so for the (G, Nb, D) input
 * (G = B*P*P groups, Nb = P+1 tokens per group) self-attention runs over the G axis for each of the Nb positions, i.e.
 * ACROSS the samples of the batch; ReLU feed-forward D -> Dff -> D; LayerNorm eps 1e-5.  Rows are r = g*Nb + t.
 * Dropout (p = 0.1, four sites: attention weights, after out_proj, after the ReLU, after linear2) uses a counter-based
 * hash mask keep = mix32(index, seed, site) >= p*2^32 -- for the attention weights (site 0) one hash per pair of adjacent keys, 16 bits
 * per decision: there p is resolved to 2^-16 (the drop rate is floor(p 2^16) / 2^16: 0.09999 for p = 0.1, and 0 for p < 2^-16) while kept
 * weights are still scaled by 1 / (1 - p), i.e. E[mask scale] deviates from 1 by at most 1.5e-5 relative (the oracle uses the same rule); csrc/common.h drop_keep / drop_keep_attn == oracle.voxel_oracle.hash_keep_mask (torch's RNG stream cannot be reproduced); dropout_p = 0 gives
 * the eval-mode layer. */
typedef struct S3dEncShape {
    int G, Nb, D, H, Dff;
    float eps;
    int split;
    float dropout_p;                       /* 0 = eval mode; 0.1 = the reference's training mode */
    const unsigned long long* seed;        /* device-resident seed (bump it once per step), required when dropout_p > 0 */
} S3dEncShape;
typedef struct S3dEncParams {
    const float *in_b, *out_b, *l1_b, *l2_b, *n1_w, *n1_b, *n2_w, *n2_b;
    const uint16_t *in_w_hi, *in_w_lo, *out_w_hi, *out_w_lo, *l1_w_hi, *l1_w_lo, *l2_w_hi, *l2_w_lo;
} S3dEncParams;
typedef struct S3dEncGrads {
    float *in_w, *in_b, *out_w, *out_b, *l1_w, *l1_b, *l2_w, *l2_b, *n1_w, *n1_b, *n2_w, *n2_b;
} S3dEncGrads;
typedef struct S3dEncActs {      /* M = G*Nb rows */
    float *x_in, *s1, *x1, *s2, *x_out;            /* fp32 [M][D] */
    float *mean1, *rstd1, *mean2, *rstd2, *lse;    /* [M] x4, [Nb*H*G] */
    uint16_t *xin_hi, *xin_lo, *qkv_hi, *qkv_lo, *att_hi, *att_lo, *x1_hi, *x1_lo;
    uint16_t *fpre, *f_hi, *f_lo;                  /* [M][Dff] */
    uint16_t* fpre_lo;                             /* optional: low plane of fpre (split-precision backward, see S3dBlockScratch::dx_a_lo) */
    unsigned int* attn_mask;                       /* optional: S3dAttnArgs::drop_mask of the layer's attention, Nb*H * ceil(G/32)^2 * 32 words */
} S3dEncActs;
int s3d_encoder_layer_fwd(const S3dEncShape* shape, const S3dEncParams* params, const S3dEncActs* acts,
                          s3d_stream_t stream);
/* in: d(x_out) in scratch->dx_a; out: d(x_in) in scratch->dx_b (fp32) and scratch->dx_b_bf (bf16 copy). */
int s3d_encoder_layer_bwd(const S3dEncShape* shape, const S3dEncParams* params, const S3dEncGrads* grads,
                          const S3dEncActs* acts, const S3dBlockScratch* scratch, s3d_stream_t stream);

/* pass-2 token assembly of group_embed (vit_3d_2d_pretrain.py:485-491): out[b*(n+1)+t] = (t==0 ? cls : src[b*n+t-1]) + pos[t]
 * and the row gather of its backward: dsrc[b*n+j] = dout[b*(n+1)+1+j]. */
int s3d_assemble_tokens(const float* src, const float* cls, const float* pos, float* out, long B, int n, int D,
                        s3d_stream_t stream);
int s3d_assemble_tokens_bwd(const float* dout, float* dsrc, long B, int n, int D, s3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------ point clouds
 * PointTransformerCls / PointTransformerSeg (models/3DViT/model.py:144-337, 341-535) geometry and normalisation operators.
 *   s3d_fps                farthest_point_sample (data/pointnet_util.py:53-73); `start` = the torch.randint draw of :65
 *   s3d_knn                square_distance + argsort()[:, :, :k] (:119-120) for k = 16, and the 3-NN + inverse-distance
 *                          weights of PointNetFeaturePropagation (:401-408) for k = 3 (out_w non-NULL)
 *   s3d_group_gather       index_points + [xyz - new_xyz | feats] concat (:126-134) -> split-bf16 GEMM operand rows (b,s,j)
 *   s3d_group_scatter      its backward (scatter-add into d(feats))
 *   s3d_group_project_*    the first 1x1 convolution of the level WITHOUT the grouped operand: by linearity
 *                          conv0([xyz_rel | feats[idx]]) = Pf[idx] + xyz_rel . Wx^T + b with Pf = feats . Wf^T computed once per
 *                          POINT (k = 16 times fewer GEMM rows than sample_and_group's [B,S,k,3+C] tensor, :126-134 + :238-239)
 *   s3d_batchnorm_fwd/bwd  train-mode nn.BatchNorm2d/1d + ReLU (:238-241, models/3DViT/model.py:52-64) over row matrices
 *                          [rows][C]; with K > 0 also the max over the K neighbours (:242) fused in
 *   s3d_interp3(_bwd)      3-NN interpolation + skip add of TransitionUp (models/3DViT/model.py:67-72)
 *   s3d_mean_points        x.mean(1) (models/3DViT/model.py:325);  s3d_bcast_rows its backward
 *   s3d_sgd_step           torch.optim.SGD(lr, momentum=0.9) (train_cls.py:91) over the flat arena */
int s3d_fps(const float* xyz, long xyz_ld, const long long* start, int B, int N, int npoint, int* out_idx, float* new_xyz,
            s3d_stream_t stream);
int s3d_knn(const float* query, const float* ref, int B, int S, int N, int K, int* out_idx, float* out_w, s3d_stream_t stream);
int s3d_group_gather(const float* xyz, const float* new_xyz, const float* feats, const int* idx, int B, int N, int S, int K,
                     int C, uint16_t* a_hi, uint16_t* a_lo, int lda, s3d_stream_t stream);
int s3d_group_scatter(const float* dA, int ldd, const int* idx, int B, int N, int S, int K, int C, float* dfeats,
                      s3d_stream_t stream);
typedef struct S3dGroupProjArgs {
    const float* xyz; const float* new_xyz; const int* idx;   /* [B,N,3], [B,S,3], neighbours [B,S,K] */
    int B, N, S, K, C;                    /* C = output channels of the convolution (multiple of 4, <= 1024) */
    const float* W; int ldw;              /* conv weight [C][ldw]; columns 0..2 multiply xyz_rel (the rest is Wf) */
    const float* bias;                    /* [C] */
    const float* Pf; long ldp;            /* fwd in: per-point projection feats . Wf^T, [B*N][ldp] */
    float* x; long ldx;                   /* fwd out: pre-BatchNorm rows (b, s, j) -> [B*S*K][ldx] */
    double* sums;                         /* fwd, optional: [2*C] column sums and sums of squares of x, ACCUMULATED (zero it first):
                                           * the statistics pass of the BatchNorm that follows (S3dBnArgs::have_sums) */
    const uint16_t* dx; long lddx;        /* bwd in: gradient wrt x as bf16 [B*S*K][lddx] */
    const int* inv_off; const int* inv_rows;   /* bwd in: transpose of idx (s3d_neighbor_csr): rows that reference each point */
    float* dPf;                           /* bwd out: [B*N][ldp] = sum of dx over the rows that reference the point (written, no atomics) */
    float* dW; float* dbias;              /* bwd out, accumulated: dW[c][0..2] (row pitch ldw) and dbias[c] */
} S3dGroupProjArgs;
int s3d_group_project_fwd(const S3dGroupProjArgs* args, s3d_stream_t stream);
int s3d_group_project_bwd(const S3dGroupProjArgs* args, s3d_stream_t stream);
/* Transpose of the neighbour lists idx [B][S*K] (values in [0, N)): inv_off [B][N+1] (exclusive prefix of the in-degrees) and
 * inv_rows [B][S*K] = for every point the ascending list of entries e = s*K + j with idx[b][e] == point.  Lets the backward
 * GATHER the rows that touch a point (deterministic, no atomics) instead of scatter-adding R x C gradients. */
int s3d_neighbor_csr(const int* idx, int B, int N, int S, int K, int* inv_off, int* inv_rows, s3d_stream_t stream);
typedef struct S3dBnArgs {
    const float* x; int ldx;              /* pre-normalisation activations [rows][ldx] */
    long rows; int C; int K;              /* K > 0: rows = groups*K and the max over K is fused (y = [groups][C], arg) */
    float eps; float momentum;
    const float* gamma; const float* beta;
    float* mean; float* rstd;             /* [C] batch statistics (written by fwd, read by bwd) */
    float* run_mean; float* run_var;      /* running statistics, updated in place by fwd (may be NULL) */
    double* sums;                         /* scratch [2*C] */
    float* y; uint16_t* y_hi; uint16_t* y_lo; int ldo;     /* fwd outputs: fp32 and/or split planes */
    unsigned char* arg;                   /* K > 0: argmax neighbour per (group, channel) */
    const float* dy; int lddy;            /* bwd: gradient wrt the (ReLU / max) output */
    const uint16_t* dy_bf;                /* bwd: the same gradient as bf16 (row pitch lddy) instead of dy -- lets the dgrad GEMM that
                                           * produces it write 2 bytes per element */
    uint16_t* dx; int lddx;               /* bwd: gradient wrt x as bf16 [rows][lddx] */
    float* dgamma; float* dbeta;
    int eval_mode;                        /* fwd: normalise with the running statistics (model.eval()), no update */
    int have_sums;                        /* fwd: `sums` already holds the column sums / sums of squares of x (the producer of x
                                           * accumulated them, e.g. s3d_group_project_fwd): skip the statistics pass */
    const float* momentum_dev;            /* optional: device-resident momentum (overrides `momentum`), so that a captured HIP graph
                                           * follows the reference's per-epoch BN-momentum decay (train_partseg.py:126-130) */
    int sums_zeroed;                      /* the caller has zeroed `sums` (e.g. one memset over the statistics of every layer of a model):
                                           * the library does not enqueue its own 2C-double memset in front of the statistics pass */
} S3dBnArgs;
int s3d_batchnorm_fwd(const S3dBnArgs* args, s3d_stream_t stream);
int s3d_batchnorm_bwd(const S3dBnArgs* args, s3d_stream_t stream);
int s3d_interp3(const float* f1, int S, const float* f2, const int* idx, const float* w, int B, int N, int C, float* out,
                s3d_stream_t stream);
int s3d_interp3_bwd(const float* dout, const int* idx, const float* w, int B, int S, int N, int C, float* df1,
                    s3d_stream_t stream);
int s3d_mean_points(const float* x, int B, int N, int C, float* out, s3d_stream_t stream);
int s3d_bcast_rows(const float* x, int N, int C, long rows, float scale, float* y, s3d_stream_t stream);
int s3d_pack_rows(const float* x, int C, int ldx, long rows, uint16_t* hi, uint16_t* lo, int ldo, s3d_stream_t stream);
int s3d_add_inplace(float* a, const float* b, long n, s3d_stream_t stream);
int s3d_sgd_step(float* p, float* g, float* buf, uint16_t* hi, uint16_t* lo, long n, float lr, float momentum,
                 float grad_scale, int* step_counter, s3d_stream_t stream);
/* the same step with {lr, momentum, grad_scale} read from device memory at run time: a captured HIP graph follows the
 * reference's per-epoch learning-rate decay (train_partseg.py:121-125) without re-capture */
int s3d_sgd_step_dev(float* p, float* g, float* buf, uint16_t* hi, uint16_t* lo, long n, const float* hyper,
                     int* step_counter, s3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------ evaluation + input format
 *   s3d_cls_eval       pred = logits.max(1)[1]; counts[0] += correct, counts[1+c] += correct of class c, counts[1+C+c] += seen of
 *                      class c (the accuracy / mean-class-accuracy bookkeeping of train_cls_voxel.py:315-329, train_cls.py:22-41)
 *   s3d_partseg_eval   train_partseg.py:181-206: argmax restricted to the parts [first, first+count) of the shape's own category
 *                      (category of target[b][0]; part_range[2*l] = first, part_range[2*l+1] = count for every part label l),
 *                      per-shape mean part IoU (1.0 for a part that is neither present nor predicted), per-part seen / correct
 *                      counts in counts[1+l] / counts[1+P+l], counts[0] = total correct points
 *   s3d_unpack_voxels  1 bit per voxel (z fastest, LSB first, 32 voxels per word) -> fp32 grid: the device half of the
 *                      .binvox reader (utils/binvox_rw.py:117-151 does the RLE decode on the host) */
int s3d_cls_eval(const float* logits, int ld, const long long* target, long rows, int C, int* pred, long long* counts,
                 s3d_stream_t stream);
int s3d_partseg_eval(const float* logits, int ld, const long long* target, int B, int N, int num_part, const int* part_range,
                     int* pred, double* shape_iou, int* shape_first, long long* counts, s3d_stream_t stream);
int s3d_unpack_voxels(const unsigned int* bits, float* out, long nwords, s3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* S3D_HIP_H */
