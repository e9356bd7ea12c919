// extern "C" surface of libs3d_hip.so (include/s3d_hip.h) + the per-block launch sequences.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <map>
#include <string>
#include <vector>

#include "adam_fill.h"
#include "attention.h"
#include "bwd_gemm.h"
#include "fused_block.h"
#include "fused_mlp.h"
#include "gemm.h"
#include "kernels.h"
#include "s3d_hip.h"

static thread_local char g_err[512] = "";
static int g_deterministic = -1;                  // -1: not set yet -> S3D_DETERMINISTIC decides
bool s3d_deterministic() {
    if (g_deterministic < 0) {
        const char* v = getenv("S3D_DETERMINISTIC");
        g_deterministic = (v && atoi(v) > 0) ? 1 : 0;
    }
    return g_deterministic != 0;
}

static int g_knobs[16] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
int s3d_knob(int id) { return (id >= 0 && id < 16) ? g_knobs[id] : -1; }

// ---- launch coverage (common.h: S3D_CHECK_LAUNCH_V)
bool g_s3d_cov_on = false;
static std::map<std::pair<std::string, long long>, long> g_cov;
void s3d_cov_note(const char* name, long long variant) { ++g_cov[{std::string(name), variant}]; }

void s3d_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define S3D_TRY(expr)            \
    do {                         \
        int rc__ = (expr);       \
        if (rc__ != 0) return rc__; \
    } while (0)

static inline hipStream_t st(s3d_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---------------------------------------------------------------------------------------------- block sequences
namespace {

GemmArgs gemm_zero() {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.alpha = 1.f;
    return g;
}

// next_p / next_a: the block that follows (s3d_blocks_fwd), whose norm1 can ride on this block's fc2 launch; ln1_done: this block's
// norm1 was already produced that way.  Returns 0 / error; *next_ln1_done tells the caller whether the next block may skip norm1.
int block_fwd(const S3dBlockShape& sh, const S3dBlockParams& p, const S3dBlockActs& a, hipStream_t s, bool ln1_done = false,
              const S3dBlockParams* next_p = nullptr, const S3dBlockActs* next_a = nullptr, bool* next_ln1_done = nullptr,
              bool cls_only = false) {
    const long M = (long)sh.Bb * sh.N;
    const int D = sh.D, Hd = sh.hidden;
    const bool split = sh.split != 0;
    // cls_only: the row-local tail of the block (proj .. fc2) runs on the Bb class rows, addressed in place with row pitch N*D
    const long M2 = cls_only ? sh.Bb : M;
    const long pd = cls_only ? (long)sh.N * D : D, ph = cls_only ? (long)sh.N * Hd : Hd;
    S3D_REQUIRE(M < (1L << 31), "block: too many rows");
    if (next_ln1_done) *next_ln1_done = false;
    // Fused launches (fused_block.hip): norm1 + qkv + attention per (sample pair, head), norm2 + fc1 + GELU per (64-row band, hidden
    // slice).  Split-bf16 forward of the small-batch shapes only; S3dBlockShape::fuse = -1 keeps the seven-launch sequence.
    const bool fuse_ok = split && sh.fuse >= 0 && sh.ln_tickets == nullptr && !ln1_done && a.hpre_lo == nullptr;
    const bool fuse_attn = fuse_ok && s3d_fused_attn_ok(sh.Bb, sh.N, D, sh.H);
    const bool fuse_mlp = fuse_ok && !cls_only && s3d_fused_mlp1_ok(M, D, Hd);
    LnArgs ln;
    memset(&ln, 0, sizeof(ln));
    ln.D = D; ln.eps = sh.eps;
    GemmArgs g = gemm_zero();
    if (fuse_attn) {
        FusedAttnArgs fa;
        fa.x = a.x_in; fa.gamma = p.ln1_w; fa.beta = p.ln1_b; fa.eps = sh.eps;
        fa.w_hi = p.qkv_w_hi; fa.w_lo = p.qkv_w_lo; fa.bias = p.qkv_b;
        fa.xn_hi = a.xn1_hi; fa.xn_lo = a.xn1_lo; fa.mean = a.mean1; fa.rstd = a.rstd1;
        fa.qkv_hi = a.qkv_hi; fa.att_hi = a.att_hi; fa.att_lo = a.att_lo; fa.lse = a.lse;
        fa.Bb = sh.Bb; fa.N = sh.N; fa.H = sh.H; fa.scale = 1.0f / sqrtf((float)(D / sh.H));
        AttnArgs bw;                                                       // the problem as block_bwd will pose it: which lse layout?
        memset(&bw, 0, sizeof(bw));
        bw.Bb = sh.Bb; bw.H = sh.H; bw.N = sh.N; bw.D = D; bw.sb = sh.N; bw.st = 1;
        fa.lse_packed = s3d_attention_pairs_packed(bw) ? 1 : 0;
        S3D_TRY(s3d_launch_fused_attn(fa, D, s));
    } else {
    // 1. norm1
    ln.x = a.x_in; ln.ldx = D; ln.rows = M; ln.gamma = p.ln1_w; ln.beta = p.ln1_b;
    ln.out_hi = a.xn1_hi; ln.out_lo = split ? a.xn1_lo : nullptr; ln.ldo = D; ln.mean = a.mean1; ln.rstd = a.rstd1;
    if (!ln1_done) S3D_TRY(s3d_launch_ln_fwd(ln, s));
    // 2. qkv = xn1 @ Wqkv^T + b
    g.A_hi = a.xn1_hi; g.A_lo = a.xn1_lo; g.lda = D; g.B_hi = p.qkv_w_hi; g.B_lo = p.qkv_w_lo; g.ldb = D;
    g.M = (int)M; g.N = 3 * D; g.K = D; g.bias = p.qkv_b; g.O_hi = a.qkv_hi; g.O_lo = split ? a.qkv_lo : nullptr; g.ldo = 3 * D;
    S3D_TRY(s3d_launch_gemm(false, false, split, EPI_BF16_BIAS, g, 1, s));
    // 3. attention
    AttnArgs at;
    memset(&at, 0, sizeof(at));
    at.qkv_hi = a.qkv_hi; at.qkv_lo = a.qkv_lo; at.ld = 3 * D; at.out_hi = a.att_hi; at.out_lo = split ? a.att_lo : nullptr;
    at.ldo = D; at.lse = a.lse; at.Bb = sh.Bb; at.H = sh.H; at.N = sh.N; at.D = D; at.sb = sh.N; at.st = 1;
    at.scale = 1.0f / sqrtf((float)(D / sh.H));
    S3D_TRY(s3d_launch_attention_fwd(at, split, s));
    }
    // 4. x_mid = x_in + att @ Wproj^T + b   [+ norm2 by the last-arriving tile of every row band]
    g = gemm_zero();
    g.A_hi = a.att_hi; g.A_lo = a.att_lo; g.lda = pd; g.B_hi = p.proj_w_hi; g.B_lo = p.proj_w_lo; g.ldb = D;
    g.M = (int)M2; g.N = D; g.K = D; g.bias = p.proj_b; g.R = a.x_in; g.ldr = pd; g.C = a.x_mid; g.ldc = pd;
    g.ln_tickets = cls_only ? nullptr : sh.ln_tickets; g.ln_gamma = p.ln2_w; g.ln_beta = p.ln2_b; g.ln_eps = sh.eps; g.ln_hi = a.xn2_hi;
    g.ln_lo = split ? a.xn2_lo : nullptr; g.ld_ln = D; g.ln_mean = a.mean2; g.ln_rstd = a.rstd2;
    const bool fused2 = s3d_gemm_ln_fusable(split, g);
    S3D_TRY(s3d_launch_gemm(false, false, split, EPI_RESID, g, 1, s));
    // the whole MLP branch as one launch (fused_mlp.hip): deit_tiny at thousands of rows -- the point path's transformer
    static const int mlp_full_off = s3d_tune_int("S3D_FUSED_MLP_FULL");            // 0: the three-launch sequence (A/B in the tuning build)
    const bool mlp_full = split && sh.fuse >= 0 && !cls_only && sh.ln_tickets == nullptr && a.hpre_lo == nullptr && mlp_full_off != 0 &&
                          s3d_fused_mlp_full_ok(M, D, Hd);
    if (mlp_full) {
        FusedMlpArgs fm;
        memset(&fm, 0, sizeof(fm));
        fm.x = a.x_mid; fm.gamma = p.ln2_w; fm.beta = p.ln2_b; fm.eps = sh.eps;
        fm.w_hi = p.fc1_w_hi; fm.w_lo = p.fc1_w_lo; fm.bias = p.fc1_b;
        fm.xn_hi = a.xn2_hi; fm.xn_lo = nullptr; fm.mean = a.mean2; fm.rstd = a.rstd2;
        fm.hpre = a.hpre; fm.hact_hi = a.hact_hi; fm.M = M; fm.hidden = Hd;
        return s3d_launch_fused_mlp_full(fm, p.fc2_w_hi, p.fc2_w_lo, p.fc2_b, a.x_out, D, s);
    }
    if (fuse_mlp) {
        FusedMlpArgs fm;
        fm.x = a.x_mid; fm.gamma = p.ln2_w; fm.beta = p.ln2_b; fm.eps = sh.eps;
        fm.w_hi = p.fc1_w_hi; fm.w_lo = p.fc1_w_lo; fm.bias = p.fc1_b;
        fm.xn_hi = a.xn2_hi; fm.xn_lo = a.xn2_lo; fm.mean = a.mean2; fm.rstd = a.rstd2;
        fm.hpre = a.hpre; fm.hact_hi = a.hact_hi; fm.hact_lo = a.hact_lo;
        fm.M = M; fm.hidden = Hd; fm.nslice = Hd / 192;
        S3D_TRY(s3d_launch_fused_mlp1(fm, D, s));
    } else {
    // 5. norm2 (stand-alone only when the GEMM could not carry it)
    ln.x = a.x_mid; ln.gamma = p.ln2_w; ln.beta = p.ln2_b; ln.out_hi = a.xn2_hi; ln.out_lo = split ? a.xn2_lo : nullptr;
    ln.mean = a.mean2; ln.rstd = a.rstd2; ln.rows = M2; ln.ldx = pd; ln.ldo = pd;       // cls_only: statistics of class row b at [b]
    if (!fused2) S3D_TRY(s3d_launch_ln_fwd(ln, s));
    // 6. h = gelu(xn2 @ W1^T + b1)
    g = gemm_zero();
    g.A_hi = a.xn2_hi; g.A_lo = a.xn2_lo; g.lda = pd; g.B_hi = p.fc1_w_hi; g.B_lo = p.fc1_w_lo; g.ldb = D;
    g.M = (int)M2; g.N = Hd; g.K = D; g.bias = p.fc1_b; g.aux = a.hpre; g.aux_lo = a.hpre_lo; g.ldaux = ph; g.O_hi = a.hact_hi;
    g.O_lo = split ? a.hact_lo : nullptr; g.ldo = ph;
    S3D_TRY(s3d_launch_gemm(false, false, split, EPI_GELU, g, 1, s));
    }
    // 7. x_out = x_mid + h @ W2^T + b2   [+ the next block's norm1]
    g = gemm_zero();
    g.A_hi = a.hact_hi; g.A_lo = a.hact_lo; g.lda = ph; g.B_hi = p.fc2_w_hi; g.B_lo = p.fc2_w_lo; g.ldb = Hd;
    g.M = (int)M2; g.N = D; g.K = Hd; g.bias = p.fc2_b; g.R = a.x_mid; g.ldr = pd; g.C = a.x_out; g.ldc = pd;
    if (next_p && next_a && !cls_only) {
        g.ln_tickets = sh.ln_tickets; g.ln_gamma = next_p->ln1_w; g.ln_beta = next_p->ln1_b; g.ln_eps = sh.eps;
        g.ln_hi = next_a->xn1_hi; g.ln_lo = split ? next_a->xn1_lo : nullptr; g.ld_ln = D; g.ln_mean = next_a->mean1; g.ln_rstd = next_a->rstd1;
        if (next_ln1_done) *next_ln1_done = s3d_gemm_ln_fusable(split, g);
        if (!next_ln1_done || !*next_ln1_done) g.ln_tickets = nullptr;
    }
    S3D_TRY(s3d_launch_gemm(false, false, split, EPI_RESID, g, 1, s));
    return 0;
}

// wgrad: dW[out][in] += dy^T x  (dy [M][out] bf16, x [M][in] bf16), db[out] += colsum(dy)
GemmArgs wgrad_args(const bf16_t* dy, int out, const bf16_t* x, int in, long M, float* dW, float* db, long ld_dy = 0, long ld_x = 0) {
    GemmArgs g = gemm_zero();
    g.A_hi = dy; g.lda = ld_dy ? ld_dy : out; g.B_hi = x; g.ldb = ld_x ? ld_x : in; g.M = out; g.N = in; g.K = (int)M; g.C = dW; g.ldc = in;
    g.bias_grad = db;
    return g;
}
int wgrad(const bf16_t* dy, int out, const bf16_t* x, int in, long M, float* dW, float* db, hipStream_t s) {
    return s3d_launch_gemm(true, true, false, EPI_ATOMIC, wgrad_args(dy, out, x, in, M, dW, db), 0, s);
}

// ---- backward on two streams (S3D_BWD_STREAMS=1) ------------------------------------------------------------------------------
// The block backward is a chain (fc2 dgrad -> fc1 dgrad -> norm2 -> proj dgrad -> attention -> qkv dgrad -> norm1) with four wgrads
// hanging off it.  Pairing every wgrad with "its" dgrad in one launch already overlaps those two; here the wgrads go to a side
// stream instead and run beside whatever the chain does next -- the LayerNorm / attention kernels of the chain are latency-bound
// launches that leave most CUs idle.  Works eagerly and under HIP-graph capture (fork / join through events -> graph branches).
struct BwdStreams {
    hipStream_t side = nullptr;
    hipEvent_t ready = nullptr, done = nullptr;
    bool ok = false;
    bool init(hipStream_t main) {
        if (ok) return true;
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(main, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return false;   // create outside captures only
        if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) return false;
        ok = true;
        return true;
    }
};
BwdStreams g_bwd_streams;
bool bwd_streams_enabled() {
    static const int on = s3d_tune_int("S3D_BWD_STREAMS");                  // measured and rejected: tuning builds only
    return on > 0;
}
// dgrad on `s`, wgrad on the side stream once everything `s` has enqueued so far (i.e. dy) is complete
int dgrad_and_wgrad(int epi, const GemmArgs& dg, const GemmArgs& wg, hipStream_t s, BwdStreams* bs, AdamFillQueue* fq = nullptr,
                    const GemmArgs* wg2 = nullptr) {
    if (bs == nullptr) return s3d_launch_gemm_pair(epi, dg, wg, s, fq, wg2);
    if (hipEventRecord(bs->ready, s) != hipSuccess || hipStreamWaitEvent(bs->side, bs->ready, 0) != hipSuccess) {
        s3d_set_error("backward streams: fork failed");
        return 3;
    }
    S3D_TRY(s3d_launch_gemm(true, true, false, EPI_ATOMIC, wg, 0, bs->side));
    if (wg2) S3D_TRY(s3d_launch_gemm(true, true, false, EPI_ATOMIC, *wg2, 0, bs->side));       // (behind the fork: its operands come from `s` too)
    return s3d_launch_gemm(false, true, false, epi, dg, 1, s);
}
// `s` waits for every wgrad issued so far (before a buffer they read is overwritten / before the block returns)
int bwd_streams_join(hipStream_t s, BwdStreams* bs) {
    if (bs == nullptr) return 0;
    if (hipEventRecord(bs->done, bs->side) != hipSuccess || hipStreamWaitEvent(s, bs->done, 0) != hipSuccess) {
        s3d_set_error("backward streams: join failed");
        return 3;
    }
    return 0;
}

// LayerNorm column-sum partials of one s3d_blocks_bwd call (see S3dBlockScratch::ln_partial): slot k of the call's LayerNorms
struct LnPartials {
    const float* part[64]; float* dg[64]; float* db[64];
    int rows[64];                                  // partial rows this LayerNorm's launch really writes (<= ln_partial_blocks)
    int n = 0;
    float* slot(const S3dBlockScratch& w, int D, float* dgamma, float* dbeta, int nrows = 0) {
        float* ptr = w.ln_partial + (long)n * w.ln_partial_blocks * 2 * D;
        part[n] = ptr; dg[n] = dgamma; db[n] = dbeta; rows[n] = nrows > 0 ? nrows : w.ln_partial_blocks;
        ++n;
        return ptr;
    }
    int flush(const S3dBlockScratch& w, int D, hipStream_t s) {
        const int rc = n ? s3d_launch_ln_grad_reduce_rows(part, dg, db, rows, n, D, s) : 0;
        n = 0;
        return rc;
    }
};

// ---- round 5: dgrad chain + grouped wgrads (S3dBlockScratch::wg_ring) -----------------------------------------------------------
// Every block of a s3d_blocks_bwd call keeps its four dy tensors in a ring slot; its backward is dgrad-only launches (the fc1 / qkv
// dgrads as k-slices whose planes the LayerNorm backward adds), and the wgrads of up to `slots` blocks run as ONE grouped launch
// right before the last LayerNorm backward of the group (bwd_gemm.hip).  That LayerNorm backward is the only launch that writes a
// buffer a pending wgrad may still read (the next block's d(x_out) slot, or the caller's dx_a_bf at the end of the call).
struct WgSlot { bf16_t *dxa, *dxb, *dh, *dqkv; };
inline size_t wg_pad(size_t n) { return (n + 127) / 128 * 128; }
inline size_t wg_slot_elems(size_t M, size_t D, size_t Hd) { return 2 * wg_pad(M * D) + wg_pad(M * Hd) + wg_pad(M * 3 * D); }
struct WgradDefer {
    bf16_t* ring = nullptr;
    int slots = 0, splitk = 1, accumulate = 1;
    size_t M = 0, D = 0, Hd = 0;
    S3dWgradItem items[48];
    int n = 0, pending_blocks = 0, next = 0;
    const bf16_t* dxa_cur = nullptr;               // d(x_out) (bf16) of the block about to run: the caller's dx_a_bf, then ring slots
    const float* aux = nullptr;                    // fused LayerNorm backward: the [u2 | c2 | u1 | c1] vectors, indexed by block
    int block = 0;                                 // the block about to run
    // optimizer update riding on the grouped launches (S3dAdamFill): arena ranges whose gradients are final -- the GEMM parameters of the blocks
    // whose wgrads an EARLIER launch has finished -- go to the next launch as filler shares; ranges of the blocks it computes itself follow after it
    const S3dAdamFill* af = nullptr;
    long ready[6][2]; int nready = 0;              // {offset, count} in floats
    long pend[48][2]; int npend = 0;               // ranges that become ready with the next flush
    int* n_filled = nullptr;                       // running count of the ranges reported in af->filled
    void defer_range(long off, long n) { if (npend < 48 && n >= 4) { pend[npend][0] = off; pend[npend][1] = n / 4 * 4; ++npend; } }
    void ready_range(long off, long n) { if (nready < 6 && n >= 4) { ready[nready][0] = off; ready[nready][1] = n / 4 * 4; ++nready; } }
    WgSlot slot(int i) const {
        bf16_t* b = ring + (size_t)i * wg_slot_elems(M, D, Hd);
        return WgSlot{b, b + wg_pad(M * D), b + 2 * wg_pad(M * D), b + 2 * wg_pad(M * D) + wg_pad(M * Hd)};
    }
    void push(const bf16_t* dy, int out, const bf16_t* x, int in, float* dW, float* db) {
        S3dWgradItem& q = items[n++];
        q.dy = dy; q.ld_dy = out; q.out = out; q.x = x; q.ld_x = in; q.in = in; q.dW = dW; q.ldw = in; q.db = db;
    }
    int flush(hipStream_t s) {
        AdamFill fills[6];
        int nf = 0;
        if (af != nullptr && n > 0) {
            for (int i = 0; i < nready && *n_filled * 2 + 2 <= af->filled_cap; ++i) {
                const long o = ready[i][0], cnt = ready[i][1];
                AdamFill& f = fills[nf++];
                f = adam_fill_none();
                f.p = af->p + o; f.g = af->g + o; f.m = af->m + o; f.v = af->v + o;
                f.hi = af->hi ? af->hi + o : nullptr; f.lo = af->lo ? af->lo + o : nullptr;
                f.st = af->state; f.n4 = cnt / 4; f.zero_grad = af->zero_grad;
                af->filled[2 * *n_filled] = o; af->filled[2 * *n_filled + 1] = cnt;
                ++*n_filled;
            }
            nready = 0;
        }
        const int rc = n ? s3d_launch_wgrad_group(items, n, (int)M, 1.0f, accumulate, s, nf ? fills : nullptr, nf) : 0;
        if (af != nullptr) {                           // what this launch has computed can ride on the next one (contiguous neighbours merged)
            for (int i = 0; i < npend; ++i) {
                if (nready > 0 && ready[nready - 1][0] + ready[nready - 1][1] == pend[i][0]) ready[nready - 1][1] += pend[i][1];
                else if (nready > 0 && pend[i][0] + pend[i][1] == ready[nready - 1][0]) { ready[nready - 1][0] = pend[i][0]; ready[nready - 1][1] += pend[i][1]; }
                else ready_range(pend[i][0], pend[i][1]);
            }
            npend = 0;
        }
        n = 0; pending_blocks = 0;
        return rc;
    }
};

// Split-precision backward of one block (parity mode, S3dBlockScratch::dx_a_lo): the same chain as block_bwd below -- fc2 dgrad * gelu'
// -> fc1 dgrad -> norm2 -> proj dgrad -> attention -> qkv dgrad -> norm1, with the four wgrads -- but every GEMM is a three-MFMA split
// product on hi + lo operands without split-K, the attention backward runs in fp32, and every intermediate gradient is a hi + lo pair.
int block_bwd_split(const S3dBlockShape& sh, const S3dBlockParams& p, const S3dBlockGrads& gr, const S3dBlockActs& a,
                    const S3dBlockScratch& w, hipStream_t s, LnPartials* lp, bool cls_only) {
    const long M = (long)sh.Bb * sh.N;
    const int D = sh.D, Hd = sh.hidden;
    const long M2 = cls_only ? sh.Bb : M;
    const long pd = cls_only ? (long)sh.N * D : D, ph = cls_only ? (long)sh.N * Hd : Hd;
    S3D_REQUIRE(w.dx_b_lo && w.dh_lo && w.dqkv_lo && w.datt_lo && a.hpre_lo && a.xn1_lo && a.xn2_lo && a.att_lo && a.hact_lo && a.qkv_lo,
                "split-precision backward: every lo plane (gradient scratch, saved activations, hpre_lo) is required");
    if (cls_only) S3D_REQUIRE(w.dx_b_lo_cls && w.datt_lo_cls, "split-precision backward: cls_only_block needs dx_b_lo_cls / datt_lo_cls");
    float* dxb = cls_only ? w.dx_b_cls : w.dx_b;
    bf16_t* dxb_hi = cls_only ? w.dx_b_bf_cls : w.dx_b_bf;
    bf16_t* dxb_lo = cls_only ? w.dx_b_lo_cls : w.dx_b_lo;
    bf16_t* datt_hi = cls_only ? w.datt_cls : w.datt;
    bf16_t* datt_lo = cls_only ? w.datt_lo_cls : w.datt_lo;
    auto wgrad_s = [&](const bf16_t* dy_hi, const bf16_t* dy_lo, int out, long ld_dy, const bf16_t* x_hi, const bf16_t* x_lo, int in, long ld_x,
                       long rows, float* dW, float* db) {
        GemmArgs g = wgrad_args(dy_hi, out, x_hi, in, rows, dW, db, ld_dy, ld_x);
        g.A_lo = dy_lo; g.B_lo = x_lo;
        return s3d_launch_gemm(true, true, true, EPI_ATOMIC, g, 1, s);
    };
    // ---- MLP branch: d(x_out) in dx_a (fp32) / dx_a_bf + dx_a_lo
    GemmArgs g = gemm_zero();   // dh = (dx_out @ W2) * gelu'(hpre)
    g.A_hi = w.dx_a_bf; g.A_lo = w.dx_a_lo; g.lda = pd; g.B_hi = p.fc2_w_hi; g.B_lo = p.fc2_w_lo; g.ldb = Hd; g.M = (int)M2; g.N = Hd; g.K = D;
    g.aux = a.hpre; g.aux_lo = a.hpre_lo; g.ldaux = ph; g.O_hi = w.dh; g.O_lo = w.dh_lo; g.ldo = ph;
    S3D_TRY(s3d_launch_gemm(false, true, true, EPI_DGELU, g, 1, s));
    S3D_TRY(wgrad_s(w.dx_a_bf, w.dx_a_lo, D, pd, a.hact_hi, a.hact_lo, Hd, ph, M2, gr.fc2_w, gr.fc2_b));
    g = gemm_zero();            // dxn2 = dh @ W1
    g.A_hi = w.dh; g.A_lo = w.dh_lo; g.lda = ph; g.B_hi = p.fc1_w_hi; g.B_lo = p.fc1_w_lo; g.ldb = D; g.M = (int)M2; g.N = D; g.K = Hd;
    g.C = w.dxn; g.ldc = pd;
    S3D_TRY(s3d_launch_gemm(false, true, true, EPI_F32, g, 1, s));
    S3D_TRY(wgrad_s(w.dh, w.dh_lo, Hd, ph, a.xn2_hi, a.xn2_lo, D, pd, M2, gr.fc1_w, gr.fc1_b));
    LnBwdArgs lb;
    memset(&lb, 0, sizeof(lb));
    lb.dy = w.dxn; lb.lddy = pd; lb.x = a.x_mid; lb.ldx = pd; lb.mean = a.mean2; lb.rstd = a.rstd2; lb.gamma = p.ln2_w;
    lb.dres = w.dx_a; lb.lddres = pd; lb.dx = dxb; lb.lddx = pd; lb.dx_bf = dxb_hi; lb.dx_bf_lo = dxb_lo; lb.lddxbf = pd;
    lb.dgamma = gr.ln2_w; lb.dbeta = gr.ln2_b; lb.rows = M2; lb.D = D;
    if (lp) { lb.partial = lp->slot(w, D, gr.ln2_w, gr.ln2_b); lb.partial_blocks = w.ln_partial_blocks; }
    S3D_TRY(s3d_launch_ln_bwd(lb, s));
    // ---- attention branch: d(x_mid) in dx_b
    g = gemm_zero();            // datt = dx_mid @ Wproj
    g.A_hi = dxb_hi; g.A_lo = dxb_lo; g.lda = pd; g.B_hi = p.proj_w_hi; g.B_lo = p.proj_w_lo; g.ldb = D; g.M = (int)M2; g.N = D; g.K = D;
    g.O_hi = datt_hi; g.O_lo = datt_lo; g.ldo = pd;
    S3D_TRY(s3d_launch_gemm(false, true, true, EPI_BF16_BIAS, g, 1, s));
    S3D_TRY(wgrad_s(dxb_hi, dxb_lo, D, pd, a.att_hi, a.att_lo, D, pd, M2, gr.proj_w, gr.proj_b));
    AttnArgs at;
    memset(&at, 0, sizeof(at));
    at.qkv_hi = a.qkv_hi; at.qkv_lo = a.qkv_lo; at.ld = 3 * D; at.out_hi = a.att_hi; at.out_lo = a.att_lo;
    at.ldo = D; at.lse = a.lse; at.Bb = sh.Bb; at.H = sh.H; at.N = sh.N; at.D = D; at.sb = sh.N; at.st = 1;
    at.scale = 1.0f / sqrtf((float)(D / sh.H));
    at.dout = datt_hi; at.dout_lo = datt_lo; at.lddo = D; at.dqkv = w.dqkv; at.dqkv_lo = w.dqkv_lo; at.lddq = 3 * D; at.delta = w.delta;
    S3D_TRY(s3d_launch_attention_bwd(at, s));
    g = gemm_zero();            // dxn1 = dqkv @ Wqkv
    g.A_hi = w.dqkv; g.A_lo = w.dqkv_lo; g.lda = 3 * D; g.B_hi = p.qkv_w_hi; g.B_lo = p.qkv_w_lo; g.ldb = D; g.M = (int)M; g.N = D; g.K = 3 * D;
    g.C = w.dxn; g.ldc = D;
    S3D_TRY(s3d_launch_gemm(false, true, true, EPI_F32, g, 1, s));
    S3D_TRY(wgrad_s(w.dqkv, w.dqkv_lo, 3 * D, 3 * D, a.xn1_hi, a.xn1_lo, D, D, M, gr.qkv_w, gr.qkv_b));
    lb.dy = w.dxn; lb.lddy = D; lb.ldx = D; lb.lddres = D; lb.lddx = D; lb.lddxbf = D; lb.rows = M;
    lb.x = a.x_in; lb.mean = a.mean1; lb.rstd = a.rstd1; lb.gamma = p.ln1_w; lb.dres = dxb; lb.dx = w.dx_a;
    lb.dx_bf = w.dx_a_bf; lb.dx_bf_lo = w.dx_a_lo; lb.dgamma = gr.ln1_w; lb.dbeta = gr.ln1_b;
    if (lp) lb.partial = lp->slot(w, D, gr.ln1_w, gr.ln1_b);
    S3D_TRY(s3d_launch_ln_bwd(lb, s));
    return 0;
}

// fq: optimizer shares (adam_fill.h) that ride on this block's launches -- the GEMM parameters of the block whose backward has just retired.
// Shares in sixteenths, roughly the launches' durations (16.8 / 12.9 / 6.0 / 7.5 / 7.9 / 12.9 / 6.0 us at cfg-2).
int block_bwd_chain(const S3dBlockShape& sh, const S3dBlockParams& p, const S3dBlockGrads& gr, const S3dBlockActs& a,
                    const S3dBlockScratch& w, hipStream_t s, LnPartials* lp, WgradDefer& wd, bool last_of_call);

int block_bwd(const S3dBlockShape& sh, const S3dBlockParams& p, const S3dBlockGrads& gr, const S3dBlockActs& a,
              const S3dBlockScratch& w, hipStream_t s, LnPartials* lp = nullptr, bool cls_only = false, AdamFillQueue* fq = nullptr,
              WgradDefer* wd = nullptr, bool last_of_call = true) {
    if (w.dx_a_lo != nullptr) return block_bwd_split(sh, p, gr, a, w, s, lp, cls_only);
    if (wd != nullptr && !cls_only) return block_bwd_chain(sh, p, gr, a, w, s, lp, *wd, last_of_call);
    auto share = [&](int sixteenths) { if (fq) fq->share16 = sixteenths; return fq; };
    const long M = (long)sh.Bb * sh.N;
    const int D = sh.D, Hd = sh.hidden;
    // cls_only (see block_fwd): d(x_out) is non-zero at the class rows only and proj / norm2 / mlp are row-local -> their backward
    // runs on those Bb rows (row pitch N*D); d(x_mid) and d(att) go to dedicated buffers whose other rows stay zero
    const long M2 = cls_only ? sh.Bb : M;
    const long pd = cls_only ? (long)sh.N * D : D, ph = cls_only ? (long)sh.N * Hd : Hd;
    float* dxb = cls_only ? w.dx_b_cls : w.dx_b;
    bf16_t* dxb_bf = cls_only ? w.dx_b_bf_cls : w.dx_b_bf;
    bf16_t* datt = cls_only ? w.datt_cls : w.datt;
    BwdStreams* bs = (bwd_streams_enabled() && g_bwd_streams.init(s)) ? &g_bwd_streams : nullptr;
    // ---- MLP branch: d(x_out) is in dx_a / dx_a_bf
    // every dgrad is launched together with the wgrad that consumes the same dy (one grid, two problems)
    GemmArgs g = gemm_zero();   // dh = (dx_out @ W2) * gelu'(hpre)          || dW2 += dx_out^T hact
    g.A_hi = w.dx_a_bf; g.lda = pd; g.B_hi = p.fc2_w_hi; g.ldb = Hd; g.M = (int)M2; g.N = Hd; g.K = D;
    g.aux = a.hpre; g.ldaux = ph; g.O_hi = w.dh; g.ldo = ph;
    S3D_TRY(dgrad_and_wgrad(EPI_DGELU, g, wgrad_args(w.dx_a_bf, D, a.hact_hi, Hd, M2, gr.fc2_w, gr.fc2_b, pd, ph), s, bs, share(4)));
    g = gemm_zero();            // dxn2 = dh @ W1                             || dW1 += dh^T xn2
    g.A_hi = w.dh; g.lda = ph; g.B_hi = p.fc1_w_hi; g.ldb = D; g.M = (int)M2; g.N = D; g.K = Hd; g.C = w.dxn; g.ldc = pd;
    S3D_TRY(dgrad_and_wgrad(EPI_F32, g, wgrad_args(w.dh, Hd, a.xn2_hi, D, M2, gr.fc1_w, gr.fc1_b, ph, pd), s, bs, share(3)));
    LnBwdArgs lb;
    memset(&lb, 0, sizeof(lb));
    lb.dy = w.dxn; lb.lddy = pd; lb.x = a.x_mid; lb.ldx = pd; lb.mean = a.mean2; lb.rstd = a.rstd2; lb.gamma = p.ln2_w;
    lb.dres = w.dx_a; lb.lddres = pd; lb.dx = dxb; lb.lddx = pd; lb.dx_bf = dxb_bf; lb.lddxbf = pd;
    lb.dgamma = gr.ln2_w; lb.dbeta = gr.ln2_b; lb.rows = M2; lb.D = D;
    if (lp) { lb.partial = lp->slot(w, D, gr.ln2_w, gr.ln2_b); lb.partial_blocks = w.ln_partial_blocks; }
    S3D_TRY(s3d_launch_ln_bwd(lb, s, share(1)));
    // ---- attention branch: d(x_mid) is in dx_b / dx_b_bf
    AttnArgs at;
    memset(&at, 0, sizeof(at));
    at.qkv_hi = a.qkv_hi; at.qkv_lo = a.qkv_lo; at.ld = 3 * D; at.out_hi = a.att_hi; at.out_lo = sh.split ? a.att_lo : nullptr;
    at.ldo = D; at.lse = a.lse; at.Bb = sh.Bb; at.H = sh.H; at.N = sh.N; at.D = D; at.sb = sh.N; at.st = 1;
    at.scale = 1.0f / sqrtf((float)(D / sh.H));
    // Fused (S3dBlockShape::fuse == 0, small token counts, dense block): attn.proj's dgrad runs inside the attention-backward launch, head
    // slice by head slice (fused_block.hip: blk_attn_bwd_kernel), and its wgrad rides on the qkv pair launch as a third problem: six
    // launches per block instead of seven.
    const bool fuse_bwd = sh.fuse == 0 && !cls_only && bs == nullptr && s3d_fused_attn_bwd_ok(sh.Bb, sh.N, D, sh.H);
    const GemmArgs proj_wg = wgrad_args(dxb_bf, D, a.att_hi, D, M2, gr.proj_w, gr.proj_b, pd, pd);
    if (fuse_bwd) {
        FusedAttnBwdArgs fb{};
        fb.dxm = dxb_bf; fb.lddxm = D; fb.w_hi = p.proj_w_hi; fb.qkv_hi = a.qkv_hi;
        fb.lse = a.lse; fb.dqkv = w.dqkv; fb.Bb = sh.Bb; fb.N = sh.N; fb.H = sh.H; fb.scale = at.scale;
        fb.lse_packed = s3d_attention_pairs_packed(at) ? 1 : 0;
        S3D_TRY(s3d_launch_fused_attn_bwd(fb, D, s));
    } else {
        g = gemm_zero();        // datt = dx_mid @ Wproj                      || dWproj += dx_mid^T att
        g.A_hi = dxb_bf; g.lda = pd; g.B_hi = p.proj_w_hi; g.ldb = D; g.M = (int)M2; g.N = D; g.K = D; g.O_hi = datt; g.ldo = pd;
        S3D_TRY(dgrad_and_wgrad(EPI_BF16_BIAS, g, proj_wg, s, bs, share(2)));
        at.dout = datt; at.lddo = D; at.dqkv = w.dqkv; at.lddq = 3 * D; at.delta = w.delta;
        S3D_TRY(s3d_launch_attention_bwd(at, s, share(2)));
    }
    g = gemm_zero();            // dxn1 = dqkv @ Wqkv                         || dWqkv += dqkv^T xn1  [|| dWproj += dx_mid^T att]
    g.A_hi = w.dqkv; g.lda = 3 * D; g.B_hi = p.qkv_w_hi; g.ldb = D; g.M = (int)M; g.N = D; g.K = 3 * D; g.C = w.dxn; g.ldc = D;
    S3D_TRY(dgrad_and_wgrad(EPI_F32, g, wgrad_args(w.dqkv, 3 * D, a.xn1_hi, D, M, gr.qkv_w, gr.qkv_b), s, bs, share(fuse_bwd ? 7 : 3),
                            fuse_bwd ? &proj_wg : nullptr));
    S3D_TRY(bwd_streams_join(s, bs));      // norm1 overwrites dx_a_bf (read by the fc2 wgrad); the next block reuses dh / dqkv / dx_b_bf
    lb.dy = w.dxn; lb.lddy = D; lb.ldx = D; lb.lddres = D; lb.lddx = D; lb.lddxbf = D; lb.rows = M;      // norm1 is dense again
    lb.x = a.x_in; lb.mean = a.mean1; lb.rstd = a.rstd1; lb.gamma = p.ln1_w; lb.dres = dxb; lb.dx = w.dx_a;
    lb.dx_bf = w.dx_a_bf; lb.dgamma = gr.ln1_w; lb.dbeta = gr.ln1_b;
    if (lp) lb.partial = lp->slot(w, D, gr.ln1_w, gr.ln1_b);
    S3D_TRY(s3d_launch_ln_bwd(lb, s, share(16)));        // whatever is left of the current range
    return 0;
}

// The dense block on the dgrad chain (see WgradDefer): fc2 dgrad * gelu' -> fc1 dgrad (k-slices) -> norm2 -> fused proj dgrad + attention
// backward -> qkv dgrad (k-slices) -> [grouped wgrads of the pending blocks] -> norm1.  Six launches on the critical path, none of them
// carries a wgrad.
int block_bwd_chain(const S3dBlockShape& sh, const S3dBlockParams& p, const S3dBlockGrads& gr, const S3dBlockActs& a,
                    const S3dBlockScratch& w, hipStream_t s, LnPartials* lp, WgradDefer& wd, bool last_of_call) {
    const long M = (long)sh.Bb * sh.N;
    const int D = sh.D, Hd = sh.hidden;
    const WgSlot S = wd.slot(wd.next);
    const bf16_t* dxa = wd.dxa_cur;
    const long plane = M * D;
    // fused: the two LayerNorm backward launches run as epilogues of the fc1 / qkv dgrads (bwd_gemm.hip, "Row statistics"): four launches
    // per block.  aux = this block's [u2 | c2 | u1 | c1]; rowstat = [norm2: s1, s2 | norm1: s1, s2], each [M]
    // fa: attn.proj's dgrad + the attention backward as one launch (small token counts); otherwise -- the point path's 257-token sequences --
    // the library GEMM tiles and the generic attention backward run the chain, and only the wgrads change (deferred, grouped, full K)
    const bool fa = s3d_fused_attn_bwd_ok(sh.Bb, sh.N, D, sh.H);
    const bool fused = wd.aux != nullptr && fa;
    const float* aux = fused ? wd.aux + (size_t)wd.block * 2 * (Hd + 3 * D) : nullptr;
    float* rs = w.ln_rowstat;
    const int nty = (int)((M + 63) / 64);
    S3dRowStats st;
    memset(&st, 0, sizeof(st));
    GemmArgs g = gemm_zero();   // dh = (dx_out @ W2) * gelu'(hpre)   [+ norm2's row statistics; clears norm1's]
    g.A_hi = dxa; g.lda = D; g.B_hi = p.fc2_w_hi; g.ldb = Hd; g.M = (int)M; g.N = Hd; g.K = D;
    g.aux = a.hpre; g.ldaux = Hd; g.O_hi = S.dh; g.ldo = Hd;
    if (fused) {
        st.u = aux; st.c = aux + Hd; st.rs1 = rs; st.rs2 = rs + M; st.zero_buf = rs + 2 * M; st.zero_n = (int)(2 * M);
        S3D_TRY(s3d_launch_dgrad_dgelu(g, &st, s));
    } else {
        S3D_TRY(s3d_launch_gemm(false, true, false, EPI_DGELU, g, 1, s));
    }
    wd.push(dxa, D, a.hact_hi, Hd, gr.fc2_w, gr.fc2_b);
    g = gemm_zero();            // dxn2 = dh @ W1   [-> norm2 backward -> d(x_mid)]
    g.A_hi = S.dh; g.lda = Hd; g.B_hi = p.fc1_w_hi; g.ldb = D; g.M = (int)M; g.N = D; g.K = Hd; g.C = w.dxn; g.ldc = D;
    LnBwdArgs lb;
    memset(&lb, 0, sizeof(lb));
    lb.dy = w.dxn; lb.lddy = D;
    lb.x = a.x_mid; lb.ldx = D; lb.mean = a.mean2; lb.rstd = a.rstd2; lb.gamma = p.ln2_w;
    lb.dres = w.dx_a; lb.lddres = D; lb.dx = w.dx_b; lb.lddx = D; lb.dx_bf = S.dxb; lb.lddxbf = D;
    lb.dgamma = gr.ln2_w; lb.dbeta = gr.ln2_b; lb.rows = M; lb.D = D;
    if (fused) {
        if (lp) { lb.partial = lp->slot(w, D, gr.ln2_w, gr.ln2_b, nty); lb.partial_blocks = w.ln_partial_blocks; }
        st.u = st.c = nullptr; st.rs1 = rs; st.rs2 = rs + M; st.zero_buf = nullptr; st.zero_n = 0;
        S3D_TRY(s3d_launch_dgrad_lnbwd(g, lb, &st, s));
    } else {
        // 192-wide layers at many rows (the point path): 64 x 192 tiles hold whole rows, the LayerNorm backward is the dgrad's epilogue
        const bool rows_fused = !fa && nty <= w.ln_partial_blocks && lp != nullptr && !s3d_deterministic() && s3d_dgrad_lnrows_ok(g, lb);
        if (rows_fused) {
            lb.partial = lp->slot(w, D, gr.ln2_w, gr.ln2_b, nty); lb.partial_blocks = w.ln_partial_blocks;
            S3D_TRY(s3d_launch_dgrad_lnrows(g, lb, s));
        } else {
            if (fa) {
                const int sl1 = s3d_dgrad_splitk_slices(Hd, wd.splitk);
                S3D_TRY(s3d_launch_dgrad_splitk(g, sl1, plane, s));
                lb.dy_parts = sl1; lb.dy_part_stride = plane;
            } else {
                S3D_TRY(s3d_launch_gemm(false, true, false, EPI_F32, g, 1, s));
            }
            if (lp) { lb.partial = lp->slot(w, D, gr.ln2_w, gr.ln2_b); lb.partial_blocks = w.ln_partial_blocks; }
            S3D_TRY(s3d_launch_ln_bwd(lb, s));
        }
    }
    wd.push(S.dh, Hd, a.xn2_hi, D, gr.fc1_w, gr.fc1_b);
    AttnArgs at;
    memset(&at, 0, sizeof(at));
    at.Bb = sh.Bb; at.H = sh.H; at.N = sh.N; at.D = D; at.sb = sh.N; at.st = 1;
    at.scale = 1.0f / sqrtf((float)(D / sh.H));
    if (!fa) {
        g = gemm_zero();        // datt = dx_mid @ Wproj, then dq / dk / dv
        g.A_hi = S.dxb; g.lda = D; g.B_hi = p.proj_w_hi; g.ldb = D; g.M = (int)M; g.N = D; g.K = D; g.O_hi = w.datt; g.ldo = D;
        S3D_TRY(s3d_launch_gemm(false, true, false, EPI_BF16_BIAS, g, 1, s));
        at.qkv_hi = a.qkv_hi; at.qkv_lo = a.qkv_lo; at.ld = 3 * D; at.out_hi = a.att_hi; at.out_lo = sh.split ? a.att_lo : nullptr;
        at.ldo = D; at.lse = a.lse;
        at.dout = w.datt; at.lddo = D; at.dqkv = S.dqkv; at.lddq = 3 * D; at.delta = w.delta;
        S3D_TRY(s3d_launch_attention_bwd(at, s));
    }
    FusedAttnBwdArgs fb{};      // datt = dx_mid @ Wproj per head slice, then dq / dk / dv   [+ norm1's row statistics]
    fb.dxm = S.dxb; fb.lddxm = D; fb.w_hi = p.proj_w_hi; fb.qkv_hi = a.qkv_hi;
    fb.lse = a.lse; fb.dqkv = S.dqkv; fb.Bb = sh.Bb; fb.N = sh.N; fb.H = sh.H; fb.scale = 1.0f / sqrtf((float)(D / sh.H));
    fb.lse_packed = s3d_attention_pairs_packed(at) ? 1 : 0;
    if (fused) { fb.st_u = aux + 2 * Hd; fb.st_c = aux + 2 * Hd + 3 * D; fb.st_s1 = rs + 2 * M; fb.st_s2 = rs + 3 * M; }
    if (fa) S3D_TRY(s3d_launch_fused_attn_bwd(fb, D, s));
    wd.push(S.dxb, D, a.att_hi, D, gr.proj_w, gr.proj_b);
    wd.push(S.dqkv, 3 * D, a.xn1_hi, D, gr.qkv_w, gr.qkv_b);
    ++wd.pending_blocks;
    // norm1's backward writes the NEXT block's d(x_out): the next ring slot, whose previous content a pending wgrad may still read when the
    // ring wraps -- or the caller's dx_a_bf when the call ends, which the first block of the call read.  So the grouped wgrads of the
    // pending blocks go first (fused: the qkv dgrad carries that LayerNorm backward, so they go in front of it; all their operands exist).
    const bool full = wd.pending_blocks >= wd.slots;
    const bool flush_now = full || last_of_call;
    const int next = full ? 0 : wd.next + 1;
    bf16_t* dxa_next = last_of_call ? w.dx_a_bf : wd.slot(next).dxa;
    g = gemm_zero();            // dxn1 = dqkv @ Wqkv   [-> norm1 backward -> d(x_in)]
    g.A_hi = S.dqkv; g.lda = 3 * D; g.B_hi = p.qkv_w_hi; g.ldb = D; g.M = (int)M; g.N = D; g.K = 3 * D; g.C = w.dxn; g.ldc = D;
    lb.dy = w.dxn; lb.x = a.x_in; lb.mean = a.mean1; lb.rstd = a.rstd1; lb.gamma = p.ln1_w; lb.dres = w.dx_b; lb.dx = w.dx_a;
    lb.dx_bf = dxa_next; lb.dgamma = gr.ln1_w; lb.dbeta = gr.ln1_b;
    if (fused) {
        if (flush_now) S3D_TRY(wd.flush(s));
        if (lp) lb.partial = lp->slot(w, D, gr.ln1_w, gr.ln1_b, nty);
        st.rs1 = rs + 2 * M; st.rs2 = rs + 3 * M; st.zero_buf = rs; st.zero_n = (int)(2 * M);      // norm2's statistics are consumed: cleared here
        S3D_TRY(s3d_launch_dgrad_lnbwd(g, lb, &st, s));
    } else {
        lb.dy_parts = 0; lb.dy_part_stride = 0;
        const bool rows_fused = !fa && nty <= w.ln_partial_blocks && lp != nullptr && !s3d_deterministic() && s3d_dgrad_lnrows_ok(g, lb);
        if (rows_fused) {
            if (flush_now) S3D_TRY(wd.flush(s));                     // (the epilogue writes the next block's d(x_out): pending wgrads first)
            lb.partial = lp->slot(w, D, gr.ln1_w, gr.ln1_b, nty); lb.partial_blocks = w.ln_partial_blocks;
            S3D_TRY(s3d_launch_dgrad_lnrows(g, lb, s));
            wd.next = next;
            wd.dxa_cur = dxa_next;
            return 0;
        }
        if (fa) {
            const int sl2 = s3d_dgrad_splitk_slices(3 * D, wd.splitk);
            S3D_TRY(s3d_launch_dgrad_splitk(g, sl2, plane, s));
            lb.dy_parts = sl2; lb.dy_part_stride = plane;
        } else {
            S3D_TRY(s3d_launch_gemm(false, true, false, EPI_F32, g, 1, s));
        }
        if (flush_now) S3D_TRY(wd.flush(s));
        if (lp) lb.partial = lp->slot(w, D, gr.ln1_w, gr.ln1_b);
        S3D_TRY(s3d_launch_ln_bwd(lb, s));
    }
    wd.next = next;
    wd.dxa_cur = dxa_next;
    return 0;
}

// ---- seq-first post-norm encoder layer (group_embed) ----
// dropout sites of nn.TransformerEncoderLayer: 0 attention weights, 1 after out_proj, 2 after the ReLU, 3 after linear2
struct DropCfg { const unsigned long long* seed; unsigned thr; float scale; };
DropCfg drop_cfg(const S3dEncShape& sh) {
    DropCfg d{sh.seed, 0u, 1.f};
    if (sh.dropout_p > 0.f) {
        d.thr = (unsigned)((double)sh.dropout_p * 4294967296.0);
        d.scale = 1.0f / (1.0f - sh.dropout_p);
    }
    return d;
}
#define SET_DROP(obj, site) do { (obj).drop_seed = dc.seed; (obj).drop_site = (site); (obj).drop_thr = dc.thr; (obj).drop_scale = dc.scale; } while (0)

int enc_fwd(const S3dEncShape& sh, const S3dEncParams& p, const S3dEncActs& a, hipStream_t s) {
    const long M = (long)sh.G * sh.Nb;
    const int D = sh.D, F = sh.Dff;
    const bool split = sh.split != 0;
    const DropCfg dc = drop_cfg(sh);
    S3D_REQUIRE(dc.thr == 0 || dc.seed != nullptr, "encoder layer: dropout needs a device-resident seed");
    S3D_REQUIRE(M < (1L << 31), "encoder layer: too many rows");
    S3D_TRY(s3d_launch_split(a.x_in, a.xin_hi, split ? a.xin_lo : nullptr, M, D, D, s));
    GemmArgs g = gemm_zero();                       // qkv = x @ Win^T + b
    g.A_hi = a.xin_hi; g.A_lo = a.xin_lo; g.lda = D; g.B_hi = p.in_w_hi; g.B_lo = p.in_w_lo; g.ldb = D;
    g.M = (int)M; g.N = 3 * D; g.K = D; g.bias = p.in_b; g.O_hi = a.qkv_hi; g.O_lo = split ? a.qkv_lo : nullptr; g.ldo = 3 * D;
    S3D_TRY(s3d_launch_gemm(false, false, split, EPI_BF16_BIAS, g, 1, s));
    AttnArgs at;                                    // attention over the group axis: row(b=t, token=g) = t + g*Nb
    memset(&at, 0, sizeof(at));
    at.qkv_hi = a.qkv_hi; at.qkv_lo = a.qkv_lo; at.ld = 3 * D; at.out_hi = a.att_hi; at.out_lo = split ? a.att_lo : nullptr;
    at.ldo = D; at.lse = a.lse; at.Bb = sh.Nb; at.H = sh.H; at.N = sh.G; at.D = D; at.sb = 1; at.st = sh.Nb;
    at.scale = 1.0f / sqrtf((float)(D / sh.H));
    SET_DROP(at, 0);
    at.drop_mask = a.attn_mask;
    at.p_single_plane = s3d_knob(2) == 0 ? 0 : 1;     // one bf16 plane of P in the P V product (S3dAttnArgs::p_single_plane; knob 2 = 0: A/B against the full split)
    S3D_TRY(s3d_launch_attention_fwd(at, split, s));
    g = gemm_zero();                                // s1 = x + drop(att @ Wo^T + bo)
    g.A_hi = a.att_hi; g.A_lo = a.att_lo; g.lda = D; g.B_hi = p.out_w_hi; g.B_lo = p.out_w_lo; g.ldb = D;
    g.M = (int)M; g.N = D; g.K = D; g.bias = p.out_b; g.R = a.x_in; g.ldr = D; g.C = a.s1; g.ldc = D;
    SET_DROP(g, 1);
    S3D_TRY(s3d_launch_gemm(false, false, split, EPI_RESID, g, 1, s));
    LnArgs ln;                                      // x1 = LN1(s1)
    memset(&ln, 0, sizeof(ln));
    ln.x = a.s1; ln.ldx = D; ln.rows = M; ln.D = D; ln.eps = sh.eps; ln.gamma = p.n1_w; ln.beta = p.n1_b;
    ln.out_hi = a.x1_hi; ln.out_lo = split ? a.x1_lo : nullptr; ln.out_f32 = a.x1; ln.ldo = D; ln.mean = a.mean1; ln.rstd = a.rstd1;
    S3D_TRY(s3d_launch_ln_fwd(ln, s));
    g = gemm_zero();                                // f = relu(x1 @ W1^T + b1)
    g.A_hi = a.x1_hi; g.A_lo = a.x1_lo; g.lda = D; g.B_hi = p.l1_w_hi; g.B_lo = p.l1_w_lo; g.ldb = D;
    g.M = (int)M; g.N = F; g.K = D; g.bias = p.l1_b; g.aux = a.fpre; g.aux_lo = a.fpre_lo; g.ldaux = F; g.O_hi = a.f_hi; g.O_lo = split ? a.f_lo : nullptr; g.ldo = F;
    SET_DROP(g, 2);
    S3D_TRY(s3d_launch_gemm(false, false, split, EPI_RELU, g, 1, s));
    g = gemm_zero();                                // s2 = x1 + drop(f @ W2^T + b2)
    g.A_hi = a.f_hi; g.A_lo = a.f_lo; g.lda = F; g.B_hi = p.l2_w_hi; g.B_lo = p.l2_w_lo; g.ldb = F;
    g.M = (int)M; g.N = D; g.K = F; g.bias = p.l2_b; g.R = a.x1; g.ldr = D; g.C = a.s2; g.ldc = D;
    SET_DROP(g, 3);
    S3D_TRY(s3d_launch_gemm(false, false, split, EPI_RESID, g, 1, s));
    ln.x = a.s2; ln.gamma = p.n2_w; ln.beta = p.n2_b; ln.out_hi = nullptr; ln.out_lo = nullptr; ln.out_f32 = a.x_out;
    ln.mean = a.mean2; ln.rstd = a.rstd2;           // x_out = LN2(s2)
    S3D_TRY(s3d_launch_ln_fwd(ln, s));
    return 0;
}

int enc_bwd(const S3dEncShape& sh, const S3dEncParams& p, const S3dEncGrads& gr, const S3dEncActs& a,
            const S3dBlockScratch& w, hipStream_t s) {
    const long M = (long)sh.G * sh.Nb;
    const int D = sh.D, F = sh.Dff;
    const DropCfg dc = drop_cfg(sh);
    // sp: split-precision parity mode (S3dBlockScratch::dx_a_lo) -- every GEMM a three-MFMA split product on hi + lo operands, no
    // split-K, fp32 attention backward, gradients as hi + lo pairs; the chain itself is the same
    const bool sp = w.dx_a_lo != nullptr;
    if (sp) S3D_REQUIRE(w.dx_b_lo && w.dh_lo && w.dqkv_lo && w.datt_lo && a.fpre_lo && a.xin_lo && a.x1_lo && a.f_lo && a.att_lo && a.qkv_lo,
                        "split-precision backward of the encoder layer: every lo plane is required");
    auto wgrad_x = [&](const bf16_t* dy_hi, const bf16_t* dy_lo, int out, const bf16_t* x_hi, const bf16_t* x_lo, int in, float* dW, float* db) {
        GemmArgs g = wgrad_args(dy_hi, out, x_hi, in, M, dW, db);
        if (!sp) return s3d_launch_gemm(true, true, false, EPI_ATOMIC, g, 0, s);
        g.A_lo = dy_lo; g.B_lo = x_lo;
        return s3d_launch_gemm(true, true, true, EPI_ATOMIC, g, 1, s);
    };
    LnBwdArgs lb;                                   // ds2 = LN2'(dx_out)            -> dx_b (+bf16, masked by site 3)
    memset(&lb, 0, sizeof(lb));
    SET_DROP(lb, 3);
    lb.dy = w.dx_a; lb.lddy = D; lb.x = a.s2; lb.ldx = D; lb.mean = a.mean2; lb.rstd = a.rstd2; lb.gamma = p.n2_w;
    lb.dx = w.dx_b; lb.lddx = D; lb.dx_bf = w.dx_b_bf; lb.dx_bf_lo = sp ? w.dx_b_lo : nullptr; lb.lddxbf = D;
    lb.dgamma = gr.n2_w; lb.dbeta = gr.n2_b; lb.rows = M; lb.D = D;
    S3D_TRY(s3d_launch_ln_bwd(lb, s));
    S3D_TRY(wgrad_x(w.dx_b_bf, w.dx_b_lo, D, a.f_hi, a.f_lo, F, gr.l2_w, gr.l2_b));
    GemmArgs g = gemm_zero();                       // df = (ds2 @ W2) * relu'(fpre)   -> dh
    g.A_hi = w.dx_b_bf; g.lda = D; g.B_hi = p.l2_w_hi; g.ldb = F; g.M = (int)M; g.N = F; g.K = D;
    g.aux = a.fpre; g.ldaux = F; g.O_hi = w.dh; g.ldo = F;
    if (sp) { g.A_lo = w.dx_b_lo; g.B_lo = p.l2_w_lo; g.aux_lo = a.fpre_lo; g.O_lo = w.dh_lo; }
    SET_DROP(g, 2);
    S3D_TRY(s3d_launch_gemm(false, true, sp, EPI_DRELU, g, 1, s));
    S3D_TRY(wgrad_x(w.dh, w.dh_lo, F, a.x1_hi, a.x1_lo, D, gr.l1_w, gr.l1_b));
    g = gemm_zero();                                // g1 = df @ W1 + ds2              -> dxn
    g.A_hi = w.dh; g.lda = F; g.B_hi = p.l1_w_hi; g.ldb = D; g.M = (int)M; g.N = D; g.K = F; g.R = w.dx_b; g.ldr = D;
    g.C = w.dxn; g.ldc = D;
    if (sp) { g.A_lo = w.dh_lo; g.B_lo = p.l1_w_lo; }
    S3D_TRY(s3d_launch_gemm(false, true, sp, EPI_RESID, g, 1, s));
    lb.dy = w.dxn; lb.x = a.s1; lb.mean = a.mean1; lb.rstd = a.rstd1; lb.gamma = p.n1_w; lb.dx = w.dx_a; lb.dx_bf = w.dx_a_bf;
    lb.dx_bf_lo = sp ? w.dx_a_lo : nullptr;
    lb.dgamma = gr.n1_w; lb.dbeta = gr.n1_b;        // ds1 = LN1'(g1)                  -> dx_a (+bf16, masked by site 1)
    SET_DROP(lb, 1);
    S3D_TRY(s3d_launch_ln_bwd(lb, s));
    S3D_TRY(wgrad_x(w.dx_a_bf, w.dx_a_lo, D, a.att_hi, a.att_lo, D, gr.out_w, gr.out_b));
    g = gemm_zero();                                // datt = ds1 @ Wo
    g.A_hi = w.dx_a_bf; g.lda = D; g.B_hi = p.out_w_hi; g.ldb = D; g.M = (int)M; g.N = D; g.K = D; g.O_hi = w.datt; g.ldo = D;
    if (sp) { g.A_lo = w.dx_a_lo; g.B_lo = p.out_w_lo; g.O_lo = w.datt_lo; }
    S3D_TRY(s3d_launch_gemm(false, true, sp, EPI_BF16_BIAS, g, 1, s));
    AttnArgs at;
    memset(&at, 0, sizeof(at));
    at.qkv_hi = a.qkv_hi; at.qkv_lo = a.qkv_lo; at.ld = 3 * D; at.out_hi = a.att_hi; at.out_lo = sh.split ? a.att_lo : nullptr;
    at.ldo = D; at.lse = a.lse; at.Bb = sh.Nb; at.H = sh.H; at.N = sh.G; at.D = D; at.sb = 1; at.st = sh.Nb;
    at.scale = 1.0f / sqrtf((float)(D / sh.H));
    at.dout = w.datt; at.lddo = D; at.dqkv = w.dqkv; at.lddq = 3 * D; at.delta = w.delta;
    if (sp) { at.dout_lo = w.datt_lo; at.dqkv_lo = w.dqkv_lo; }
    SET_DROP(at, 0);
    at.drop_mask = sp ? nullptr : a.attn_mask;     // (the split-precision reference kernels evaluate the hash)
    S3D_TRY(s3d_launch_attention_bwd(at, s));
    S3D_TRY(wgrad_x(w.dqkv, w.dqkv_lo, 3 * D, a.xin_hi, a.xin_lo, D, gr.in_w, gr.in_b));
    g = gemm_zero();                                // dx = dqkv @ Win + ds1           -> dx_b (+bf16 copy)
    g.A_hi = w.dqkv; g.lda = 3 * D; g.B_hi = p.in_w_hi; g.ldb = D; g.M = (int)M; g.N = D; g.K = 3 * D; g.R = w.dx_a; g.ldr = D;
    g.C = w.dx_b; g.ldc = D; g.O_hi = w.dx_b_bf; g.ldo = D;
    if (sp) { g.A_lo = w.dqkv_lo; g.B_lo = p.in_w_lo; g.O_lo = w.dx_b_lo; }
    S3D_TRY(s3d_launch_gemm(false, true, sp, EPI_RESID, g, 1, s));
    return 0;
}

}  // namespace

namespace {
// the dgrad chain + grouped wgrads replace the paired launches where the fused attention backward runs (small token counts, dense
// blocks, plain-bf16 backward) and the caller has provided the ring
// weights-only vectors of the row statistics (bwd_gemm.hip) for the dense blocks last .. first, indexed by block: one launch per <= 16 blocks
int blocks_ln_aux(const S3dBlockShape& sh, const S3dBlockParams* p, const S3dBlockScratch& w, int first, int last, hipStream_t s) {
    S3dLnAuxLayer layers[32];
    int nl = 0;
    const size_t per = 2 * ((size_t)sh.hidden + 3 * (size_t)sh.D);
    for (int i = first; i >= last; --i) {
        if (sh.cls_only_block == i + 1) continue;
        float* base = w.ln_aux + (size_t)i * per;
        layers[nl++] = S3dLnAuxLayer{p[i].fc1_w_hi, p[i].fc1_w_lo, p[i].fc1_b, p[i].ln2_w, p[i].ln2_b, base, base + sh.hidden, sh.hidden};
        layers[nl++] = S3dLnAuxLayer{p[i].qkv_w_hi, p[i].qkv_w_lo, p[i].qkv_b, p[i].ln1_w, p[i].ln1_b, base + 2 * sh.hidden, base + 2 * sh.hidden + 3 * sh.D, 3 * sh.D};
        if (nl == 32 || i == last) {
            S3D_TRY(s3d_launch_ln_aux(layers, nl, sh.D, s));
            nl = 0;
        }
    }
    if (nl) S3D_TRY(s3d_launch_ln_aux(layers, nl, sh.D, s));
    return 0;
}
bool wgrad_chain_ok(const S3dBlockShape& sh, const S3dBlockScratch& w) {
    if (w.wg_ring == nullptr || w.wg_slots < 1 || w.dx_a_lo != nullptr || bwd_streams_enabled()) return false;
    // small token counts: the fused attention backward; long sequences at many rows (where a dgrad + wgrad pair is two launches anyway): the
    // library tiles with the wgrads deferred
    if (sh.fuse != 0) return false;
    if (!s3d_fused_attn_bwd_ok(sh.Bb, sh.N, sh.D, sh.H) && (long)sh.Bb * sh.N <= 8192) return false;
    return (sh.D & 7) == 0 && (sh.hidden & 7) == 0 && w.dgrad_splitk >= 0 && w.dgrad_splitk <= 4;
}
}  // namespace

// ---- workspace layout of a block stack (mirrors engine.py::_BlockWorkspace / _BlockScratch; one allocation instead of ~25)
namespace {
struct Carver {
    unsigned char* base; size_t off = 0;
    template <typename T> T* take(size_t count) {
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += (count * sizeof(T) + 255) / 256 * 256;
        return p;
    }
};
// LayerNorm-backward partial rows per LayerNorm: engine.py::ln_partial_blocks
int ws_ln_partial_blocks(long rows, int D) {
    if (D == 192 && rows > 8192) return (int)std::max<long>(416, (rows + 63) / 64);      // one row of partials per 64-row tile (dgrad_lnrows_kernel)
    return rows <= 8192 ? 208 : 416;
}
size_t block_ws_layout(const S3dBlockShape& sh, int depth, int bwd, void* base, S3dBlockActs* acts, S3dBlockScratch* sc, size_t* zoff, size_t* zbytes) {
    const bool ring = (bwd & 2) != 0;            // with_backward = 2 / 3: + the dy ring of the dgrad chain (3 slots, 3 k-slices)
    const size_t M = (size_t)sh.Bb * sh.N, D = sh.D, Hd = sh.hidden, BHN = (size_t)sh.Bb * sh.H * sh.N;
    Carver c{static_cast<unsigned char*>(base)};
    std::vector<float*> x(depth + 1);
    for (int i = 0; i <= depth; ++i) x[i] = c.take<float>(M * D);
    uint16_t* xn1_lo = c.take<uint16_t>(M * D); uint16_t* qkv_lo = c.take<uint16_t>(M * 3 * D);
    uint16_t* xn2_lo = c.take<uint16_t>(M * D); uint16_t* hact_lo = c.take<uint16_t>(M * Hd);
    for (int i = 0; i < depth; ++i) {
        S3dBlockActs a;
        memset(&a, 0, sizeof(a));
        a.x_in = x[i]; a.x_out = x[i + 1]; a.x_mid = c.take<float>(M * D);
        a.mean1 = c.take<float>(M); a.rstd1 = c.take<float>(M); a.mean2 = c.take<float>(M); a.rstd2 = c.take<float>(M);
        a.lse = c.take<float>(BHN);
        a.xn1_hi = c.take<uint16_t>(M * D); a.xn1_lo = xn1_lo; a.qkv_hi = c.take<uint16_t>(M * 3 * D); a.qkv_lo = qkv_lo;
        a.att_hi = c.take<uint16_t>(M * D); a.att_lo = c.take<uint16_t>(M * D);         // att_lo per block: the attention backward's delta reads it
        a.xn2_hi = c.take<uint16_t>(M * D); a.xn2_lo = xn2_lo;
        a.hpre = c.take<uint16_t>(M * Hd); a.hact_hi = c.take<uint16_t>(M * Hd); a.hact_lo = hact_lo;
        if (acts) acts[i] = a;
    }
    if (zoff) *zoff = 0;
    if (zbytes) *zbytes = 0;
    if (bwd) {
        S3dBlockScratch s;
        memset(&s, 0, sizeof(s));
        s.dxn = c.take<float>(M * D * (ring ? 3 : 1)); s.dx_a = c.take<float>(M * D); s.dx_b = c.take<float>(M * D);
        s.dx_a_bf = c.take<uint16_t>(M * D); s.dx_b_bf = c.take<uint16_t>(M * D);
        s.dh = c.take<uint16_t>(M * Hd); s.dqkv = c.take<uint16_t>(M * 3 * D); s.datt = c.take<uint16_t>(M * D);
        s.delta = c.take<float>(BHN);
        s.ln_partial_blocks = ws_ln_partial_blocks((long)M, (int)D);
        s.ln_partial = c.take<float>((size_t)2 * depth * s.ln_partial_blocks * 2 * D);
        if (ring) {
            s.wg_slots = depth < 3 ? depth : 3; s.dgrad_splitk = 3;
            s.wg_ring = c.take<uint16_t>(wg_slot_elems(M, D, Hd) * (size_t)s.wg_slots);
            s.ln_aux = c.take<float>((size_t)depth * 2 * (Hd + 3 * D));
        }
        const size_t z0 = c.off;                     // everything from here on is cleared once by the caller
        if (ring) s.ln_rowstat = c.take<float>(4 * M);
        if (sh.cls_only_block) { s.dx_b_cls = c.take<float>(M * D); s.dx_b_bf_cls = c.take<uint16_t>(M * D); s.datt_cls = c.take<uint16_t>(M * D); }
        if (c.off > z0) {
            if (zoff) *zoff = z0;
            if (zbytes) *zbytes = c.off - z0;
        }
        if (sc) *sc = s;
    }
    return c.off;
}
bool block_ws_shape_ok(const S3dBlockShape* sh, int depth) {
    return sh && depth >= 1 && depth <= 64 && sh->Bb > 0 && sh->N > 0 && sh->D > 0 && sh->H > 0 && sh->hidden > 0 && sh->D % 8 == 0 && sh->hidden % 8 == 0;
}
}  // namespace

// ---------------------------------------------------------------------------------------------- extern "C"
extern "C" {

int s3d_version(void) { return 100; }
const char* s3d_last_error_string(void) { return g_err; }

size_t s3d_sizeof(const char* n) {
#define SZ(T) if (strcmp(n, #T) == 0) return sizeof(T)
    SZ(S3dGemmArgs); SZ(S3dLnArgs); SZ(S3dLnBwdArgs); SZ(S3dAttnArgs); SZ(S3dFoldArgs); SZ(S3dPosGradArgs);
    SZ(S3dHeadArgs); SZ(S3dCeArgs); SZ(S3dHeadLossArgs); SZ(S3dAdamState); SZ(S3dBlockShape); SZ(S3dBlockParams); SZ(S3dBlockGrads);
    SZ(S3dBlockActs); SZ(S3dBlockScratch); SZ(S3dEncShape); SZ(S3dEncParams); SZ(S3dEncGrads); SZ(S3dEncActs); SZ(S3dBnArgs);
    SZ(S3dGroupProjArgs); SZ(S3dAdamFill); SZ(S3dWgradItem); SZ(S3dRowStats); SZ(S3dLnAuxLayer);
#undef SZ
    return 0;
}

int s3d_set_deterministic(int on) { g_deterministic = on ? 1 : 0; return 0; }
int s3d_get_deterministic(void) { return s3d_deterministic() ? 1 : 0; }

int s3d_prof_enable(int on) { s3d_gemm_prof_enable(on != 0); return 0; }
int s3d_prof_collect(double* rows, int cap) { return s3d_gemm_prof_collect(rows, cap); }
int s3d_prof_skip(double key) { s3d_gemm_prof_skip((long long)key); return 0; }
double s3d_prof_skip_get(void) { return (double)s3d_gemm_prof_skip_get(); }
int s3d_prof_event_overhead(s3d_stream_t stream, double* us) {
    S3D_REQUIRE(us != nullptr, "s3d_prof_event_overhead: null result pointer");
    constexpr int R = 33;
    hipEvent_t e[2 * R];
    float t[R];
    for (auto& ev : e) (void)hipEventCreate(&ev);
    for (int i = 0; i < R; ++i) { (void)hipEventRecord(e[2 * i], st(stream)); (void)hipEventRecord(e[2 * i + 1], st(stream)); }
    (void)hipEventSynchronize(e[2 * R - 1]);
    for (int i = 0; i < R; ++i) { t[i] = 0.f; (void)hipEventElapsedTime(&t[i], e[2 * i], e[2 * i + 1]); }
    for (auto& ev : e) (void)hipEventDestroy(ev);
    for (int i = 1; i < R; ++i) { const float v = t[i]; int j = i - 1; while (j >= 0 && t[j] > v) { t[j + 1] = t[j]; --j; } t[j + 1] = v; }
    *us = (double)t[R / 2] * 1e3;
    return 0;
}

int s3d_gemm(int ta, int tb, int split, int epi, const S3dGemmArgs* a, int splitk, s3d_stream_t s) {
    S3D_REQUIRE(a != nullptr, "s3d_gemm: null args");
    return s3d_launch_gemm(ta != 0, tb != 0, split != 0, epi, *a, splitk, st(s));
}
int s3d_gemm_pair(int epi_dgrad, const S3dGemmArgs* dgrad, const S3dGemmArgs* wgrad, s3d_stream_t s) {
    S3D_REQUIRE(dgrad != nullptr && wgrad != nullptr, "s3d_gemm_pair: null args");
    S3D_REQUIRE(dgrad->M > 0 && dgrad->N > 0 && dgrad->K > 0 && wgrad->M > 0 && wgrad->N > 0 && wgrad->K > 0, "s3d_gemm_pair: empty problem");
    S3D_REQUIRE(wgrad->C != nullptr, "s3d_gemm_pair: the wgrad half accumulates into C");
    return s3d_launch_gemm_pair(epi_dgrad, *dgrad, *wgrad, st(s));
}
int s3d_gemm_dgrad_splitk_slices(int K, int want) { return s3d_dgrad_splitk_slices(K, want); }
int s3d_gemm_dgrad_splitk(const S3dGemmArgs* a, int nslice, long slice_stride, s3d_stream_t s) {
    S3D_REQUIRE(a != nullptr, "s3d_gemm_dgrad_splitk: null args");
    return s3d_launch_dgrad_splitk(*a, nslice, slice_stride, st(s));
}
int s3d_gemm_dgrad_dgelu(const S3dGemmArgs* a, const S3dRowStats* stats, s3d_stream_t s) {
    S3D_REQUIRE(a != nullptr, "s3d_gemm_dgrad_dgelu: null args");
    return s3d_launch_dgrad_dgelu(*a, stats, st(s));
}
int s3d_gemm_dgrad_lnbwd(const S3dGemmArgs* a, const S3dLnBwdArgs* ln, const S3dRowStats* stats, s3d_stream_t s) {
    S3D_REQUIRE(a != nullptr && ln != nullptr, "s3d_gemm_dgrad_lnbwd: null args");
    return s3d_launch_dgrad_lnbwd(*a, *ln, stats, st(s));
}
int s3d_gemm_dgrad_lnrows(const S3dGemmArgs* a, const S3dLnBwdArgs* ln, s3d_stream_t s) {
    S3D_REQUIRE(a != nullptr && ln != nullptr, "s3d_gemm_dgrad_lnrows: null args");
    return s3d_launch_dgrad_lnrows(*a, *ln, st(s));
}
int s3d_ln_aux(const S3dLnAuxLayer* layers, int n, int D, s3d_stream_t s) { return s3d_launch_ln_aux(layers, n, D, st(s)); }
int s3d_gemm_wgrad_group(const S3dWgradItem* items, int n, int K, float alpha, int accumulate, s3d_stream_t s) {
    return s3d_launch_wgrad_group(items, n, K, alpha, accumulate, st(s));
}
size_t s3d_block_wgrad_slot_bytes(const S3dBlockShape* sh) {
    if (sh == nullptr || sh->Bb <= 0 || sh->N <= 0 || sh->D <= 0 || sh->hidden <= 0) { s3d_set_error("s3d_block_wgrad_slot_bytes: bad shape"); return 0; }
    return wg_slot_elems((size_t)sh->Bb * sh->N, sh->D, sh->hidden) * sizeof(bf16_t);
}
int s3d_gemm_pair3(int epi_dgrad, const S3dGemmArgs* dgrad, const S3dGemmArgs* wgrad, const S3dGemmArgs* wgrad2, s3d_stream_t s) {
    S3D_REQUIRE(dgrad && wgrad && wgrad2, "s3d_gemm_pair3: null args");
    S3D_REQUIRE(wgrad->C != nullptr && wgrad2->C != nullptr && wgrad2->M > 0 && wgrad2->N > 0 && wgrad2->K > 0, "s3d_gemm_pair3: both wgrads accumulate into C");
    return s3d_launch_gemm_pair(epi_dgrad, *dgrad, *wgrad, st(s), nullptr, wgrad2);
}
int s3d_cov_enable(int on) {
    if (on) g_cov.clear();
    g_s3d_cov_on = on != 0;
    return 0;
}
long s3d_cov_collect(char* buf, long cap) {
    std::string out;
    for (const auto& kv : g_cov) {
        char line[160];
        snprintf(line, sizeof(line), "%s:%lld:%ld\n", kv.first.first.c_str(), kv.first.second, kv.second);
        out += line;
    }
    if (buf != nullptr && cap > 0) {
        const long n = (long)out.size() < cap - 1 ? (long)out.size() : cap - 1;
        memcpy(buf, out.data(), (size_t)n);
        buf[n] = 0;
    }
    return (long)out.size() + 1;
}
int s3d_gemm_col_sums_ok(int split, int M, int N) {
    static const bool forced = s3d_tune_int("S3D_GEMM_NT_TILE") >= 0;
    static const bool off = s3d_tune_int("S3D_GEMM_COL_SUMS") == 0;
    return (!forced && !off && M > 0 && N > 0 && (N & 7) == 0 && s3d_gemm_pick_tile(M, N, 1, split != 0) == 2) ? 1 : 0;
}
int s3d_gemm_ln_fusable(int split, const S3dGemmArgs* a) { return (a != nullptr && s3d_gemm_ln_fusable(split != 0, *a)) ? 1 : 0; }
int s3d_layernorm_fwd(const S3dLnArgs* a, s3d_stream_t s) {
    S3D_REQUIRE(a != nullptr, "s3d_layernorm_fwd: null args");
    return s3d_launch_ln_fwd(*a, st(s));
}
int s3d_layernorm_grad_reduce(const float* const* partial, float* const* dgamma, float* const* dbeta, int n_ln, int nblk, int D,
                              s3d_stream_t s) {
    S3D_REQUIRE(partial && dgamma && dbeta, "s3d_layernorm_grad_reduce: null args");
    return s3d_launch_ln_grad_reduce(partial, dgamma, dbeta, n_ln, nblk, D, st(s));
}
int s3d_layernorm_bwd(const S3dLnBwdArgs* a, s3d_stream_t s) {
    S3D_REQUIRE(a != nullptr, "s3d_layernorm_bwd: null args");
    return s3d_launch_ln_bwd(*a, st(s));
}
int s3d_attention_fwd(const S3dAttnArgs* a, int split, s3d_stream_t s) {
    S3D_REQUIRE(a != nullptr, "s3d_attention_fwd: null args");
    return s3d_launch_attention_fwd(*a, split != 0, st(s));
}
int s3d_attention_bwd(const S3dAttnArgs* a, s3d_stream_t s) {
    S3D_REQUIRE(a != nullptr, "s3d_attention_bwd: null args");
    return s3d_launch_attention_bwd(*a, st(s));
}
int s3d_voxel_fold(const S3dFoldArgs* a, s3d_stream_t s) {
    S3D_REQUIRE(a != nullptr, "s3d_voxel_fold: null args");
    return s3d_launch_fold(*a, st(s));
}
int s3d_token_grads(const S3dPosGradArgs* a, s3d_stream_t s) {
    S3D_REQUIRE(a != nullptr, "s3d_token_grads: null args");
    return s3d_launch_posgrad(*a, st(s));
}
int s3d_split_bf16(const float* src, uint16_t* hi, uint16_t* lo, long rows, long cols, long ld, s3d_stream_t s) {
    return s3d_launch_split(src, hi, lo, rows, cols, ld, st(s));
}
int s3d_head_fwd(const S3dHeadArgs* a, s3d_stream_t s) {
    S3D_REQUIRE(a != nullptr, "s3d_head_fwd: null args");
    return s3d_launch_head_fwd(*a, st(s));
}
int s3d_head_bwd(const S3dHeadArgs* a, s3d_stream_t s) {
    S3D_REQUIRE(a != nullptr, "s3d_head_bwd: null args");
    return s3d_launch_head_bwd(*a, st(s));
}
int s3d_head_loss_fused(const S3dHeadLossArgs* a, s3d_stream_t s) {
    S3D_REQUIRE(a != nullptr, "s3d_head_loss_fused: null args");
    return s3d_launch_head_loss(*a, st(s));
}
int s3d_l2norm_rows_fwd(const float* x, long ldx, long rows, int D, float* inv_norm, uint16_t* hi, uint16_t* lo, long ldo, s3d_stream_t s) {
    return s3d_launch_l2norm_rows_fwd(x, ldx, rows, D, inv_norm, hi, lo, ldo, st(s));
}
int s3d_l2norm_rows_bwd(const float* dxn, long lddxn, const float* x, long ldx, const float* inv_norm, long rows, int D, float* dx, long lddx,
                        s3d_stream_t s) {
    return s3d_launch_l2norm_rows_bwd(dxn, lddxn, x, ldx, inv_norm, rows, D, dx, lddx, st(s));
}
int s3d_am_weight_fwd(const float* W, int D, int C, float scale, float* Wl, int ldw, float* inv_w, s3d_stream_t s) {
    return s3d_launch_am_weight_fwd(W, D, C, scale, Wl, ldw, inv_w, st(s));
}
int s3d_am_weight_bwd(const float* dWl, int ldw, const float* W, const float* inv_w, int D, int C, float scale, float* dW, s3d_stream_t s) {
    return s3d_launch_am_weight_bwd(dWl, ldw, W, inv_w, D, C, scale, dW, st(s));
}
int s3d_cross_entropy(const S3dCeArgs* a, s3d_stream_t s) {
    S3D_REQUIRE(a != nullptr, "s3d_cross_entropy: null args");
    return s3d_launch_ce(*a, st(s));
}
int s3d_image_patchify(const float* img, uint16_t* a_hi, uint16_t* a_lo, long lda, int B, int C, int H, int W, int p, s3d_stream_t s) {
    S3D_REQUIRE(img && a_hi, "s3d_image_patchify: null args");
    return s3d_launch_patchify(img, a_hi, a_lo, lda, B, C, H, W, p, st(s));
}
int s3d_adam_step(float* p, float* g, float* m, float* v, uint16_t* hi, uint16_t* lo, long n, S3dAdamState* state,
                  int zero_grad, s3d_stream_t s) {
    return s3d_launch_adam(p, g, m, v, hi, lo, n, state, zero_grad, nullptr, st(s));
}
int s3d_adam_step_wire(float* p, float* g, const uint16_t* g_wire, float* m, float* v, uint16_t* hi, uint16_t* lo, long n,
                       S3dAdamState* state, int zero_grad, s3d_stream_t s) {
    S3D_REQUIRE(g_wire != nullptr, "s3d_adam_step_wire: the bf16 gradient buffer is required");
    return s3d_launch_adam(p, g, m, v, hi, lo, n, state, zero_grad, g_wire, st(s));
}
int s3d_adam_begin(S3dAdamState* state, s3d_stream_t s) {
    S3D_REQUIRE(state != nullptr, "s3d_adam_begin: null state");
    return s3d_launch_adam_begin(state, st(s));
}
int s3d_adam_apply(float* p, float* g, const uint16_t* g_wire, float* m, float* v, uint16_t* hi, uint16_t* lo, long n,
                   const S3dAdamState* state, int zero_grad, int max_workgroups, s3d_stream_t s) {
    S3D_REQUIRE(p && g && m && v && hi && lo && state, "s3d_adam_apply: null pointer");
    return s3d_launch_adam_apply(p, g, m, v, hi, lo, n, state, zero_grad, g_wire, max_workgroups, st(s));
}
int s3d_adam_apply_ranges(float* p, float* g, float* m, float* v, uint16_t* hi, uint16_t* lo, const long* ranges, int n,
                          const S3dAdamState* state, int zero_grad, s3d_stream_t s) {
    S3D_REQUIRE(p && g && m && v && state && (ranges || n == 0), "s3d_adam_apply_ranges: null pointer");
    return s3d_launch_adam_ranges(p, g, m, v, hi, lo, ranges, n, state, zero_grad, st(s));
}
int s3d_pack_bf16(const float* src, uint16_t* dst, long n, s3d_stream_t s) { return s3d_launch_pack_bf16(src, dst, n, st(s)); }

// ---- events recorded INSIDE a stream capture as external event-record nodes (see s3d_hip.h)
int s3d_event_create(void** out) {
    S3D_REQUIRE(out != nullptr, "s3d_event_create: null result pointer");
    hipEvent_t e = nullptr;
    const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    S3D_REQUIRE(rc == hipSuccess, "s3d_event_create: %s", hipGetErrorString(rc));
    *out = e;
    return 0;
}
int s3d_event_destroy(void* ev) {
    if (ev) (void)hipEventDestroy(static_cast<hipEvent_t>(ev));
    return 0;
}
// A marker in a stream capture: a kernel node that s3d_graph_events_at_markers replaces by an event-record node.  (Recording with
// hipEventRecordWithFlags(.., hipEventRecordExternal) on the capturing stream is refused -- "invalid argument" -- by the HIP runtime
// PyTorch 2.10+rocm7.0 bundles; editing the captured graph works.)
__global__ void s3d_marker_kernel(int) {}
int s3d_graph_marker(int id, s3d_stream_t s) {
    hipLaunchKernelGGL(s3d_marker_kernel, dim3(1), dim3(1), 0, st(s), id);
    S3D_CHECK_LAUNCH("graph_marker");
    return 0;
}
// Measurement aid (parallel.py: the stand-in collectives of BucketedGradReducer / ShardedDataParallelTrainer): copies nbytes with a few workgroups
// paced to `gbps` GB/s against the 100 MHz wall clock -- the footprint (a few CUs) and duration a ring collective of the same bytes has on RCCL's
// stream, for measuring what a live side branch costs a captured step graph on one GPU.  Round 6: four 16-byte loads in flight per thread
// (~25 GB/s per workgroup instead of ~7), so that 286 GB/s takes 16 workgroups, not 58 -- a ring kernel's channel count, not a quarter of the chip.
__global__ __launch_bounds__(256) void s3d_paced_copy_kernel(u32x4* dst, const u32x4* src, long n16, float bytes_per_tick) {
    const long per = ((n16 + gridDim.x - 1) / gridDim.x + 1023) / 1024 * 1024, b = (long)blockIdx.x * per, e = b + per < n16 ? b + per : n16;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    long chunk = 0;
    for (long i = b; i < e; i += 1024, ++chunk) {
        u32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const long k = i + j * 256 + threadIdx.x; v[j] = k < e ? src[k] : u32x4{0u, 0u, 0u, 0u}; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const long k = i + j * 256 + threadIdx.x; if (k < e) dst[k] = v[j]; }
        const float due = (float)((chunk + 1) * 16384) / bytes_per_tick;
        while ((float)(__builtin_amdgcn_s_memrealtime() - t0) < due) __builtin_amdgcn_s_sleep(2);
    }
}
int s3d_debug_knob(int id, int value) {
    S3D_REQUIRE(id >= 0 && id < 16, "s3d_debug_knob: id 0 .. 15");
    g_knobs[id] = value;
    return 0;
}
int s3d_debug_paced_copy(void* dst, const void* src, long nbytes, float gbps, s3d_stream_t s) {
    S3D_REQUIRE(dst && src && nbytes >= 16 && (nbytes & 15) == 0 && gbps > 0.f, "s3d_debug_paced_copy: 16-byte multiples, a positive rate");
    int wgs = (int)(gbps / 18.0f) + 1;
    wgs = wgs < 4 ? 4 : wgs > 64 ? 64 : wgs;
    const float bytes_per_tick = gbps * 10.0f / wgs;                    // GB/s = bytes per ns; one tick of the 100 MHz clock = 10 ns
    hipLaunchKernelGGL(s3d_paced_copy_kernel, dim3(wgs), dim3(256), 0, st(s), static_cast<u32x4*>(dst), static_cast<const u32x4*>(src), nbytes / 16,
                       bytes_per_tick);
    S3D_CHECK_LAUNCH("paced_copy");
    return 0;
}
int s3d_graph_events_at_markers(void* graph, void* const* events, int n) {
    S3D_REQUIRE(graph && events && n >= 1 && n <= 64, "s3d_graph_events_at_markers: graph, 1..64 events");
    hipGraph_t g = static_cast<hipGraph_t>(graph);
#define S3D_HIP(expr)                                                                                          \
    do {                                                                                                       \
        const hipError_t e__ = (expr);                                                                         \
        if (e__ != hipSuccess) {                                                                               \
            s3d_set_error("s3d_graph_events_at_markers: %s failed: %s", #expr, hipGetErrorString(e__));        \
            return 3;                                                                                          \
        }                                                                                                      \
    } while (0)
    size_t count = 0;
    S3D_HIP(hipGraphGetNodes(g, nullptr, &count));
    std::vector<hipGraphNode_t> nodes(count);
    S3D_HIP(hipGraphGetNodes(g, nodes.data(), &count));
    std::vector<hipGraphNode_t> marker(n, nullptr);
    for (size_t i = 0; i < count; ++i) {
        hipGraphNodeType ty;
        S3D_HIP(hipGraphNodeGetType(nodes[i], &ty));
        if (ty != hipGraphNodeTypeKernel) continue;
        hipKernelNodeParams kp;
        memset(&kp, 0, sizeof(kp));
        S3D_HIP(hipGraphKernelNodeGetParams(nodes[i], &kp));
        if (kp.func != reinterpret_cast<void*>(s3d_marker_kernel) || kp.kernelParams == nullptr) continue;
        const int id = *static_cast<const int*>(kp.kernelParams[0]);
        S3D_REQUIRE(id >= 0 && id < n && marker[id] == nullptr, "s3d_graph_events_at_markers: unexpected marker id %d", id);
        marker[id] = nodes[i];
    }
    for (int k = 0; k < n; ++k) {
        S3D_REQUIRE(marker[k] != nullptr, "s3d_graph_events_at_markers: marker %d not found in the graph", k);
        size_t nd = 0, nx = 0;
        S3D_HIP(hipGraphNodeGetDependencies(marker[k], nullptr, &nd));
        S3D_HIP(hipGraphNodeGetDependentNodes(marker[k], nullptr, &nx));
        std::vector<hipGraphNode_t> deps(nd ? nd : 1), nexts(nx ? nx : 1);
        if (nd) S3D_HIP(hipGraphNodeGetDependencies(marker[k], deps.data(), &nd));
        if (nx) S3D_HIP(hipGraphNodeGetDependentNodes(marker[k], nexts.data(), &nx));
        hipGraphNode_t rec = nullptr;              // in line (prev -> record -> next): the graph stays one chain, no fork / join
        S3D_HIP(hipGraphAddEventRecordNode(&rec, g, nd ? deps.data() : nullptr, nd, static_cast<hipEvent_t>(events[k])));
        for (size_t j = 0; j < nx; ++j) S3D_HIP(hipGraphAddDependencies(g, &rec, &nexts[j], 1));
        S3D_HIP(hipGraphDestroyNode(marker[k]));
    }
#undef S3D_HIP
    return 0;
}
int s3d_stream_wait_event(s3d_stream_t s, void* ev) {
    S3D_REQUIRE(ev != nullptr, "s3d_stream_wait_event: null event");
    const hipError_t rc = hipStreamWaitEvent(st(s), static_cast<hipEvent_t>(ev), 0);
    S3D_REQUIRE(rc == hipSuccess, "s3d_stream_wait_event: %s", hipGetErrorString(rc));
    return 0;
}

size_t s3d_block_workspace_bytes(const S3dBlockShape* sh, int depth, int with_backward) {
    if (!block_ws_shape_ok(sh, depth)) { s3d_set_error("s3d_block_workspace_bytes: bad shape / depth"); return 0; }
    return block_ws_layout(*sh, depth, with_backward, nullptr, nullptr, nullptr, nullptr, nullptr);
}
int s3d_block_workspace_carve(const S3dBlockShape* sh, int depth, int with_backward, void* base, size_t bytes, S3dBlockActs* acts,
                              S3dBlockScratch* scratch, size_t* zero_offset, size_t* zero_bytes) {
    S3D_REQUIRE(block_ws_shape_ok(sh, depth), "s3d_block_workspace_carve: bad shape / depth");
    S3D_REQUIRE(base != nullptr && acts != nullptr && (!with_backward || scratch != nullptr), "s3d_block_workspace_carve: base, acts (and scratch with with_backward) required");
    S3D_REQUIRE(((uintptr_t)base & 255) == 0, "s3d_block_workspace_carve: base must be 256-byte aligned");
    const size_t need = block_ws_layout(*sh, depth, with_backward, nullptr, nullptr, nullptr, nullptr, nullptr);
    S3D_REQUIRE(bytes >= need, "s3d_block_workspace_carve: %zu bytes given, %zu needed", bytes, need);
    block_ws_layout(*sh, depth, with_backward, base, acts, scratch, zero_offset, zero_bytes);
    return 0;
}
int s3d_block_fwd(const S3dBlockShape* sh, const S3dBlockParams* p, const S3dBlockActs* a, s3d_stream_t s) {
    S3D_REQUIRE(sh && p && a, "s3d_block_fwd: null args");
    return block_fwd(*sh, *p, *a, st(s));
}
int s3d_block_bwd(const S3dBlockShape* sh, const S3dBlockParams* p, const S3dBlockGrads* g, const S3dBlockActs* a,
                  const S3dBlockScratch* w, s3d_stream_t s) {
    S3D_REQUIRE(sh && p && g && a && w, "s3d_block_bwd: null args");
    LnPartials lp;
    const bool partial = w->ln_partial != nullptr && w->ln_partial_blocks > 0;
    S3D_TRY(block_bwd(*sh, *p, *g, *a, *w, st(s), partial ? &lp : nullptr));
    return lp.flush(*w, sh->D, st(s));
}
int s3d_blocks_fwd(const S3dBlockShape* sh, const S3dBlockParams* p, const S3dBlockActs* a, int depth, s3d_stream_t s) {
    S3D_REQUIRE(sh && p && a, "s3d_blocks_fwd: null args");
    bool ln1_done = false;
    for (int i = 0; i < depth; ++i) {
        const bool has_next = i + 1 < depth && a[i + 1].x_in == a[i].x_out;     // the next block normalises exactly this block's output
        bool next_done = false;
        S3D_TRY(block_fwd(*sh, p[i], a[i], st(s), ln1_done, has_next ? &p[i + 1] : nullptr, has_next ? &a[i + 1] : nullptr, &next_done,
                          sh->cls_only_block == i + 1));
        ln1_done = next_done;
    }
    return 0;
}
int s3d_blocks_fwd_range(const S3dBlockShape* sh, const S3dBlockParams* p, const S3dBlockActs* a, int depth, int first, int last, s3d_stream_t s) {
    S3D_REQUIRE(sh && p && a && first >= 0 && first <= last && last < depth, "s3d_blocks_fwd_range: null args or not 0 <= first <= last < depth");
    bool ln1_done = false;
    for (int i = first; i <= last; ++i) {
        const bool has_next = i < last && a[i + 1].x_in == a[i].x_out;     // the range end never touches block last + 1's parameters
        bool next_done = false;
        S3D_TRY(block_fwd(*sh, p[i], a[i], st(s), ln1_done, has_next ? &p[i + 1] : nullptr, has_next ? &a[i + 1] : nullptr, &next_done,
                          sh->cls_only_block == i + 1));
        ln1_done = next_done;
    }
    return 0;
}
int s3d_blocks_bwd(const S3dBlockShape* sh, const S3dBlockParams* p, const S3dBlockGrads* g, const S3dBlockActs* a,
                   const S3dBlockScratch* w, int first, int last, s3d_stream_t s) {
    S3D_REQUIRE(sh && p && g && a && w, "s3d_blocks_bwd: null args");
    LnPartials lp;
    const bool partial = w->ln_partial != nullptr && w->ln_partial_blocks > 0;
    // optimizer shares riding on the launches (S3dAdamFill): while block i runs, the GEMM parameters of block i + 1 are updated
    const S3dAdamFill* af = (w->dx_a_lo == nullptr) ? w->adam_fill : nullptr;
    AdamFillQueue fq;
    int n_filled = 0;
    if (af) {
        S3D_REQUIRE(af->p && af->g && af->m && af->v && af->state && af->filled && af->n_filled && af->filled_cap >= 2,
                    "s3d_blocks_bwd: S3dAdamFill needs the arena base pointers, the optimizer state and the `filled` report array");
        fq.base = af;
        *af->n_filled = 0;
    }
    auto drain = [&]() -> int {               // shares no launch could carry (a launch path without filler support): a plain update
        while (af && !fq.empty()) {
            fq.share16 = 16;
            const AdamFill f = fq.take(256);
            if (f.n4 == 0) break;
            S3D_TRY(s3d_launch_adam_apply(f.p, f.g, f.m, f.v, f.hi, f.lo, f.n4 * 4, af->state, af->zero_grad, nullptr, 0, st(s)));
        }
        return 0;
    };
    WgradDefer wd;
    const bool chain = wgrad_chain_ok(*sh, *w);
    if (chain) {
        wd.ring = w->wg_ring; wd.slots = w->wg_slots < 12 ? w->wg_slots : 12; wd.splitk = w->dgrad_splitk > 0 ? w->dgrad_splitk : 1;
        wd.M = (size_t)sh->Bb * sh->N; wd.D = sh->D; wd.Hd = sh->hidden;
        wd.dxa_cur = w->dx_a_bf;
        wd.accumulate = w->wg_overwrite ? 0 : 1;
        wd.af = af; wd.n_filled = &n_filled;
        if (w->ln_aux != nullptr && w->ln_rowstat != nullptr && !s3d_deterministic()) {
            if (!w->ln_aux_valid) S3D_TRY(blocks_ln_aux(*sh, p, *w, first, last, st(s)));
            wd.aux = w->ln_aux;
        }
    }
    for (int i = first; i >= last; --i) {
        const bool cls_only = sh->cls_only_block == i + 1;
        if (cls_only) S3D_REQUIRE(w->dx_b_cls && w->dx_b_bf_cls && w->datt_cls, "s3d_blocks_bwd: cls_only_block needs the *_cls scratch buffers");
        const bool on_chain = chain && !cls_only;
        wd.block = i;
        S3D_TRY(block_bwd(*sh, p[i], g[i], a[i], *w, st(s), partial ? &lp : nullptr, cls_only, (af && !chain) ? &fq : nullptr, chain ? &wd : nullptr, i == last));
        if (lp.n + 2 > 64) S3D_TRY(lp.flush(*w, sh->D, st(s)));
        if (af && chain) {
            // dgrad chain: a block's GEMM gradients are final once the grouped launch that holds its wgrads has run (WgradDefer::flush moves
            // its ranges from `pend` to `ready`); a block off the chain (class rows only) is final right here
            const S3dBlockGrads& b = g[i];
            const long D = sh->D, Hd = sh->hidden;
            struct T { float* q; long n; } t[8] = {{b.qkv_w, 3 * D * D}, {b.qkv_b, 3 * D}, {b.proj_w, D * D}, {b.proj_b, D},
                                                   {b.fc1_w, Hd * D}, {b.fc1_b, Hd}, {b.fc2_w, D * Hd}, {b.fc2_b, D}};
            for (int x = 1; x < 8; ++x) for (int y = x; y > 0 && t[y].q < t[y - 1].q; --y) { const T tmp = t[y]; t[y] = t[y - 1]; t[y - 1] = tmp; }
            long off = -1, len = 0;
            bool whole = true;
            for (int x = 0; x < 8; ++x) {
                const long o = t[x].q - af->g;
                if (t[x].q == nullptr || o < 0 || (o & 3) != 0 || (t[x].n & 3) != 0 || (off >= 0 && o != off + len)) { whole = false; break; }
                if (off < 0) off = o;
                len += t[x].n;
            }
            if (whole && off >= 0) {                   // (the arena keeps a block's GEMM parameters contiguous; anything else is left to the caller)
                // NOTE: a chain block's ranges were registered BEFORE its flush when the flush happened inside block_bwd (group full / end of
                // call): those gradients are final as well -- the flush has been enqueued -- so they go straight to `ready`
                if (on_chain && wd.pending_blocks > 0) wd.defer_range(off, len);
                else wd.ready_range(off, len);
            }
        } else if (af) {
            S3D_TRY(drain());
            fq.reset();
            if (i > last) {
                // block i's GEMM parameters are final now: qkv / proj / fc1 / fc2 weights and biases, merged into the contiguous
                // ranges the caller's arena gives them (the LayerNorm gradients wait for the partial-sum reduction at the very end)
                const S3dBlockGrads& b = g[i];
                const long D = sh->D, Hd = sh->hidden;
                struct T { float* q; long n; } t[8] = {{b.qkv_w, 3 * D * D}, {b.qkv_b, 3 * D}, {b.proj_w, D * D}, {b.proj_b, D},
                                                       {b.fc1_w, Hd * D}, {b.fc1_b, Hd}, {b.fc2_w, D * Hd}, {b.fc2_b, D}};
                for (int x = 1; x < 8; ++x) for (int y = x; y > 0 && t[y].q < t[y - 1].q; --y) { const T tmp = t[y]; t[y] = t[y - 1]; t[y - 1] = tmp; }
                long off = -1, len = 0;
                auto flush = [&]() {
                    if (off >= 0 && len >= 4 && n_filled * 2 + 2 <= af->filled_cap && fq.nseg < 4) {
                        const long n = len / 4 * 4;
                        fq.push(off, n);
                        af->filled[2 * n_filled] = off; af->filled[2 * n_filled + 1] = n;
                        ++n_filled;
                    }
                };
                for (int x = 0; x < 8; ++x) {
                    const long o = t[x].q - af->g;
                    if (t[x].q == nullptr || o < 0 || (o & 3) != 0 || (t[x].n & 3) != 0) { flush(); off = -1; len = 0; continue; }   // not ours to touch
                    if (off >= 0 && o == off + len) len += t[x].n;              // adjacent in the arena: one range
                    else { flush(); off = o; len = t[x].n; }
                }
                flush();
            }
        }
    }
    if (af) { S3D_TRY(drain()); *af->n_filled = n_filled; }
    return lp.flush(*w, sh->D, st(s));
}

int s3d_blocks_ln_aux(const S3dBlockShape* sh, const S3dBlockParams* p, const S3dBlockScratch* w, int first, int last, s3d_stream_t s) {
    S3D_REQUIRE(sh && p && w && w->ln_aux && first >= last && last >= 0, "s3d_blocks_ln_aux: shape, params, scratch->ln_aux, first >= last >= 0");
    return blocks_ln_aux(*sh, p, *w, first, last, st(s));
}

int s3d_encoder_layer_fwd(const S3dEncShape* sh, const S3dEncParams* p, const S3dEncActs* a, s3d_stream_t s) {
    S3D_REQUIRE(sh && p && a, "s3d_encoder_layer_fwd: null args");
    return enc_fwd(*sh, *p, *a, st(s));
}
int s3d_encoder_layer_bwd(const S3dEncShape* sh, const S3dEncParams* p, const S3dEncGrads* g, const S3dEncActs* a,
                          const S3dBlockScratch* w, s3d_stream_t s) {
    S3D_REQUIRE(sh && p && g && a && w, "s3d_encoder_layer_bwd: null args");
    return enc_bwd(*sh, *p, *g, *a, *w, st(s));
}
int s3d_assemble_tokens(const float* src, const float* cls, const float* pos, float* out, long B, int n, int D, s3d_stream_t s) {
    return s3d_launch_assemble(src, cls, pos, out, B, n, D, st(s));
}
int s3d_assemble_tokens_bwd(const float* dout, float* dsrc, long B, int n, int D, s3d_stream_t s) {
    return s3d_launch_assemble_bwd(dout, dsrc, B, n, D, st(s));
}

int s3d_fps(const float* xyz, long xyz_ld, const long long* start, int B, int N, int npoint, int* out_idx, float* new_xyz, s3d_stream_t s) {
    return s3d_launch_fps(xyz, xyz_ld, start, B, N, npoint, out_idx, new_xyz, st(s));
}
int s3d_knn(const float* q, const float* r, int B, int S, int N, int K, int* out_idx, float* out_w, s3d_stream_t s) {
    return s3d_launch_knn(q, r, B, S, N, K, out_idx, out_w, st(s));
}
int s3d_group_gather(const float* xyz, const float* new_xyz, const float* feats, const int* idx, int B, int N, int S, int K, int C,
                     uint16_t* a_hi, uint16_t* a_lo, int lda, s3d_stream_t s) {
    return s3d_launch_group_gather(xyz, new_xyz, feats, idx, B, N, S, K, C, a_hi, a_lo, lda, st(s));
}
int s3d_group_scatter(const float* dA, int ldd, const int* idx, int B, int N, int S, int K, int C, float* dfeats, s3d_stream_t s) {
    return s3d_launch_group_scatter(dA, ldd, idx, B, N, S, K, C, dfeats, st(s));
}
int s3d_group_project_fwd(const S3dGroupProjArgs* a, s3d_stream_t s) {
    S3D_REQUIRE(a != nullptr, "s3d_group_project_fwd: null args");
    return s3d_launch_group_project_fwd(*a, st(s));
}
int s3d_group_project_bwd(const S3dGroupProjArgs* a, s3d_stream_t s) {
    S3D_REQUIRE(a != nullptr, "s3d_group_project_bwd: null args");
    return s3d_launch_group_project_bwd(*a, st(s));
}
int s3d_neighbor_csr(const int* idx, int B, int N, int S, int K, int* inv_off, int* inv_rows, s3d_stream_t s) {
    S3D_REQUIRE(idx && inv_off && inv_rows, "s3d_neighbor_csr: null pointer");
    return s3d_launch_neighbor_csr(idx, B, N, S, K, inv_off, inv_rows, st(s));
}
int s3d_batchnorm_fwd(const S3dBnArgs* a, s3d_stream_t s) {
    S3D_REQUIRE(a != nullptr, "s3d_batchnorm_fwd: null args");
    return s3d_launch_bn_fwd(*a, st(s));
}
int s3d_batchnorm_bwd(const S3dBnArgs* a, s3d_stream_t s) {
    S3D_REQUIRE(a != nullptr, "s3d_batchnorm_bwd: null args");
    return s3d_launch_bn_bwd(*a, st(s));
}
int s3d_interp3(const float* f1, int S, const float* f2, const int* idx, const float* w, int B, int N, int C, float* out, s3d_stream_t s) {
    return s3d_launch_interp3(f1, S, f2, idx, w, B, N, C, out, st(s));
}
int s3d_interp3_bwd(const float* dout, const int* idx, const float* w, int B, int S, int N, int C, float* df1, s3d_stream_t s) {
    return s3d_launch_interp3_bwd(dout, idx, w, B, S, N, C, df1, st(s));
}
int s3d_mean_points(const float* x, int B, int N, int C, float* out, s3d_stream_t s) { return s3d_launch_mean_points(x, B, N, C, out, st(s)); }
int s3d_bcast_rows(const float* x, int N, int C, long rows, float scale, float* y, s3d_stream_t s) {
    return s3d_launch_bcast_rows(x, N, C, rows, scale, y, st(s));
}
int s3d_pack_rows(const float* x, int C, int ldx, long rows, uint16_t* hi, uint16_t* lo, int ldo, s3d_stream_t s) {
    return s3d_launch_pack_rows(x, C, ldx, rows, hi, lo, ldo, st(s));
}
int s3d_add_inplace(float* a, const float* b, long n, s3d_stream_t s) { return s3d_launch_add_inplace(a, b, n, st(s)); }
int s3d_sgd_step(float* p, float* g, float* buf, uint16_t* hi, uint16_t* lo, long n, float lr, float momentum, float grad_scale,
                 int* step_counter, s3d_stream_t s) {
    return s3d_launch_sgd(p, g, buf, hi, lo, n, lr, momentum, grad_scale, step_counter, nullptr, st(s));
}
int s3d_sgd_step_dev(float* p, float* g, float* buf, uint16_t* hi, uint16_t* lo, long n, const float* hyper, int* step_counter,
                     s3d_stream_t s) {
    S3D_REQUIRE(hyper != nullptr, "s3d_sgd_step_dev: hyper = device float[3] {lr, momentum, grad_scale} required");
    return s3d_launch_sgd(p, g, buf, hi, lo, n, 0.f, 0.f, 1.f, step_counter, hyper, st(s));
}

int s3d_cls_eval(const float* logits, int ld, const long long* target, long rows, int C, int* pred, long long* counts, s3d_stream_t s) {
    return s3d_launch_cls_eval(logits, ld, target, rows, C, pred, counts, st(s));
}
int s3d_partseg_eval(const float* logits, int ld, const long long* target, int B, int N, int num_part, const int* part_range, int* pred,
                     double* shape_iou, int* shape_first, long long* counts, s3d_stream_t s) {
    S3D_REQUIRE(num_part > 0 && part_range != nullptr, "s3d_partseg_eval: part table required");
    return s3d_launch_partseg_eval(logits, ld, target, B, N, num_part, part_range, pred, shape_iou, shape_first, counts, st(s));
}
int s3d_unpack_voxels(const unsigned int* bits, float* out, long nwords, s3d_stream_t s) {
    return s3d_launch_unpack_bits(bits, out, nwords, st(s));
}

}  // extern "C"
