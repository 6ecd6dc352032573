// Training step of the ACT policy on MI355X: forward with saved activations, L1+KL loss, full backward,
// fused AdamW.  Reference semantics: ACTPolicy.__call__ training branch (policy.py:288-320), DETRVAE.forward with
// actions (detr_vae.py:107-161, 163-254), torch autograd, torch.optim.AdamW with the two parameter groups of
// detr/main.py:102-110.  Every contraction (forward, data gradient, weight gradient, attention products) runs on
// the fp32 MFMA kernel family of gemm.hip; the remaining pieces are the kernels of bwd.hip / misc.hip.
//
// Quirks preserved: only decoder layer 0 reaches the loss; layers 1.. get exactly-zero gradients and
// therefore only AdamW's decoupled weight decay; is_pad_head gets no gradient at all (skipped by the optimizer, like
// a parameter whose .grad is None); the decoder layer-0 self-attention on tgt = 0 reduces to out_proj(b_v) + b_o, so
// its q/k projections and the v weight receive exactly-zero gradients while b_v, out_proj and norm1 do not.
#include "engine.h"
#include "dropout.h"

#include <cmath>
#include <cstring>

namespace {

#define CHK(expr)                                                                               \
    do {                                                                                        \
        int _rc = (expr);                                                                       \
        if (_rc != 0) {                                                                         \
            if (ctx->err.empty()) ctx->err = std::string("failed: ") + #expr;                   \
            return _rc < -5 ? ACTMI_E_LAUNCH : _rc;                                             \
        }                                                                                       \
    } while (0)

#define HIPCHK(expr)                                                                            \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(_e);                       \
            return ACTMI_E_LAUNCH;                                                              \
        }                                                                                       \
    } while (0)

int talloc(actmi_ctx* ctx, float** p, int64_t n) {
    void* q = nullptr;
    if (hipMalloc(&q, (size_t)(n > 0 ? n : 1) * sizeof(float)) != hipSuccess) {
        ctx->err = "hipMalloc failed (training buffers)";
        return ACTMI_E_NOMEM;
    }
    ctx->allocs.push_back(q);
    *p = reinterpret_cast<float*>(q);
    return 0;
}

GemmArgs G0() {
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.groups = 1;
    return a;
}

// every GEMM of the training step: forward operand forms go through ctx_gemm (handle precision + pre-split weights where
// B is a parameter), the backward forms take the handle precision directly
int tgemm(actmi_ctx* ctx, GemmArgs a, hipStream_t st) {
    if (a.ta == 0 && a.tb == 0 && a.mode != 2) return ctx_gemm(ctx, a, st);
    a.prec = ctx->prec_override ? ctx->prec_override : ctx->gemm_prec;
    TrainState* T = ctx->train;
    if (a.splitk > 1 && a.split_stride == 0 && T && T->det_ws && !a.rowmap && !a.C2 && !a.mask && a.groups_inner == 0) {
        // C += A B with the contraction split over the grid.  The atomic form (every split adds into C) is not run-to-run
        // repeatable; here every split stores a plain slice and a second kernel adds the slices to C in split order.
        const int groups = a.groups > 0 ? a.groups : 1;
        const int nk = (a.K + 31) / 32;
        int S = a.splitk;
        while (S >= 2 && (S - 1) * ((nk + S - 1) / S) >= nk) --S;              // every split must own a K tile
        const int64_t slice = (int64_t)a.M * a.N;
        while (S >= 2 && slice * groups * S > T->det_ws_floats) --S;
        if (S >= 2 && (a.N & 3) == 0) {
            GemmArgs p = a;
            p.C = T->det_ws; p.ldc = a.N; p.gC = slice * S;
            p.splitk = S; p.split_stride = slice;
            p.res = nullptr;
            int rc = launch_gemm(p, st, &ctx->err);
            if (rc) return rc;
            SplitCombineArgs c{};
            c.part = T->det_ws; c.nsplit = S; c.split_stride = slice; c.gP = slice * S; c.ldp = a.N;
            c.res = a.C; c.ldres = a.ldc; c.gRes = a.gC;                        // accumulate: C = C + sum of slices
            c.C = a.C; c.ldc = a.ldc; c.gC = a.gC;
            c.M = a.M; c.N = a.N; c.groups = groups;
            rc = launch_splitk_combine(c, st);
            if (rc) ctx->err = "splitk combine launch failed";
            return rc;
        }
        a.splitk = 0;                                   // cannot slice: one pass over the whole contraction, C += through res
        a.res = a.C; a.ldres = a.ldc; a.gRes = a.gC;
    }
    return launch_gemm(a, st, &ctx->err);
}

// LayerNorm backward / column sums with their partials summed in a fixed order (TrainState::det_ws)
int ln_bwd_d(actmi_ctx* ctx, const float* x, const float* w, const float* dy, const float* dx_add, float* dx, float* dw, float* db,
             int M, int D, float eps, hipStream_t st, unsigned* dx_amax = nullptr) {
    return launch_ln_bwd(x, w, dy, dx_add, dx, dw, db, M, D, eps, st, ctx->train->det_ws, ctx->train->det_ws_floats, dx_amax);
}
int colsum_d(actmi_ctx* ctx, const float* src, int64_t ld, float* out, int M, int N, hipStream_t st) {
    return launch_colsum(src, ld, out, M, N, st, ctx->train->det_ws, ctx->train->det_ws_floats);
}

// Gradient operands of the f16x3 backward GEMMs get a power-of-two scale computed on the device from their largest
// magnitude (gradients range from ~1e-7 at the stem to ~1e2 in the CVAE encoder: no single loss scale fits the fp16
// range).  A dgrad / wgrad pair on the same dY computes it once (one-shot reuse, cleared by any other request).
constexpr int SCALE_SLOTS = 1024;
const float* dyn_scale(actmi_ctx* ctx, const float* x, int64_t ld, int M, int N, hipStream_t st, bool keep = false) {
    if (ctx->gemm_prec != ACTMI_PREC_F16X3) return nullptr;
    TrainState& T = *ctx->train;
    if (T.amax_key_ptr == x && T.amax_key_ld == ld && T.amax_key_m == M && T.amax_key_n == N) {
        const float* slot = T.amax_key_slot;
        T.amax_key_ptr = nullptr;
        return slot;
    }
    if (T.amax_pre_ptr == x) {
        // the kernel that wrote x accumulated the bits of its largest magnitude (amax_pre): no pass over x
        float* slot = T.amax_pre_slot;
        T.amax_pre_ptr = nullptr;
        if (launch_pow2_from_bits(slot, st) != 0) return nullptr;
        // always kept for one more request: these tensors feed a weight-gradient / data-gradient pair
        T.amax_key_ptr = x; T.amax_key_ld = ld; T.amax_key_m = M; T.amax_key_n = N; T.amax_key_slot = slot;
        return slot;
    }
    float* slot = T.scale_slots + 2 * (T.scale_next++ % SCALE_SLOTS);
    if (launch_pow2_scale(x, ld, M, N, slot, st) != 0) return nullptr;
    T.amax_key_ptr = keep ? x : nullptr; T.amax_key_ld = ld; T.amax_key_m = M; T.amax_key_n = N; T.amax_key_slot = slot;
    return slot;
}

// a fresh scale slot for a caller that manages it by hand: slot[0] = the scale (after launch_pow2_from_bits), slot[1] = the bits
// word the producers raise (zero between uses)
float* scale_slot(actmi_ctx* ctx) {
    if (ctx->gemm_prec != ACTMI_PREC_F16X3) return nullptr;
    TrainState& T = *ctx->train;
    return T.scale_slots + 2 * (T.scale_next++ % SCALE_SLOTS);
}

// The kernel that WRITES a gradient tensor can collect the amax bits itself (relu_bn_bwd, the GEMM epilogue: an integer
// atomicMax, independent of arrival order): amax_pre() hands it the bits word of a fresh slot, and the dyn_scale() request
// for that tensor then only derives the scale from it.  Returns nullptr when no scale will be asked for.
unsigned* amax_pre(actmi_ctx* ctx, const float* out, hipStream_t st) {
    if (ctx->gemm_prec != ACTMI_PREC_F16X3) return nullptr;
    TrainState& T = *ctx->train;
    if (T.amax_pre_ptr)          // an unclaimed registration: re-arm its word (does not happen on the paths below)
        if (hipMemsetAsync(T.amax_pre_slot + 1, 0, sizeof(unsigned), st) != hipSuccess) return nullptr;
    if (T.amax_key_ptr == out) T.amax_key_ptr = nullptr;         // the tensor is being rewritten
    float* slot = T.scale_slots + 2 * (T.scale_next++ % SCALE_SLOTS);
    T.amax_pre_ptr = out; T.amax_pre_slot = slot;
    return reinterpret_cast<unsigned*>(slot + 1);
}

// weight operands of the backward GEMMs (f16x3) are split on the fly with the handle's static power-of-two scale
// ctx->bwd_wscale (2^8 unless a parameter is too large for it: engine_calibrate_weight_scales)

int pick_splitk(int M, int N, int groups, int K) {
    // outputs of at least one 128x128 tile per side: the launch runs 128x128 tiles, two per CU = 512 at a time.  Choose the split
    // that fills whole rounds of 512 (layer4's weight gradient: 576 workgroups unsplit = one full round + a round of 64 that lasts
    // as long -- 132 TF; layer3's: 432): the smallest S within 3 % of the best fill, at least 16 K tiles per split.
    static const bool fill_rule = !(getenv("ACTMI_WGRAD_SPLIT_FILL") && getenv("ACTMI_WGRAD_SPLIT_FILL")[0] == '0');
    if (fill_rule && M >= 128 && N >= 128) {
        const long w = (long)((M + 127) / 128) * ((N + 127) / 128) * groups;
        const long nk = (K + 31) / 32;
        long smax = nk / 16;
        if (smax > 64) smax = 64;
        if (smax < 1) smax = 1;
        double best = 0.0;
        for (long S = 1; S <= smax; ++S) {
            const double e = (double)(w * S) / (512.0 * (double)((w * S + 511) / 512));
            if (e > best) best = e;
        }
        for (long S = 1; S <= smax; ++S) {
            const double e = (double)(w * S) / (512.0 * (double)((w * S + 511) / 512));
            if (e >= best - 0.03) return (int)S;
        }
    }
    const long tiles = (long)((M + 127) / 128) * ((N + 63) / 64) * groups;
    long s = 1024 / (tiles > 0 ? tiles : 1);
    const long nk = (K + 31) / 32;
    if (s > nk / 4) s = nk / 4;          // keep >= 4 k-tiles per split
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    return (int)s;
}

// y[M][N] = x[M][K] W[N][K]^T + b (+res) (relu)
int lin_fwd(actmi_ctx* ctx, const float* x, int64_t ldx, int M, int K, const float* W, int N, const float* b, float* y,
            int64_t ldy, const float* res, int relu, hipStream_t st, float drop_p = 0.f, uint64_t drop_seed = 0) {
    GemmArgs a = G0();
    a.A = x; a.lda = ldx; a.M = M; a.K = K; a.N = N; a.Bw = W; a.ldb = K; a.bias = b; a.C = y; a.ldc = ldy;
    a.res = res; a.ldres = ldy; a.relu = relu; a.drop_p = drop_p; a.drop_seed = drop_seed;
    return tgemm(ctx, a, st);
}

// dx[M][K] = dy[M][N] W[N][K] (+res) (masked by mask>0)
int lin_dgrad(actmi_ctx* ctx, const float* dy, int64_t lddy, int M, int N, const float* W, int K, float* dx, int64_t lddx,
              const float* res, const float* mask, hipStream_t st, float alpha = 1.f, bool dx_feeds_gemm = false,
              const float* dy_scale = nullptr) {
    GemmArgs a = G0();
    a.A = dy; a.lda = lddy; a.M = M; a.K = N; a.N = K; a.Bw = W; a.ldb = K; a.tb = 1; a.C = dx; a.ldc = lddx;
    a.res = res; a.ldres = lddx; a.mask = mask; a.ldmask = lddx; a.alpha = alpha;
    a.b_scale = ctx->bwd_wscale;
    a.a_scale_dev = dy_scale ? dy_scale : dyn_scale(ctx, dy, lddy, M, N, st, true);
    // dx is the dY operand of the next data / weight gradient pair (same pointer, ld, M and K columns): its maximum is taken
    // in this product's epilogue
    if (dx_feeds_gemm && lddx == K) a.amax_out = amax_pre(ctx, dx, st);
    return tgemm(ctx, a, st);
}

// dropout sites of one layer: 0 attention weights, 1 after the attention out-projection, 2 FFN hidden, 3 after linear2
struct Drop {
    float p; uint64_t seed; uint32_t base;
    uint64_t s(uint32_t k) const { return actmi_site_seed(seed, base + k); }
};

// dW[N][K] += dy[M][N]^T x'[M][K],  x' = x + x_add[m % add_mod];  db[N] += colsum(dy)
int lin_wgrad(actmi_ctx* ctx, const float* dy, int64_t lddy, int M, int N, const float* x, int64_t ldx, int K,
              const float* x_add, int add_mod, float* dW, float* db, hipStream_t st, const float* dy_scale = nullptr) {
    if (dW) {
        GemmArgs a = G0();
        a.A = dy; a.lda = lddy; a.ta = 1; a.M = N; a.K = M; a.Bw = x; a.ldb = ldx; a.tb = 1; a.N = K;
        a.B_add = x_add; a.ld_badd = K; a.badd_mod = add_mod > 0 ? add_mod : 1;
        a.C = dW; a.ldc = K;
        a.splitk = pick_splitk(N, K, 1, M);
        if (a.splitk <= 1) { a.splitk = 0; a.res = dW; a.ldres = K; }
        a.a_scale_dev = dy_scale ? dy_scale : dyn_scale(ctx, dy, lddy, M, N, st);
        CHK(tgemm(ctx, a, st));
    }
    if (db) CHK(colsum_d(ctx, dy, lddy, db, M, N, st));
    return 0;
}

// ---- attention backward through materialised probabilities (batched MFMA products) -------------------------------
struct AttnBwd {
    const float *Q, *K, *V, *O, *dO, *lse;
    int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs;     // forward operand strides (q_bs = 0: shared queries)
    float *dQ, *dK, *dV;
    int64_t dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs;
    const uint8_t* kpm; int64_t kpm_bs;
    int B, H, Nq, Nk, HD;
    float drop_p; uint64_t drop_seed;
    unsigned* out_amax;        // optional: one bits word that the dQ, dK and dV products all raise (shared operand scale of gQKV)
};

// long sequences (the encoder's 1202 x 1202 self-attention) take the fused kernels of attn_bwd.hip: no P / dS buffers
static bool attn_bwd_fused(const actmi_ctx* ctx, int Nq, int Nk, int HD, bool shared_q) {
    static const bool flash = !(getenv("ACTMI_ATTN_BWD_FLASH") && getenv("ACTMI_ATTN_BWD_FLASH")[0] == '0');
    return flash && ctx->gemm_prec == ACTMI_PREC_F16X3 && !shared_q && Nq >= 256 && Nk >= 256 && (HD == 64 || HD == 32 || HD == 16);
}

int attn_bwd(actmi_ctx* ctx, const AttnBwd& t, hipStream_t st) {
    TrainState& T = *ctx->train;
    const int G = t.B * t.H, D = t.H * t.HD;
    {
        if (attn_bwd_fused(ctx, t.Nq, t.Nk, t.HD, t.q_bs == 0)) {
            CHK(launch_attn_delta(t.dO, t.O, T.delta, t.B, t.H, t.Nq, t.HD, st));
            AttnBwdArgs a{};
            a.Q = t.Q; a.K = t.K; a.V = t.V; a.dO = t.dO; a.lse = t.lse; a.delta = T.delta;
            a.dO_scale = dyn_scale(ctx, t.dO, D, t.B * t.Nq, D, st);
            a.dQ = t.dQ; a.dK = t.dK; a.dV = t.dV;
            a.q_bs = t.q_bs; a.q_rs = t.q_rs; a.k_bs = t.k_bs; a.k_rs = t.k_rs; a.v_bs = t.v_bs; a.v_rs = t.v_rs;
            a.do_bs = (int64_t)t.Nq * D; a.do_rs = D;
            a.dq_bs = t.dq_bs; a.dq_rs = t.dq_rs; a.dk_bs = t.dk_bs; a.dk_rs = t.dk_rs; a.dv_bs = t.dv_bs; a.dv_rs = t.dv_rs;
            a.kpm = t.kpm; a.kpm_bs = t.kpm_bs;
            a.B = t.B; a.H = t.H; a.Nq = t.Nq; a.Nk = t.Nk; a.HD = t.HD;
            a.scale = 1.0f / sqrtf((float)t.HD); a.drop_p = t.drop_p; a.drop_seed = t.drop_seed;
            a.amax_out = t.out_amax;
            return launch_attention_bwd(a, st, &ctx->err);
        }
    }
    const int ldp = (t.Nk + 3) & ~3;
    const float scale = 1.0f / sqrtf((float)t.HD);
    float* P = T.Pbuf;
    float* dP = T.dPbuf;
    const int64_t pg = (int64_t)t.Nq * ldp;
    // S = scale * Q K^T
    GemmArgs s = G0();
    s.A = t.Q; s.lda = t.q_rs; s.M = t.Nq; s.K = t.HD; s.Bw = t.K; s.ldb = t.k_rs; s.N = t.Nk; s.C = P; s.ldc = ldp;
    s.alpha = scale; s.groups = G; s.groups_inner = t.H;
    s.gA = t.q_bs; s.gA2 = t.HD; s.gB = t.k_bs; s.gB2 = t.HD; s.gC = pg * t.H; s.gC2 = pg;
    // P = exp(S - lse) (key-padded columns zero) straight from the epilogue of the score product; the columns Nk .. ldp-1 of
    // the buffer are cleared by a small kernel (the buffer is shared by attention calls of different widths)
    static const bool fuse = !(getenv("ACTMI_ATTN_BWD_FUSE") && getenv("ACTMI_ATTN_BWD_FUSE")[0] == '0');
    static const int sd_tile = getenv("ACTMI_ATTN_BWD_TILE") ? atoi(getenv("ACTMI_ATTN_BWD_TILE")) : 0;
    s.tile_hint = sd_tile;
    if (fuse) {
        s.epi = 1; s.epi_row = t.lse; s.gRow = (int64_t)t.H * t.Nq; s.gRow2 = t.Nq;
        s.epi_colkill = t.kpm; s.gColkill = t.kpm_bs;
        CHK(tgemm(ctx, s, st));
        if (ldp != t.Nk) CHK(launch_zero_cols(P, (int64_t)G * t.Nq, ldp, t.Nk, st));
    } else {
        CHK(tgemm(ctx, s, st));
        CHK(launch_attn_probs(P, t.lse, t.kpm, t.kpm_bs, G, t.H, t.Nq, t.Nk, ldp, st));
    }
    CHK(launch_attn_delta(t.dO, t.O, T.delta, t.B, t.H, t.Nq, t.HD, st));
    // dV[key][d] = sum_q Pd[q][key] dO[q][d]   (Pd = dropped weights; staged in the dP buffer before dP overwrites it)
    const float* Pv = P;
    if (t.drop_p > 0.f) {
        CHK(launch_attn_drop(P, dP, t.drop_seed, t.drop_p, G, t.Nq, t.Nk, ldp, st));
        Pv = dP;
    }
    GemmArgs v = G0();
    v.A = Pv; v.lda = ldp; v.ta = 1; v.M = t.Nk; v.K = t.Nq; v.Bw = t.dO; v.ldb = D; v.tb = 1; v.N = t.HD;
    v.C = t.dV; v.ldc = t.dv_rs; v.groups = G; v.groups_inner = t.H;
    v.gA = pg * t.H; v.gA2 = pg; v.gB = (int64_t)t.Nq * D; v.gB2 = t.HD; v.gC = t.dv_bs; v.gC2 = t.HD;
    v.a_scale = 256.f;          // probabilities (<= 1/(1-p)) are mostly ~1/Nk
    const float* dO_sc = dyn_scale(ctx, t.dO, D, t.B * t.Nq, D, st);
    v.b_scale_dev = dO_sc;
    v.amax_out = t.out_amax;
    CHK(tgemm(ctx, v, st));
    // dP = dO V^T
    GemmArgs d = G0();
    d.A = t.dO; d.lda = D; d.M = t.Nq; d.K = t.HD; d.Bw = t.V; d.ldb = t.v_rs; d.N = t.Nk; d.C = dP; d.ldc = ldp;
    d.groups = G; d.groups_inner = t.H;
    d.gA = (int64_t)t.Nq * D; d.gA2 = t.HD; d.gB = t.v_bs; d.gB2 = t.HD; d.gC = pg * t.H; d.gC2 = pg;
    d.a_scale_dev = dO_sc;
    d.tile_hint = sd_tile;
    const bool fuse_ds = fuse && !(t.drop_p > 0.f);
    if (fuse_ds) {
        // dS = P * (dP - delta) * scale in the epilogue of the dP product (dP itself is never stored), with the operand-scale
        // maximum of dS collected on the way out
        d.epi = 2; d.epi_scale = scale; d.epi_row = T.delta; d.gRow = (int64_t)t.H * t.Nq; d.gRow2 = t.Nq;
        d.res = P; d.ldres = ldp; d.gRes = pg * t.H; d.gRes2 = pg;
        d.amax_out = amax_pre(ctx, dP, st);
        CHK(tgemm(ctx, d, st));
        if (ldp != t.Nk) CHK(launch_zero_cols(dP, (int64_t)G * t.Nq, ldp, t.Nk, st));
    } else {
        CHK(tgemm(ctx, d, st));
        // dS = P * (dP - delta) * scale   (in place of dP; with dropout dP = dPd * mask / (1-p))
        if (t.drop_p > 0.f) CHK(launch_attn_ds_drop(P, dP, T.delta, scale, t.drop_seed, t.drop_p, G, t.Nq, t.Nk, ldp, st));
        else CHK(launch_attn_ds(P, dP, T.delta, scale, G, t.Nq, t.Nk, ldp, st, amax_pre(ctx, dP, st)));
    }
    // dQ[q][d] = sum_key dS[q][key] K[key][d]
    GemmArgs q = G0();
    q.A = dP; q.lda = ldp; q.M = t.Nq; q.K = t.Nk; q.Bw = t.K; q.ldb = t.k_rs; q.tb = 1; q.N = t.HD;
    q.C = t.dQ; q.ldc = t.dq_rs; q.groups = G; q.groups_inner = t.H;
    q.gA = pg * t.H; q.gA2 = pg; q.gB = t.k_bs; q.gB2 = t.HD; q.gC = t.dq_bs; q.gC2 = t.HD;
    const float* dS_sc = dyn_scale(ctx, dP, ldp, G * t.Nq, t.Nk, st);
    q.a_scale_dev = dS_sc;
    q.amax_out = t.out_amax;
    CHK(tgemm(ctx, q, st));
    // dK[key][d] = sum_q dS[q][key] Q[q][d]
    GemmArgs k = G0();
    k.A = dP; k.lda = ldp; k.ta = 1; k.M = t.Nk; k.K = t.Nq; k.Bw = t.Q; k.ldb = t.q_rs; k.tb = 1; k.N = t.HD;
    k.C = t.dK; k.ldc = t.dk_rs; k.groups = G; k.groups_inner = t.H;
    k.gA = pg * t.H; k.gA2 = pg; k.gB = t.q_bs; k.gB2 = t.HD; k.gC = t.dk_bs; k.gC2 = t.HD;
    k.a_scale_dev = dS_sc;
    k.amax_out = t.out_amax;
    CHK(tgemm(ctx, k, st));
    return 0;
}

// ---- one post-norm encoder layer, forward with saved activations (transformer.py:211-224) ------------------------
int enc_fwd(actmi_ctx* ctx, const EncW& w, EncSave& s, float* out, const float* pos, int B, int n, const uint8_t* kpm,
            const Drop& dr, hipStream_t st) {
    const actmi_config& g = ctx->cfg;
    const int D = g.hidden_dim, F = g.dim_feedforward, M = B * n, hd = D / g.nheads;
    GemmArgs qkv = G0();
    qkv.A = s.x_in; qkv.lda = D; qkv.M = M; qkv.K = D; qkv.Bw = w.attn.in_w; qkv.ldb = D; qkv.N = 3 * D;
    qkv.bias = w.attn.in_b; qkv.C = s.QKV; qkv.ldc = 3 * D;
    qkv.A_add = pos; qkv.ld_add = D; qkv.add_mod = n; qkv.add_ncols = 2 * D;
    CHK(tgemm(ctx, qkv, st));
    AttnArgs at;
    memset(&at, 0, sizeof(at));
    at.Q = s.QKV; at.q_bs = (int64_t)n * 3 * D; at.q_rs = 3 * D;
    at.K = s.QKV + D; at.k_bs = at.q_bs; at.k_rs = 3 * D;
    at.V = s.QKV + 2 * D; at.v_bs = at.q_bs; at.v_rs = 3 * D;
    at.O = s.ATT; at.o_bs = (int64_t)n * D; at.o_rs = D;
    at.kpm = kpm; at.kpm_bs = n; at.lse = s.lse;
    at.B = B; at.H = g.nheads; at.Nq = n; at.Nk = n; at.HD = hd; at.scale = 1.0f / sqrtf((float)hd);
    at.ws = ctx->attn_ws; at.ws_floats = ctx->attn_ws_floats;
    at.prec = ctx->gemm_prec;
    at.drop_p = dr.p; at.drop_seed = dr.s(0);
    CHK(launch_attention(at, st, &ctx->err));
    CHK(lin_fwd(ctx, s.ATT, D, M, D, w.attn.out_w, D, w.attn.out_b, s.Y1, D, s.x_in, 0, st, dr.p, dr.s(1)));
    CHK(launch_layernorm(s.Y1, nullptr, 0, w.n1w, w.n1b, nullptr, nullptr, s.X1, M, D, 1e-5f, st, &ctx->err));
    CHK(lin_fwd(ctx, s.X1, D, M, D, w.l1w, F, w.l1b, s.Hb, F, nullptr, 1, st, dr.p, dr.s(2)));
    CHK(lin_fwd(ctx, s.Hb, F, M, F, w.l2w, D, w.l2b, s.Y2, D, s.X1, 0, st, dr.p, dr.s(3)));
    CHK(launch_layernorm(s.Y2, nullptr, 0, w.n2w, w.n2b, nullptr, nullptr, out, M, D, 1e-5f, st, &ctx->err));
    return 0;
}

// backward of the same layer: dOut -> dIn (dIn may alias T.gB); grads accumulate into the gradient arena
int enc_bwd(actmi_ctx* ctx, const EncW& w, const EncSave& s, const float* dOut, float* dIn, const float* pos, int B, int n,
            const uint8_t* kpm, float* dpos2 /* [2][D] additional_pos_embed grad or null */, const Drop& dr, hipStream_t st) {
    TrainState& T = *ctx->train;
    const actmi_config& g = ctx->cfg;
    const int D = g.hidden_dim, F = g.dim_feedforward, M = B * n, hd = D / g.nheads;
    auto Gp = [&](const float* p) { return T.gbase + (p - ctx->pbase); };
    float* gA = T.gA; float* gC = T.gC; float* gH = T.gH; float* gQKV = T.gQKV;
    const bool drop = dr.p > 0.f;
    const float inv_keep = drop ? 1.f / (1.f - dr.p) : 1.f;
    // norm2:  Y2 = X1 + drop3(linear2(Hb))
    CHK(ln_bwd_d(ctx, s.Y2, w.n2w, dOut, nullptr, gA, Gp(w.n2w), Gp(w.n2b), M, D, 1e-5f, st,
                 drop ? nullptr : amax_pre(ctx, gA, st)));                                                   // gA = dY2
    const float* dz2 = gA;
    if (drop) { CHK(launch_dropout_bwd(gA, gC, dr.s(3), dr.p, (int64_t)M * D, st)); dz2 = gC; }
    // linear2 / dropout / relu / linear1:  Hb = drop2(relu(linear1(X1))); dropped or negative entries are 0 in Hb
    CHK(lin_dgrad(ctx, dz2, D, M, D, w.l2w, F, gH, F, nullptr, s.Hb, st, inv_keep, true));                   // gH = dHpre
    CHK(lin_wgrad(ctx, dz2, D, M, D, s.Hb, F, F, nullptr, 0, Gp(w.l2w), Gp(w.l2b), st));
    CHK(lin_dgrad(ctx, gH, F, M, F, w.l1w, D, gC, D, gA, nullptr, st));                                      // gC = dX1
    CHK(lin_wgrad(ctx, gH, F, M, F, s.X1, D, D, nullptr, 0, Gp(w.l1w), Gp(w.l1b), st));
    // norm1:  Y1 = x_in + drop1(out_proj(ATT))
    CHK(ln_bwd_d(ctx, s.Y1, w.n1w, gC, nullptr, gA, Gp(w.n1w), Gp(w.n1b), M, D, 1e-5f, st,
                 drop ? nullptr : amax_pre(ctx, gA, st)));                                                   // gA = dY1
    const float* dz1 = gA;
    if (drop) { CHK(launch_dropout_bwd(gA, gC, dr.s(1), dr.p, (int64_t)M * D, st)); dz1 = gC; }
    float* dATT = gH;                                                                                        // [M][D] view
    CHK(lin_dgrad(ctx, dz1, D, M, D, w.attn.out_w, D, dATT, D, nullptr, nullptr, st, 1.f, true));            // dATT = dO of the attention
    CHK(lin_wgrad(ctx, dz1, D, M, D, s.ATT, D, D, nullptr, 0, Gp(w.attn.out_w), Gp(w.attn.out_b), st));
    // attention
    AttnBwd t;
    memset(&t, 0, sizeof(t));
    const int64_t bs = (int64_t)n * 3 * D;
    t.Q = s.QKV; t.K = s.QKV + D; t.V = s.QKV + 2 * D; t.O = s.ATT; t.dO = dATT; t.lse = s.lse;
    t.q_bs = t.k_bs = t.v_bs = bs; t.q_rs = t.k_rs = t.v_rs = 3 * D;
    t.dQ = gQKV; t.dK = gQKV + D; t.dV = gQKV + 2 * D;
    t.dq_bs = t.dk_bs = t.dv_bs = bs; t.dq_rs = t.dk_rs = t.dv_rs = 3 * D;
    t.kpm = kpm; t.kpm_bs = n; t.B = B; t.H = g.nheads; t.Nq = n; t.Nk = n; t.HD = hd;
    t.drop_p = dr.p; t.drop_seed = dr.s(0);
    // ONE operand scale for gQKV = [dQ | dK | dV]: the three products that write it raise the same bits word, and the data
    // gradient and the two weight gradients below all use it (three strided amax passes over gQKV otherwise)
    float* qkv_slot = scale_slot(ctx);
    t.out_amax = qkv_slot ? reinterpret_cast<unsigned*>(qkv_slot + 1) : nullptr;
    CHK(attn_bwd(ctx, t, st));
    if (qkv_slot && launch_pow2_from_bits(qkv_slot, st) != 0) return ACTMI_E_LAUNCH;
    // in_proj: dIn = dQKV W_in + dY1 ; dW rows [0,2D) see x+pos, rows [2D,3D) see x
    CHK(lin_dgrad(ctx, gQKV, 3 * D, M, 3 * D, w.attn.in_w, D, dIn, D, gA, nullptr, st, 1.f, false, qkv_slot));
    CHK(lin_wgrad(ctx, gQKV, 3 * D, M, 2 * D, s.x_in, D, D, pos, n, Gp(w.attn.in_w), nullptr, st, qkv_slot));
    CHK(lin_wgrad(ctx, gQKV + 2 * D, 3 * D, M, D, s.x_in, D, D, nullptr, 0, Gp(w.attn.in_w) + (int64_t)2 * D * D, nullptr, st, qkv_slot));
    CHK(colsum_d(ctx, gQKV, 3 * D, Gp(w.attn.in_b), M, 3 * D, st));
    if (dpos2) {
        // additional_pos_embed rows: d(x+pos)[b][j] = dQK[b][j] W_in[0:2D], summed over the batch, j in {0,1}
        GemmArgs a = G0();
        a.A = gQKV; a.lda = 3 * D; a.a_rowmap = T.pos_rows; a.M = 2 * B; a.K = 2 * D; a.Bw = w.attn.in_w; a.ldb = D; a.tb = 1;
        a.N = D; a.C = T.tmp2BD; a.ldc = D;
        a.b_scale = ctx->bwd_wscale;
        a.a_scale_dev = dyn_scale(ctx, gQKV, 3 * D, M, 2 * D, st);
        CHK(tgemm(ctx, a, st));
        CHK(launch_sum_batch(T.tmp2BD, 2 * D, D, dpos2, B, 2, D, 1, st));
    }
    return 0;
}

// ---- convolution backward pieces --------------------------------------------------------------------------------------
int conv_wgrad(actmi_ctx* ctx, const ConvLayer& cl, int li, const float* dys, const float* x, int B, hipStream_t st) {
    TrainState& T = *ctx->train;
    const int C = ctx->cfg.num_cams;
    static const bool direct_on = !(getenv("ACTMI_WGRAD_DIRECT") && getenv("ACTMI_WGRAD_DIRECT")[0] == '0');
    if (direct_on && ctx->gemm_prec == ACTMI_PREC_F16X3 && cl.cin == 64 && cl.cout == 64 && cl.k == 3 && cl.stride == 1 && cl.pad == 1 &&
        T.det_ws && T.det_ws_floats >= (int64_t)C * 64 * 576) {
        // layer1: the direct kernel (wgrad3.hip) + fixed-order sum of its per-workgroup partials into the packed gradient
        const float* sc = dyn_scale(ctx, dys, cl.cout, C * B * cl.Ho * cl.Wo, cl.cout, st);
        int nwg = 0;
        if (launch_wgrad3x3_c64(dys, x, T.det_ws, T.det_ws_floats, sc, C, B, cl.H, cl.W, &nwg, st) != 0) {
            ctx->err = "wgrad3x3_c64 launch failed";
            return ACTMI_E_LAUNCH;
        }
        SplitCombineArgs c{};
        const int64_t slice = (int64_t)64 * 576;
        c.part = T.det_ws; c.nsplit = nwg; c.split_stride = slice; c.gP = slice * nwg; c.ldp = 576;
        c.res = T.conv_gw[li]; c.ldres = cl.K; c.gRes = (int64_t)cl.cout * cl.K;         // accumulate like autograd
        c.C = T.conv_gw[li]; c.ldc = cl.K; c.gC = (int64_t)cl.cout * cl.K;
        c.M = 64; c.N = 576; c.groups = C;
        if (launch_splitk_combine(c, st) != 0) { ctx->err = "splitk combine launch failed"; return ACTMI_E_LAUNCH; }
        return 0;
    }
    GemmArgs a = G0();
    a.A = dys; a.lda = cl.cout; a.ta = 1; a.M = cl.cout; a.K = B * cl.Ho * cl.Wo;
    a.Bw = x; a.tb = 2; a.N = cl.K; a.H = cl.H; a.W = cl.W; a.Cin = cl.cin; a.KH = a.KW = cl.k; a.stride = cl.stride;
    a.pad = cl.pad; a.Ho = cl.Ho; a.Wo = cl.Wo; a.img_stride = (int64_t)cl.H * cl.W * cl.cin;
    a.C = T.conv_gw[li]; a.ldc = cl.K; a.groups = C;
    a.gA = (int64_t)B * cl.Ho * cl.Wo * cl.cout; a.gB = (int64_t)B * cl.H * cl.W * cl.cin; a.gC = (int64_t)cl.cout * cl.K;
    a.splitk = pick_splitk(cl.cout, cl.K, C, a.K);
    if (a.splitk <= 1) { a.splitk = 0; a.res = a.C; a.ldres = cl.K; a.gRes = a.gC; }
    a.a_scale_dev = dyn_scale(ctx, dys, cl.cout, C * B * cl.Ho * cl.Wo, cl.cout, st);
    return tgemm(ctx, a, st);
}

// layer1's data gradients (64 -> 64 channels, 3x3 / s1 / p1) run as a forward convolution of dY with the flipped, transposed
// weights on the direct kernel of the inference path (conv3.hip: LDS-resident patch, 1.3 ms against the gather GEMM's 2.5 ms)
bool dgrad_direct(const actmi_ctx* ctx, const ConvLayer& cl) {
    static const bool on = !(getenv("ACTMI_DGRAD_DIRECT") && getenv("ACTMI_DGRAD_DIRECT")[0] == '0');
    return on && ctx->gemm_prec == ACTMI_PREC_F16X3 && cl.cin == 64 && cl.cout == 64 && cl.k == 3 && cl.stride == 1 && cl.pad == 1;
}

// dx[C][B][H][W][cin] = dgrad(dys) (+res) , then masked by (mask > 0) and multiplied by scale[cin] (previous BN)
int conv_dgrad(actmi_ctx* ctx, const ConvLayer& cl, int li, const float* dys, float* dx, const float* res, const float* mask,
               const float* scale, int B, hipStream_t st, bool dx_feeds_gemm = false) {
    TrainState& T = *ctx->train;
    const int C = ctx->cfg.num_cams;
    if (dgrad_direct(ctx, cl) && T.conv_wd16[li]) {
        Conv3Args c{};
        c.x = dys; c.w16 = T.conv_wd16[li]; c.res = res; c.out = dx; c.G = C; c.B = B; c.H = cl.H; c.W = cl.W; c.relu = 0;
        c.w_scale = ctx->bwd_wscale;
        c.x_scale_dev = dyn_scale(ctx, dys, cl.cout, C * B * cl.Ho * cl.Wo, cl.cout, st, true);
        c.mask = mask; c.post_scale = scale;
        if (dx_feeds_gemm) c.amax_out = amax_pre(ctx, dx, st);
        return launch_conv3x3_c64(c, st, &ctx->err) == 0 ? 0 : ACTMI_E_LAUNCH;
    }
    GemmArgs a = G0();
    a.mode = 2; a.A = dys; a.H = cl.H; a.W = cl.W; a.Cin = cl.cin; a.KH = a.KW = cl.k; a.stride = cl.stride; a.pad = cl.pad;
    a.Ho = cl.Ho; a.Wo = cl.Wo; a.img_stride = (int64_t)cl.Ho * cl.Wo * cl.cout;
    a.M = B * cl.H * cl.W; a.N = cl.cin; a.K = cl.k * cl.k * cl.cout;
    a.Bw = T.conv_wd[li]; a.ldb = a.K; a.C = dx; a.ldc = cl.cin; a.groups = C;
    a.gA = (int64_t)B * cl.Ho * cl.Wo * cl.cout; a.gB = (int64_t)cl.cin * a.K; a.gC = (int64_t)a.M * cl.cin;
    a.res = res; a.ldres = cl.cin; a.gRes = a.gC; a.mask = mask; a.ldmask = cl.cin; a.gMask = a.gC;
    a.scale = scale; a.gSB = cl.cin;
    a.b_scale = ctx->bwd_wscale;
    a.a_scale_dev = dyn_scale(ctx, dys, cl.cout, C * B * cl.Ho * cl.Wo, cl.cout, st, true);
    if (dx_feeds_gemm) a.amax_out = amax_pre(ctx, dx, st);       // dx is the dY operand of the next wgrad / dgrad pair
    return tgemm(ctx, a, st);
}

}  // namespace

// ----------------------------------------------------------------------------------------------------------------------
// allocation
// ----------------------------------------------------------------------------------------------------------------------

int train_create(actmi_ctx* ctx) {
    const actmi_config& g = ctx->cfg;
    ctx->train = new TrainState();
    TrainState& T = *ctx->train;
    const int B = g.max_batch, C = g.num_cams, D = g.hidden_dim, F = g.dim_feedforward, Q = g.num_queries, N = ctx->N,
              w0 = g.base_width, H = g.nheads, L = g.latent_dim, A = g.action_dim;
    int rc;
#define TA(ptr, n) if ((rc = talloc(ctx, &(ptr), (n)))) return rc
    TA(T.gbase, ctx->ptotal); TA(T.mbase, ctx->ptotal); TA(T.vbase, ctx->ptotal);
    if (hipMemset(T.gbase, 0, ctx->ptotal * 4) != hipSuccess || hipMemset(T.mbase, 0, ctx->ptotal * 4) != hipSuccess ||
        hipMemset(T.vbase, 0, ctx->ptotal * 4) != hipSuccess) { ctx->err = "hipMemset failed"; return ACTMI_E_LAUNCH; }
    {
        // optimizer group per 64-float slot: 0 skip (buffers; is_pad_head never gets a grad), 1 lr, 2 lr_backbone
        std::vector<uint8_t> grp((size_t)(ctx->ptotal / 64), 0);
        for (const Param& p : ctx->params) {
            uint8_t gcode = 0;
            if (!p.is_buffer && p.key.rfind("is_pad_head", 0) != 0) gcode = p.key.find("backbone") != std::string::npos ? 2 : 1;
            for (int64_t c = p.off / 64; c < (p.off + ((p.numel + 63) & ~int64_t(63))) / 64; ++c) grp[(size_t)c] = gcode;
        }
        float* gp = nullptr;
        TA(gp, (int64_t)(grp.size() + 3) / 4);
        T.group = reinterpret_cast<uint8_t*>(gp);
        if (hipMemcpy(T.group, grp.data(), grp.size(), hipMemcpyHostToDevice) != hipSuccess) { ctx->err = "hipMemcpy failed"; return ACTMI_E_LAUNCH; }
    }
    // backbone saves
    const int64_t n2 = (int64_t)C * B * ctx->H2 * ctx->W2 * w0;
    const int64_t n1 = (int64_t)C * B * ctx->H1 * ctx->W1 * w0;
    TA(T.xn4, (int64_t)C * B * g.image_h * g.image_w * 4);
    TA(T.pool, n2);
    TA(T.g_act1, n1);
    for (int i = 0; i < 4; ++i) TA(T.gbuf[i], n2);
    {
        size_t ci = 0;
        for (int li = 1; li <= 4; ++li)
            for (int bi = 0; bi < 2; ++bi) {
                const ConvLayer& k1 = ctx->convs[ci];
                BlockSave bs;
                bs.c1 = (int)ci; bs.c2 = (int)ci + 1; bs.ds = (bi == 0 && li > 1) ? (int)ci + 2 : -1;
                ci += (bs.ds >= 0) ? 3 : 2;
                const int64_t nout = (int64_t)C * B * k1.Ho * k1.Wo * k1.cout;
                TA(bs.y1, nout); TA(bs.out, nout);
                T.blocks.push_back(bs);
            }
    }
    for (auto& cl : ctx->convs) {
        float *gw, *wd;
        TA(gw, (int64_t)C * cl.cout * cl.K); TA(wd, (int64_t)C * cl.cin * cl.k * cl.k * cl.cout);
        T.conv_gw.push_back(gw); T.conv_wd.push_back(wd);
        float* wd16 = nullptr;
        if (dgrad_direct(ctx, cl)) TA(wd16, (int64_t)C * cl.cin * cl.k * cl.k * cl.cout);
        T.conv_wd16.push_back(wd16);
    }
    TA(T.conv1_gw, (int64_t)C * w0 * 196);
    {
        float* pa = nullptr;           // argmax codes of the stem max-pool, one byte per pooled element
        TA(pa, ((int64_t)C * B * ctx->H2 * ctx->W2 * w0 + 3) / 4);
        T.pool_arg = reinterpret_cast<uint8_t*>(pa);
    }
    TA(T.scale_slots, 2 * SCALE_SLOTS);
    // deterministic reductions: slices of split weight-gradient contractions and per-block partials of the LayerNorm /
    // bias-gradient sums, all combined in a fixed order (no float atomics: gradients are bitwise repeatable)
    if (hipEventCreateWithFlags(&T.ev_phase1, hipEventDisableTiming) != hipSuccess) { ctx->err = "hipEventCreate failed"; return ACTMI_E_LAUNCH; }
    T.det_ws_floats = (int64_t)48 << 20;
    TA(T.det_ws, T.det_ws_floats);
    if (hipMemset(T.scale_slots, 0, 2 * SCALE_SLOTS * 4) != hipSuccess) { ctx->err = "hipMemset failed"; return ACTMI_E_LAUNCH; }
    // transformer saves
    auto alloc_enc = [&](std::vector<EncSave>& v, int n, float* first_in) -> int {
        const int64_t M = (int64_t)B * n;
        for (int l = 0; l < g.enc_layers; ++l) {
            EncSave s;
            if (l == 0 && first_in) s.x_in = first_in; else TA(s.x_in, M * D);
            TA(s.QKV, M * 3 * D); TA(s.ATT, M * D); TA(s.Y1, M * D); TA(s.X1, M * D); TA(s.Hb, M * F); TA(s.Y2, M * D);
            TA(s.lse, (int64_t)B * H * n);
            v.push_back(s);
        }
        return 0;
    };
    if ((rc = alloc_enc(T.en, N, ctx->X))) return rc;
    TA(T.mem, (int64_t)B * N * D);
    if (g.has_cvae_encoder) {
        TA(T.Xc, (int64_t)B * (Q + 2) * D);
        if ((rc = alloc_enc(T.cv, Q + 2, T.Xc))) return rc;
        TA(T.cv_out, (int64_t)B * (Q + 2) * D);
        float* t1; TA(t1, (int64_t)B * Q); T.cmap = reinterpret_cast<int*>(t1);
        float* t2; TA(t2, ((int64_t)B * (Q + 2) + 3) / 4 + 1); T.ckpm = reinterpret_cast<uint8_t*>(t2);
        const int Lp = g.vq ? g.vq_class * g.vq_dim : 2 * L, Lz = g.vq ? g.vq_class * g.vq_dim : L;
        TA(T.latent_info, (int64_t)B * Lp); TA(T.z, (int64_t)B * Lz); TA(T.eps, (int64_t)B * Lz);
        TA(T.d_latent_info, (int64_t)B * Lp); TA(T.dz, (int64_t)B * Lz);
        if (g.vq) TA(T.vq_probs, (int64_t)B * Lp);
    }
    // decoder layer 0 saves
    const int64_t BQ = (int64_t)B * Q;
    TA(T.sa_tmp, D); TA(T.t1, D); TA(T.qin, (int64_t)Q * D); TA(T.dq, (int64_t)Q * D);
    TA(T.KV, (int64_t)B * N * 2 * D); TA(T.lse_c, (int64_t)B * H * Q); TA(T.Oc, BQ * D); TA(T.Y2pre, BQ * D);
    TA(T.T2, BQ * D); TA(T.Hd, BQ * F); TA(T.Y3pre, BQ * D); TA(T.T3, BQ * D); TA(T.hs, BQ * D);
    TA(T.a_hat, BQ * A); TA(T.actions, BQ * A);
    TA(T.qkd, (int64_t)Q * 2 * D); TA(T.sO, BQ * D); TA(T.lse_s, (int64_t)B * H * Q); TA(T.saB, BQ * D); TA(T.T1B, BQ * D);
    TA(T.dqB, BQ * D); TA(T.gT1, BQ * D); TA(T.dsaB, BQ * D); TA(T.dqkB, BQ * 2 * D); TA(T.dvB, BQ * D); TA(T.dqk_d, (int64_t)Q * 2 * D);
    TA(T.tmpQD, (int64_t)Q * D);
    { float* t; TA(t, (BQ + 3) / 4 + 1); T.is_pad = reinterpret_cast<uint8_t*>(t); }
    TA(T.losses, 4 + 520);               // [l1, kl, loss, -] + block partials of the l1 sum (launch_losses)
    // backward scratch
    const int64_t MN = (int64_t)B * N;
    TA(T.gA, MN * D); TA(T.gB, MN * D); TA(T.gC, MN * D); TA(T.gH, MN * F); TA(T.gQKV, MN * 3 * D);
    const int ldp = (N + 3) & ~3;
    {
        // materialised P / dS of the attention calls that do not take the fused backward: decoder cross-attention (Q x N), the
        // CVAE encoder ((Q+2)^2), and -- only without the fused kernels -- the encoder's N x N (2 x 3 GB at B = 64)
        const int ldq = (Q + 2 + 3) & ~3;
        int64_t prow = (int64_t)Q * ldp;
        if ((int64_t)(Q + 2) * ldq > prow) prow = (int64_t)(Q + 2) * ldq;
        if (!attn_bwd_fused(ctx, N, N, D / H, false) && (int64_t)N * ldp > prow) prow = (int64_t)N * ldp;
        TA(T.Pbuf, (int64_t)B * H * prow); TA(T.dPbuf, (int64_t)B * H * prow); TA(T.delta, (int64_t)B * H * (N > Q + 2 ? N : Q + 2));
    }
    TA(T.dXg, (int64_t)B * (C * ctx->P_ > Q ? C * ctx->P_ : Q) * D);
    TA(T.tmp2BD, (int64_t)2 * B * D); TA(T.tmpD, 4 * D); TA(T.dqb, BQ * D);
    {
        std::vector<int> rows(2 * B);
        for (int b = 0; b < B; ++b) { rows[2 * b] = b * N; rows[2 * b + 1] = b * N + 1; }
        float* t; TA(t, 2 * B); T.pos_rows = reinterpret_cast<int*>(t);
        if (hipMemcpy(T.pos_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { ctx->err = "hipMemcpy failed"; return ACTMI_E_LAUNCH; }
        // note: rows are for the max batch layout; entries b < B_call are valid for any smaller batch too
    }
#undef TA
    return 0;
}

// ----------------------------------------------------------------------------------------------------------------------
// forward (training)
// ----------------------------------------------------------------------------------------------------------------------

int train_forward(actmi_ctx* ctx, const float* qpos, const void* image, int fmt, const float* actions, const uint8_t* is_pad,
                  const float* eps, uint64_t dropout_seed, float dropout_p, int B, float* losses, float* a_hat_out,
                  float* mu_out, float* logvar_out, hipStream_t st) {
    ctx->err.clear();
    if (!ctx->train) { ctx->err = "handle was created without enable_training"; return ACTMI_E_STATE; }
    if (!ctx->finalized) { ctx->err = "forward before finalize"; return ACTMI_E_STATE; }
    if (B < 1 || B > ctx->cfg.max_batch) { ctx->err = "batch exceeds max_batch"; return ACTMI_E_INVALID; }
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) { ctx->err = "dropout_p must be in [0, 1)"; return ACTMI_E_INVALID; }
    PrecScope prec_scope(ctx);               // the opt-in bf16 product mode covers the GEMMs of this call only
    TrainState& T = *ctx->train;
    const actmi_config& g = ctx->cfg;
    const int C = g.num_cams, D = g.hidden_dim, F = g.dim_feedforward, Q = g.num_queries, N = ctx->N, w0 = g.base_width,
              L = g.latent_dim, A = g.action_dim, S = g.state_dim, hd = D / g.nheads;
    T.B = B; T.fmt = fmt; T.drop_p = dropout_p; T.drop_seed = dropout_seed;
    const Drop dr_dec{dropout_p, dropout_seed, 200};
    HIPCHK(hipMemcpyAsync(T.actions, actions, (size_t)B * Q * A * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(T.is_pad, is_pad, (size_t)B * Q, hipMemcpyDeviceToDevice, st));
    T.qpos = qpos;

    // ---- CVAE encoder (detr_vae.py:117-151)
    if (g.has_cvae_encoder) {
        const int Lp = g.vq ? g.vq_class * g.vq_dim : 2 * L, Lz = g.vq ? g.vq_class * g.vq_dim : L;
        if (!eps && !g.vq) { ctx->err = "eps is required (the reference draws it in reparametrize, detr_vae.py:19-22)"; return ACTMI_E_INVALID; }
        T.have_eps = eps != nullptr;          // VQ: eps carries the one-hot code; NULL = draw it on the device
        if (eps) HIPCHK(hipMemcpyAsync(T.eps, eps, (size_t)B * Lz * 4, hipMemcpyDeviceToDevice, st));
        CHK(launch_cvae_maps(T.cmap, T.ckpm, T.is_pad, B, Q, st));
        const int n = Q + 2;
        CHK(launch_fill_rows(T.Xc, D, (int64_t)n * D, ctx->P("cls_embed.weight"), 0, B, D, st));
        CHK(launch_small_linear(qpos, S, ctx->P("encoder_joint_proj.weight"), ctx->P("encoder_joint_proj.bias"), T.Xc + D,
                                (int64_t)n * D, B, D, S, st));
        GemmArgs ap = G0();
        ap.A = T.actions; ap.lda = A; ap.M = B * Q; ap.K = A; ap.Bw = ctx->P("encoder_action_proj.weight"); ap.ldb = A; ap.N = D;
        ap.bias = ctx->P("encoder_action_proj.bias"); ap.C = T.Xc; ap.ldc = D; ap.rowmap = T.cmap;
        CHK(tgemm(ctx, ap, st));
        for (int l = 0; l < g.enc_layers; ++l) {
            float* out = (l + 1 < g.enc_layers) ? T.cv[l + 1].x_in : T.cv_out;
            CHK(enc_fwd(ctx, ctx->cvae[l], T.cv[l], out, ctx->P("pos_table"), B, n, T.ckpm, Drop{dropout_p, dropout_seed, (uint32_t)(8 * l)}, st));
        }
        CHK(lin_fwd(ctx, T.cv_out, (int64_t)n * D, B, D, ctx->P("latent_proj.weight"), Lp, ctx->P("latent_proj.bias"),
                    T.latent_info, Lp, nullptr, 0, st));
        if (g.vq) {
            // VQ-ACT (detr_vae.py:137-145): probs = softmax per class; code = given one-hot sample or a device draw;
            // straight-through: latent_input = latent_out_proj(code).  mu_out / logvar_out carry probs / code.
            CHK(launch_vq_code(T.latent_info, T.have_eps ? T.eps : nullptr, actmi_site_seed(dropout_seed, 900), T.vq_probs, T.z, B,
                               g.vq_class, g.vq_dim, st));
            if (mu_out) HIPCHK(hipMemcpyAsync(mu_out, T.vq_probs, (size_t)B * Lp * 4, hipMemcpyDeviceToDevice, st));
            if (logvar_out) HIPCHK(hipMemcpyAsync(logvar_out, T.z, (size_t)B * Lz * 4, hipMemcpyDeviceToDevice, st));
        } else {
            CHK(launch_reparam(T.latent_info, T.eps, T.z, mu_out, logvar_out, B, L, st));
        }
        CHK(launch_small_linear(T.z, Lz, ctx->P("latent_out_proj.weight"), ctx->P("latent_out_proj.bias"), ctx->X,
                                (int64_t)N * D, B, D, Lz, st));
    } else {
        CHK(launch_fill_rows(ctx->X, D, (int64_t)N * D, ctx->P("latent_out_proj.bias"), 0, B, D, st));
    }

    // ---- backbone with saved maps
    CHK(launch_normalize_pad(image, fmt, ctx->lut, T.xn4, B, C, g.image_h, g.image_w, st));
    {
        Conv1Args c1;
        c1.image = image; c1.fmt = fmt; c1.lut = ctx->lut; c1.w = ctx->conv1_w; c1.scale = ctx->conv1_scale;
        c1.bias = ctx->conv1_bias; c1.out = ctx->act1; c1.B = B; c1.C = C; c1.H = g.image_h; c1.W = g.image_w;
        c1.Ho = ctx->H1; c1.Wo = ctx->W1; c1.Cout = w0;
        c1.prec = ctx->gemm_prec;
        c1.wimg = reinterpret_cast<const unsigned char*>(ctx->conv1_wimg);
        c1.wscale = ctx->conv1_wscale;
        CHK(launch_conv1(c1, st, &ctx->err));
        CHK(launch_maxpool_idx(ctx->act1, T.pool, T.pool_arg, C * B, ctx->H1, ctx->W1, w0, ctx->H2, ctx->W2, st));
    }
    auto run_conv = [&](const ConvLayer& cl, const float* in, float* out, const float* res, int relu) -> int {
        if (ctx->gemm_prec == ACTMI_PREC_F16X3 && cl.k == 3 && cl.stride == 1 && cl.pad == 1 && cl.cin == 64 && cl.cout == 64) {
            Conv3Args c3;           // layer1: direct convolution (conv3.hip), as in the inference engine
            c3.x = in; c3.w16 = cl.w16; c3.scale = cl.scale; c3.bias = cl.bias; c3.res = res; c3.out = out;
            c3.G = C; c3.B = B; c3.H = cl.H; c3.W = cl.W; c3.relu = relu; c3.w_scale = cl.w16_scale;
            return launch_conv3x3_c64(c3, st, &ctx->err);
        }
        GemmArgs a = G0();
        a.mode = 1;
        a.A = in; a.H = cl.H; a.W = cl.W; a.Cin = cl.cin; a.KH = a.KW = cl.k; a.stride = cl.stride; a.pad = cl.pad;
        a.Ho = cl.Ho; a.Wo = cl.Wo; a.img_stride = (int64_t)cl.H * cl.W * cl.cin;
        a.M = B * cl.Ho * cl.Wo; a.N = cl.cout; a.K = cl.K;
        a.Bw = cl.w; a.ldb = cl.K; a.scale = cl.scale; a.bias = cl.bias; a.res = res; a.ldres = cl.cout; a.relu = relu;
        a.C = out; a.ldc = cl.cout; a.groups = C;
        a.gA = (int64_t)B * cl.H * cl.W * cl.cin; a.gB = (int64_t)cl.cout * cl.K; a.gSB = cl.cout;
        a.gC = (int64_t)a.M * cl.cout; a.gRes = a.gC;
        return tgemm(ctx, a, st);
    };
    const float* x = T.pool;
    for (auto& bs : T.blocks) {
        CHK(run_conv(ctx->convs[bs.c1], x, bs.y1, nullptr, 1));
        const float* idt = x;
        if (bs.ds >= 0) {
            CHK(run_conv(ctx->convs[bs.ds], x, T.gbuf[0], nullptr, 0));
            idt = T.gbuf[0];
        }
        CHK(run_conv(ctx->convs[bs.c2], bs.y1, bs.out, idt, 1));
        x = bs.out;
    }
    if (ctx->rowmap_B != B) {
        CHK(launch_build_rowmap(ctx->rowmap, B, C, ctx->fh, ctx->fw, N, st));
        ctx->rowmap_B = B;
    }
    {
        GemmArgs ip = G0();
        ip.A = x; ip.lda = 8 * w0; ip.M = C * B * ctx->P_; ip.K = 8 * w0; ip.Bw = ctx->P("input_proj.weight"); ip.ldb = 8 * w0;
        ip.N = D; ip.bias = ctx->P("input_proj.bias"); ip.C = ctx->X; ip.ldc = D; ip.rowmap = ctx->rowmap;
        CHK(tgemm(ctx, ip, st));
    }
    CHK(launch_small_linear(qpos, S, ctx->P("input_proj_robot_state.weight"), ctx->P("input_proj_robot_state.bias"),
                            ctx->X + D, (int64_t)N * D, B, D, S, st));

    // ---- encoder
    for (int l = 0; l < g.enc_layers; ++l) {
        float* out = (l + 1 < g.enc_layers) ? T.en[l + 1].x_in : T.mem;
        CHK(enc_fwd(ctx, ctx->enc[l], T.en[l], out, ctx->pos_tokens, B, N, nullptr, Drop{dropout_p, dropout_seed, (uint32_t)(100 + 8 * l)}, st));
    }

    // ---- decoder layer 0 (transformer.py:274-295 with tgt = 0) + final norm + head
    const DecW& d = ctx->dec[0];
    const int M = B * Q;
    const bool gen = dropout_p > 0.f;      // general decoder self-attention path (dropout makes it batch dependent)
    if (!gen) {
        // tgt = 0: self-attention output is out_proj(b_v) + b_o for every query
        CHK(lin_fwd(ctx, d.self_attn.in_b + 2 * D, D, 1, D, d.self_attn.out_w, D, d.self_attn.out_b, T.sa_tmp, D, nullptr, 0, st));
        CHK(launch_layernorm(T.sa_tmp, nullptr, 0, d.n1w, d.n1b, nullptr, nullptr, T.t1, 1, D, 1e-5f, st, &ctx->err));
        // qin = query_embed + t1 (materialised: it is the x operand of the q weight gradient), q = qin Wq^T + bq
        HIPCHK(hipMemcpyAsync(T.qin, ctx->P("query_embed.weight"), (size_t)Q * D * 4, hipMemcpyDeviceToDevice, st));
        CHK(launch_bcast_add_rows(T.qin, T.t1, Q, D, st));
        CHK(lin_fwd(ctx, T.qin, D, Q, D, d.cross.in_w, D, d.cross.in_b, T.dq, D, nullptr, 0, st));
    } else {
        // with dropout the self-attention is no longer constant: weights softmax(q k^T) of q = k = query_pos are dropped per
        // (batch, head, query, key), the value rows are all b_v, so out[b,i] = b_v * rowsum(dropped weights) per head
        CHK(lin_fwd(ctx, ctx->P("query_embed.weight"), D, Q, D, d.self_attn.in_w, 2 * D, d.self_attn.in_b, T.qkd, 2 * D, nullptr, 0, st));
        AttnArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.Q = T.qkd; sa.q_bs = 0; sa.q_rs = 2 * D;
        sa.K = T.qkd + D; sa.k_bs = 0; sa.k_rs = 2 * D;
        sa.V = d.self_attn.in_b + 2 * D; sa.v_bs = 0; sa.v_rs = 0;
        sa.O = T.sO; sa.o_bs = (int64_t)Q * D; sa.o_rs = D; sa.lse = T.lse_s;
        sa.B = B; sa.H = g.nheads; sa.Nq = Q; sa.Nk = Q; sa.HD = hd; sa.scale = 1.0f / sqrtf((float)hd);
        sa.drop_p = dropout_p; sa.drop_seed = dr_dec.s(0);
        CHK(launch_attention(sa, st, &ctx->err));
        CHK(lin_fwd(ctx, T.sO, D, M, D, d.self_attn.out_w, D, d.self_attn.out_b, T.saB, D, nullptr, 0, st, dropout_p, dr_dec.s(1)));
        CHK(launch_layernorm(T.saB, nullptr, 0, d.n1w, d.n1b, nullptr, nullptr, T.T1B, M, D, 1e-5f, st, &ctx->err));
        GemmArgs qg = G0();
        qg.A = T.T1B; qg.lda = D; qg.M = M; qg.K = D; qg.Bw = d.cross.in_w; qg.ldb = D; qg.N = D; qg.bias = d.cross.in_b;
        qg.C = T.dqB; qg.ldc = D; qg.A_add = ctx->P("query_embed.weight"); qg.ld_add = D; qg.add_mod = Q; qg.add_ncols = D;
        CHK(tgemm(ctx, qg, st));
    }
    {
        GemmArgs kv = G0();
        kv.A = T.mem; kv.lda = D; kv.M = B * N; kv.K = D; kv.Bw = d.cross.in_w + (int64_t)D * D; kv.ldb = D; kv.N = 2 * D;
        kv.bias = d.cross.in_b + D; kv.C = T.KV; kv.ldc = 2 * D;
        kv.A_add = ctx->pos_tokens; kv.ld_add = D; kv.add_mod = N; kv.add_ncols = D;
        CHK(tgemm(ctx, kv, st));
        AttnArgs at;
        memset(&at, 0, sizeof(at));
        at.Q = gen ? T.dqB : T.dq; at.q_bs = gen ? (int64_t)Q * D : 0; at.q_rs = D;
        at.K = T.KV; at.k_bs = (int64_t)N * 2 * D; at.k_rs = 2 * D;
        at.V = T.KV + D; at.v_bs = at.k_bs; at.v_rs = 2 * D;
        at.O = T.Oc; at.o_bs = (int64_t)Q * D; at.o_rs = D; at.lse = T.lse_c;
        at.B = B; at.H = g.nheads; at.Nq = Q; at.Nk = N; at.HD = hd; at.scale = 1.0f / sqrtf((float)hd);
        at.ws = ctx->attn_ws; at.ws_floats = ctx->attn_ws_floats;
        at.prec = ctx->gemm_prec;
    at.prec = ctx->gemm_prec;
        at.drop_p = dropout_p; at.drop_seed = dr_dec.s(4);
        CHK(launch_attention(at, st, &ctx->err));
    }
    {
        GemmArgs op = G0();
        op.A = T.Oc; op.lda = D; op.M = M; op.K = D; op.Bw = d.cross.out_w; op.ldb = D; op.N = D; op.bias = d.cross.out_b;
        op.C = T.Y2pre; op.ldc = D; op.res = gen ? T.T1B : T.t1; op.ldres = D; op.res_mod = gen ? 0 : 1;
        op.drop_p = dropout_p; op.drop_seed = dr_dec.s(5);
        CHK(tgemm(ctx, op, st));
    }
    CHK(launch_layernorm(T.Y2pre, nullptr, 0, d.n2w, d.n2b, nullptr, nullptr, T.T2, M, D, 1e-5f, st, &ctx->err));
    CHK(lin_fwd(ctx, T.T2, D, M, D, d.l1w, F, d.l1b, T.Hd, F, nullptr, 1, st, dropout_p, dr_dec.s(2)));
    CHK(lin_fwd(ctx, T.Hd, F, M, F, d.l2w, D, d.l2b, T.Y3pre, D, T.T2, 0, st, dropout_p, dr_dec.s(3)));
    CHK(launch_layernorm(T.Y3pre, nullptr, 0, d.n3w, d.n3b, nullptr, nullptr, T.T3, M, D, 1e-5f, st, &ctx->err));
    CHK(launch_layernorm(T.T3, nullptr, 0, ctx->P("transformer.decoder.norm.weight"), ctx->P("transformer.decoder.norm.bias"),
                         nullptr, nullptr, T.hs, M, D, 1e-5f, st, &ctx->err));
    CHK(lin_fwd(ctx, T.hs, D, M, D, ctx->P("action_head.weight"), A, ctx->P("action_head.bias"), T.a_hat, A, nullptr, 0, st));
    if (a_hat_out) HIPCHK(hipMemcpyAsync(a_hat_out, T.a_hat, (size_t)M * A * 4, hipMemcpyDeviceToDevice, st));
    CHK(launch_losses(T.a_hat, T.actions, T.is_pad, (g.has_cvae_encoder && !g.vq) ? T.latent_info : nullptr, T.losses, B, Q, A, L,
                      g.kl_weight, st));
    if (losses) HIPCHK(hipMemcpyAsync(losses, T.losses, 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
    CHK(launch_check_finite(T.losses, 3, ctx->flags, ACTMI_FLAG_LOSS, st));       // default-on: a loss that is not finite
    T.have_forward = true;
    return 0;
}

// ----------------------------------------------------------------------------------------------------------------------
// backward
// ----------------------------------------------------------------------------------------------------------------------

int train_backward(actmi_ctx* ctx, float loss_scale, hipStream_t st) {
    ctx->err.clear();
    if (!ctx->train || !ctx->train->have_forward) { ctx->err = "backward before forward_train"; return ACTMI_E_STATE; }
    PrecScope prec_scope(ctx);
    TrainState& T = *ctx->train;
    const actmi_config& g = ctx->cfg;
    const int B = T.B, C = g.num_cams, D = g.hidden_dim, F = g.dim_feedforward, Q = g.num_queries, N = ctx->N,
              w0 = g.base_width, L = g.latent_dim, A = g.action_dim, S = g.state_dim, hd = D / g.nheads, H = g.nheads;
    auto Gp = [&](const float* p) { return T.gbase + (p - ctx->pbase); };
    auto GP = [&](const char* key) { return T.gbase + (ctx->P(key) - ctx->pbase); };
    const int M = B * Q;
    const DecW& d = ctx->dec[0];

    // ---- optional global loss scale (f16x3, ACTMI_LOSS_SCALE_LOG2; tuning aid): the seed is multiplied by a power of two
    // and the finished gradient arena multiplied back (both exact); gradients already in the arena (accumulation
    // without zero_grad) are carried through the same scale.  Off by default -- see dyn_scale.
    T.amax_key_ptr = nullptr;
    T.amax_pre_ptr = nullptr;
    float LS = 1.f;
    if (ctx->gemm_prec == ACTMI_PREC_F16X3) {
        static const char* ls_env = getenv("ACTMI_LOSS_SCALE_LOG2");
        int e = ls_env ? atoi(ls_env) : 0;         // off by default: operands are scaled individually (dyn_scale)
        if (e < 0) e = 0;
        if (e > 24) e = 24;
        LS = ldexpf(1.f, e);
    }
    if (LS != 1.f && T.grads_dirty) CHK(launch_scale(T.gbase, ctx->ptotal, LS, st));
    loss_scale *= LS;
    // ---- loss -> a_hat -> heads
    float* d_ahat = T.gQKV;                 // scratch [M][A]
    CHK(launch_l1_bwd(T.a_hat, T.actions, T.is_pad, d_ahat, B, Q, A, loss_scale, st));
    float* dhs = T.gA;
    CHK(lin_dgrad(ctx, d_ahat, A, M, A, ctx->P("action_head.weight"), D, dhs, D, nullptr, nullptr, st));
    CHK(lin_wgrad(ctx, d_ahat, A, M, A, T.hs, D, D, nullptr, 0, GP("action_head.weight"), GP("action_head.bias"), st));
    // decoder.norm, norm3
    float* dT3 = T.gC;
    CHK(ln_bwd_d(ctx, T.T3, ctx->P("transformer.decoder.norm.weight"), dhs, nullptr, dT3, GP("transformer.decoder.norm.weight"),
                      GP("transformer.decoder.norm.bias"), M, D, 1e-5f, st));
    float* dY3 = T.gA;
    CHK(ln_bwd_d(ctx, T.Y3pre, d.n3w, dT3, nullptr, dY3, Gp(d.n3w), Gp(d.n3b), M, D, 1e-5f, st));
    const float dp = T.drop_p;
    const bool gen = dp > 0.f;
    const Drop dr_dec{dp, T.drop_seed, 200};
    const float inv_keep = gen ? 1.f / (1.f - dp) : 1.f;
    // FFN:  Y3pre = T2 + drop3(linear2(Hd)),  Hd = drop(relu(linear1(T2)))
    const float* dz3 = dY3;
    if (gen) { CHK(launch_dropout_bwd(dY3, T.gC, dr_dec.s(3), dp, (int64_t)M * D, st)); dz3 = T.gC; }
    CHK(lin_dgrad(ctx, dz3, D, M, D, d.l2w, F, T.gH, F, nullptr, T.Hd, st, inv_keep));
    CHK(lin_wgrad(ctx, dz3, D, M, D, T.Hd, F, F, nullptr, 0, Gp(d.l2w), Gp(d.l2b), st));
    float* dT2 = T.gC;
    CHK(lin_dgrad(ctx, T.gH, F, M, F, d.l1w, D, dT2, D, dY3, nullptr, st));
    CHK(lin_wgrad(ctx, T.gH, F, M, F, T.T2, D, D, nullptr, 0, Gp(d.l1w), Gp(d.l1b), st));
    // norm2:  Y2pre = t1 + drop2(out_proj(Oc))
    float* dY2 = T.gA;
    CHK(ln_bwd_d(ctx, T.Y2pre, d.n2w, dT2, nullptr, dY2, Gp(d.n2w), Gp(d.n2b), M, D, 1e-5f, st));
    const float* dz2 = dY2;
    if (gen) { CHK(launch_dropout_bwd(dY2, T.gC, dr_dec.s(5), dp, (int64_t)M * D, st)); dz2 = T.gC; }
    float* dOc = T.gH;                      // [M][D] view of the big scratch
    CHK(lin_dgrad(ctx, dz2, D, M, D, d.cross.out_w, D, dOc, D, nullptr, nullptr, st));
    CHK(lin_wgrad(ctx, dz2, D, M, D, T.Oc, D, D, nullptr, 0, Gp(d.cross.out_w), Gp(d.cross.out_b), st));
    float* dt1 = T.tmpD;                    // [D]  (constant path: t1 is one broadcast row)
    HIPCHK(hipMemsetAsync(T.tmpD, 0, 4 * D * sizeof(float), st));
    if (!gen) CHK(colsum_d(ctx, dY2, D, dt1, M, D, st));
    else HIPCHK(hipMemcpyAsync(T.gT1, dY2, (size_t)M * D * 4, hipMemcpyDeviceToDevice, st));     // residual branch: dT1 = dY2
    // cross attention
    float* dKV = T.gQKV;                    // [B*N][2D]
    {
        AttnBwd t;
        memset(&t, 0, sizeof(t));
        t.Q = gen ? T.dqB : T.dq; t.q_bs = gen ? (int64_t)Q * D : 0; t.q_rs = D;
        t.K = T.KV; t.k_bs = (int64_t)N * 2 * D; t.k_rs = 2 * D;
        t.V = T.KV + D; t.v_bs = t.k_bs; t.v_rs = 2 * D;
        t.O = T.Oc; t.dO = dOc; t.lse = T.lse_c;
        t.dQ = T.dqb; t.dq_bs = (int64_t)Q * D; t.dq_rs = D;
        t.dK = dKV; t.dk_bs = (int64_t)N * 2 * D; t.dk_rs = 2 * D;
        t.dV = dKV + D; t.dv_bs = t.dk_bs; t.dv_rs = 2 * D;
        t.B = B; t.H = H; t.Nq = Q; t.Nk = N; t.HD = hd;
        t.drop_p = dp; t.drop_seed = dr_dec.s(4);
        CHK(attn_bwd(ctx, t, st));
    }
    if (!gen) {
        // q = (query_embed + t1) Wq^T + bq  (shared over the batch)
        float* ddq = T.gA;                      // [Q][D]
        CHK(launch_sum_batch(T.dqb, (int64_t)Q * D, D, ddq, B, Q, D, 0, st));
        float* dqin = T.gC;                     // [Q][D]
        CHK(lin_dgrad(ctx, ddq, D, Q, D, d.cross.in_w, D, dqin, D, nullptr, nullptr, st));
        CHK(lin_wgrad(ctx, ddq, D, Q, D, T.qin, D, D, nullptr, 0, Gp(d.cross.in_w), Gp(d.cross.in_b), st));
        CHK(launch_axpy(GP("query_embed.weight"), dqin, (int64_t)Q * D, st));
        CHK(colsum_d(ctx, dqin, D, dt1, Q, D, st));
        // t1 = norm1(out_proj(b_v) + b_o)
        float* dsa = T.tmpD + D;
        CHK(ln_bwd_d(ctx, T.sa_tmp, d.n1w, dt1, nullptr, dsa, Gp(d.n1w), Gp(d.n1b), 1, D, 1e-5f, st));
        float* dbv = T.tmpD + 2 * D;
        CHK(lin_dgrad(ctx, dsa, D, 1, D, d.self_attn.out_w, D, dbv, D, nullptr, nullptr, st));
        CHK(lin_wgrad(ctx, dsa, D, 1, D, d.self_attn.in_b + 2 * D, D, D, nullptr, 0, Gp(d.self_attn.out_w), Gp(d.self_attn.out_b), st));
        CHK(launch_axpy(Gp(d.self_attn.in_b) + 2 * D, dbv, D, st));
    } else {
        // q[b] = (T1[b] + query_embed) Wq^T + bq
        float* dqin = T.gA;                     // [M][D]
        CHK(lin_dgrad(ctx, T.dqb, D, M, D, d.cross.in_w, D, dqin, D, nullptr, nullptr, st));
        CHK(lin_wgrad(ctx, T.dqb, D, M, D, T.T1B, D, D, ctx->P("query_embed.weight"), Q, Gp(d.cross.in_w), Gp(d.cross.in_b), st));
        CHK(launch_sum_batch(dqin, (int64_t)Q * D, D, GP("query_embed.weight"), B, Q, D, 1, st));
        CHK(launch_axpy(T.gT1, dqin, (int64_t)M * D, st));
        // T1 = norm1(drop1(out_proj(sO)))
        CHK(ln_bwd_d(ctx, T.saB, d.n1w, T.gT1, nullptr, T.dsaB, Gp(d.n1w), Gp(d.n1b), M, D, 1e-5f, st));
        CHK(launch_dropout_bwd(T.dsaB, T.gT1, dr_dec.s(1), dp, (int64_t)M * D, st));            // gT1 := d(out_proj output)
        float* dsO = T.gA;
        CHK(lin_dgrad(ctx, T.gT1, D, M, D, d.self_attn.out_w, D, dsO, D, nullptr, nullptr, st));
        CHK(lin_wgrad(ctx, T.gT1, D, M, D, T.sO, D, D, nullptr, 0, Gp(d.self_attn.out_w), Gp(d.self_attn.out_b), st));
        // self-attention: q = k = query_pos (shared over the batch), every value row = b_v
        AttnBwd t;
        memset(&t, 0, sizeof(t));
        t.Q = T.qkd; t.q_bs = 0; t.q_rs = 2 * D;
        t.K = T.qkd + D; t.k_bs = 0; t.k_rs = 2 * D;
        t.V = d.self_attn.in_b + 2 * D; t.v_bs = 0; t.v_rs = 0;
        t.O = T.sO; t.dO = dsO; t.lse = T.lse_s;
        t.dQ = T.dqkB; t.dq_bs = (int64_t)Q * 2 * D; t.dq_rs = 2 * D;
        t.dK = T.dqkB + D; t.dk_bs = t.dq_bs; t.dk_rs = 2 * D;
        t.dV = T.dvB; t.dv_bs = (int64_t)Q * D; t.dv_rs = D;
        t.B = B; t.H = H; t.Nq = Q; t.Nk = Q; t.HD = hd;
        t.drop_p = dp; t.drop_seed = dr_dec.s(0);
        CHK(attn_bwd(ctx, t, st));
        CHK(colsum_d(ctx, T.dvB, D, Gp(d.self_attn.in_b) + 2 * D, M, D, st));                     // d b_v (sum over keys and batch)
        CHK(launch_sum_batch(T.dqkB, (int64_t)Q * 2 * D, 2 * D, T.dqk_d, B, Q, 2 * D, 0, st));    // q/k are shared over the batch
        CHK(lin_dgrad(ctx, T.dqk_d, 2 * D, Q, 2 * D, d.self_attn.in_w, D, T.tmpQD, D, nullptr, nullptr, st));
        CHK(launch_axpy(GP("query_embed.weight"), T.tmpQD, (int64_t)Q * D, st));
        CHK(lin_wgrad(ctx, T.dqk_d, 2 * D, Q, 2 * D, ctx->P("query_embed.weight"), D, D, nullptr, 0, Gp(d.self_attn.in_w),
                      Gp(d.self_attn.in_b), st));
    }
    // k = (memory + pos) Wk^T, v = memory Wv^T
    float* dmem = T.gB;
    CHK(lin_dgrad(ctx, dKV, 2 * D, B * N, 2 * D, d.cross.in_w + (int64_t)D * D, D, dmem, D, nullptr, nullptr, st));
    CHK(lin_wgrad(ctx, dKV, 2 * D, B * N, D, T.mem, D, D, ctx->pos_tokens, N, Gp(d.cross.in_w) + (int64_t)D * D, nullptr, st));
    CHK(lin_wgrad(ctx, dKV + D, 2 * D, B * N, D, T.mem, D, D, nullptr, 0, Gp(d.cross.in_w) + (int64_t)2 * D * D, nullptr, st));
    CHK(colsum_d(ctx, dKV, 2 * D, Gp(d.cross.in_b) + D, B * N, 2 * D, st));
    float* dpos2 = GP("additional_pos_embed.weight");
    {
        GemmArgs a = G0();
        a.A = dKV; a.lda = 2 * D; a.a_rowmap = T.pos_rows; a.M = 2 * B; a.K = D; a.Bw = d.cross.in_w + (int64_t)D * D; a.ldb = D;
        a.tb = 1; a.N = D; a.C = T.tmp2BD; a.ldc = D;
        a.b_scale = ctx->bwd_wscale;
        a.a_scale_dev = dyn_scale(ctx, dKV, 2 * D, B * N, D, st);
        CHK(tgemm(ctx, a, st));
        CHK(launch_sum_batch(T.tmp2BD, 2 * D, D, dpos2, B, 2, D, 1, st));
    }
    // ---- encoder layers, last to first.  dOut lives in gB; each layer returns its dIn in gB again.
    for (int l = g.enc_layers - 1; l >= 0; --l)
        CHK(enc_bwd(ctx, ctx->enc[l], T.en[l], T.gB, T.gB, ctx->pos_tokens, B, N, nullptr, dpos2,
                    Drop{dp, T.drop_seed, (uint32_t)(100 + 8 * l)}, st));
    // every gradient of transformer.* (decoder + main encoder: the head of the arena) is final here, while the backbone and
    // CVAE-encoder backward are still to be enqueued: a data-parallel caller lets its collective stream wait for this event
    // and reduces that range under the rest of the backward (actmi_wait_grad_phase)
    if (T.ev_phase1) HIPCHK(hipEventRecord(T.ev_phase1, st));
    float* dX = T.gB;                       // grad wrt the token matrix [B][N][D]
    // token 1: proprio = W_s qpos + b_s
    CHK(launch_small_linear_wgrad(dX + D, (int64_t)N * D, T.qpos, S, GP("input_proj_robot_state.weight"), B, D, S, st));
    CHK(colsum_d(ctx, dX + D, (int64_t)N * D, GP("input_proj_robot_state.bias"), B, D, st));
    // token 0: latent_input = W_lo z + b_lo
    if (g.has_cvae_encoder) {
        const int Lz = g.vq ? g.vq_class * g.vq_dim : L;
        CHK(launch_small_linear_wgrad(dX, (int64_t)N * D, T.z, Lz, GP("latent_out_proj.weight"), B, D, Lz, st));
        CHK(lin_dgrad(ctx, dX, (int64_t)N * D, B, D, ctx->P("latent_out_proj.weight"), Lz, T.dz, Lz, nullptr, nullptr, st));
    }
    CHK(colsum_d(ctx, dX, (int64_t)N * D, GP("latent_out_proj.bias"), B, D, st));
    // tokens 2..: input_proj (1x1 conv) of the layer4 maps
    const int MP = C * B * ctx->P_;
    CHK(launch_gather_rows(dX, ctx->rowmap, T.dXg, MP, D, st));
    const BlockSave& last = T.blocks.back();
    float* gcur = T.gbuf[0];                // grad wrt the current block output
    CHK(lin_dgrad(ctx, T.dXg, D, MP, D, ctx->P("input_proj.weight"), 8 * w0, gcur, 8 * w0, nullptr, nullptr, st));
    CHK(lin_wgrad(ctx, T.dXg, D, MP, D, last.out, 8 * w0, 8 * w0, nullptr, 0, GP("input_proj.weight"), GP("input_proj.bias"), st));

    // ---- backbone: BasicBlocks in reverse.  Frozen BN: y = conv * scale + bias  =>  dconv = dy * scale.
    for (size_t li = 0; li < ctx->convs.size(); ++li) {
        const ConvLayer& cl = ctx->convs[li];
        HIPCHK(hipMemsetAsync(T.conv_gw[li], 0, (size_t)C * cl.cout * cl.K * 4, st));
        const bool direct = dgrad_direct(ctx, cl) && T.conv_wd16[li];
        CHK(launch_repack_dgrad_w(cl.w, T.conv_wd[li], C, cl.cout, cl.cin, cl.k * cl.k, st, direct ? 1 : 0));
        if (direct) CHK(launch_split16(T.conv_wd[li], T.conv_wd16[li], (int64_t)C * cl.cin * cl.k * cl.k * cl.cout, ctx->bwd_wscale, st));
    }
    HIPCHK(hipMemsetAsync(T.conv1_gw, 0, (size_t)C * w0 * 196 * 4, st));
    float* dz = T.gbuf[1];
    float* dzs = T.gbuf[2];
    float* dsc = T.gbuf[3];
    for (int bi = (int)T.blocks.size() - 1; bi >= 0; --bi) {
        const BlockSave& bs = T.blocks[bi];
        const ConvLayer& k1 = ctx->convs[bs.c1];
        const ConvLayer& k2 = ctx->convs[bs.c2];
        const float* x = bi > 0 ? T.blocks[bi - 1].out : T.pool;
        const int64_t per = (int64_t)B * k2.Ho * k2.Wo * k2.cout;
        // dz = dout * (out > 0) ; dzs = dz * scale_bn2 ; (downsample) dsc = dz * scale_ds
        CHK(launch_relu_bn_bwd(gcur, nullptr, bs.out, k2.scale, dz, dzs, C, per, k2.cout, st, amax_pre(ctx, dzs, st)));
        CHK(conv_wgrad(ctx, k2, bs.c2, dzs, bs.y1, B, st));
        // d(pre-bn1) = dgrad_conv2(dzs) * (y1 > 0) * scale_bn1
        CHK(conv_dgrad(ctx, k2, bs.c2, dzs, dsc, nullptr, bs.y1, k1.scale, B, st, true));
        CHK(conv_wgrad(ctx, k1, bs.c1, dsc, x, B, st));
        // dx = dgrad_conv1(dsc) + identity path
        float* dx = gcur;                     // gcur (dout) is dead after relu_bn_bwd
        if (bs.ds < 0) {
            CHK(conv_dgrad(ctx, k1, bs.c1, dsc, dx, dz, nullptr, nullptr, B, st));
        } else {
            const ConvLayer& ds = ctx->convs[bs.ds];
            CHK(conv_dgrad(ctx, k1, bs.c1, dsc, dx, nullptr, nullptr, nullptr, B, st));
            CHK(launch_relu_bn_bwd(dz, nullptr, nullptr, ds.scale, nullptr, dzs, C, per, ds.cout, st, amax_pre(ctx, dzs, st)));   // dzs = dz * scale_ds
            CHK(conv_wgrad(ctx, ds, bs.ds, dzs, x, B, st));
            CHK(conv_dgrad(ctx, ds, bs.ds, dzs, dx, dx, nullptr, nullptr, B, st));
        }
    }
    // stem: maxpool, relu, bn1, conv1 weight gradient through the NHWC4 normalised image
    // (ReLU + FrozenBN backward of the stem and the operand-scale maximum ride on the pool's backward: one pass over the map)
    CHK(launch_maxpool_bwd_idx(T.pool_arg, gcur, T.g_act1, C * B, ctx->H1, ctx->W1, w0, ctx->H2, ctx->W2, st, ctx->act1,
                               ctx->conv1_scale, B, amax_pre(ctx, T.g_act1, st)));
    static const bool stem_direct = !(getenv("ACTMI_WGRAD_DIRECT") && getenv("ACTMI_WGRAD_DIRECT")[0] == '0');
    if (stem_direct && ctx->gemm_prec == ACTMI_PREC_F16X3 && w0 == 64 && T.det_ws && ctx->H1 == (g.image_h - 1) / 2 + 1 &&
        ctx->W1 == (g.image_w - 1) / 2 + 1 && T.det_ws_floats >= (int64_t)C * 64 * 196) {
        // the direct kernel (wgrad7.hip) + fixed-order sum of its per-workgroup partials into the packed gradient
        const float* sc = dyn_scale(ctx, T.g_act1, w0, C * B * ctx->H1 * ctx->W1, w0, st);
        int nwg = 0;
        if (launch_wgrad7x7s2(T.g_act1, T.xn4, T.det_ws, T.det_ws_floats, sc, C, B, g.image_h, g.image_w, ctx->H1, ctx->W1, &nwg, st) != 0) {
            ctx->err = "wgrad7x7s2 launch failed";
            return ACTMI_E_LAUNCH;
        }
        SplitCombineArgs c{};
        const int64_t slice = (int64_t)64 * 196;
        c.part = T.det_ws; c.nsplit = nwg; c.split_stride = slice; c.gP = slice * nwg; c.ldp = 196;
        c.res = T.conv1_gw; c.ldres = 196; c.gRes = slice;
        c.C = T.conv1_gw; c.ldc = 196; c.gC = slice; c.M = 64; c.N = 196; c.groups = C;
        if (launch_splitk_combine(c, st) != 0) { ctx->err = "splitk combine launch failed"; return ACTMI_E_LAUNCH; }
    } else
    {
        GemmArgs a = G0();
        a.A = T.g_act1; a.lda = w0; a.ta = 1; a.M = w0; a.K = B * ctx->H1 * ctx->W1;
        a.Bw = T.xn4; a.tb = 2; a.N = 196; a.H = g.image_h; a.W = g.image_w; a.Cin = 4; a.KH = a.KW = 7; a.stride = 2; a.pad = 3;
        a.Ho = ctx->H1; a.Wo = ctx->W1; a.img_stride = (int64_t)g.image_h * g.image_w * 4;
        a.C = T.conv1_gw; a.ldc = 196; a.groups = C;
        a.gA = (int64_t)B * ctx->H1 * ctx->W1 * w0; a.gB = (int64_t)B * g.image_h * g.image_w * 4; a.gC = (int64_t)w0 * 196;
        a.splitk = pick_splitk(w0, 196, C, a.K);
        if (a.splitk <= 1) { a.splitk = 0; a.res = a.C; a.ldres = 196; a.gRes = a.gC; }
        a.a_scale_dev = dyn_scale(ctx, T.g_act1, w0, C * B * ctx->H1 * ctx->W1, w0, st);
        CHK(tgemm(ctx, a, st));
    }
    // packed conv gradients -> OIHW state_dict gradients (one launch per layer over the cameras: same-named parameters of
    // consecutive backbones are a constant stride apart in the arena)
    {
        const std::string p0 = "backbones.0.0.body.", p1 = "backbones." + std::to_string(C > 1 ? 1 : 0) + ".0.body.";
        const int64_t cam_stride = ctx->P(p1 + "conv1.weight") - ctx->P(p0 + "conv1.weight");
        CHK(launch_unpack_wgrad(T.conv1_gw, GP((p0 + "conv1.weight").c_str()), w0, 3, 7, 7, 196, 4, st, C, (int64_t)w0 * 196, cam_stride));
        for (size_t li = 0; li < ctx->convs.size(); ++li) {
            const ConvLayer& cl = ctx->convs[li];
            CHK(launch_unpack_wgrad(T.conv_gw[li], GP((p0 + cl.name + ".weight").c_str()), cl.cout, cl.cin, cl.k, cl.k, cl.K, cl.cin, st,
                                    C, (int64_t)cl.cout * cl.K, cam_stride));
        }
    }

    // ---- CVAE encoder
    if (g.has_cvae_encoder) {
        const int n = Q + 2;
        // d latent_info from the reparametrisation and the KL term
        const int Lp = g.vq ? g.vq_class * g.vq_dim : 2 * L;
        if (g.vq) CHK(launch_vq_bwd(T.vq_probs, T.dz, T.d_latent_info, B, g.vq_class, g.vq_dim, st));   // straight-through -> softmax
        else CHK(launch_reparam_kl_bwd(T.latent_info, T.eps, T.dz, T.d_latent_info, B, L, g.kl_weight * loss_scale, st));
        float* dcv = T.gB;                   // grad wrt the CVAE encoder output [B][n][D]: only the CLS rows are non-zero
        HIPCHK(hipMemsetAsync(dcv, 0, (size_t)B * n * D * 4, st));
        CHK(lin_dgrad(ctx, T.d_latent_info, Lp, B, Lp, ctx->P("latent_proj.weight"), D, dcv, (int64_t)n * D, nullptr, nullptr, st));
        CHK(lin_wgrad(ctx, T.d_latent_info, Lp, B, Lp, T.cv_out, (int64_t)n * D, D, nullptr, 0, GP("latent_proj.weight"),
                      GP("latent_proj.bias"), st));
        for (int l = g.enc_layers - 1; l >= 0; --l)
            CHK(enc_bwd(ctx, ctx->cvae[l], T.cv[l], T.gB, T.gB, ctx->P("pos_table"), B, n, T.ckpm, nullptr,
                        Drop{dp, T.drop_seed, (uint32_t)(8 * l)}, st));
        float* dXc = T.gB;
        CHK(launch_sum_batch(dXc, (int64_t)n * D, D, GP("cls_embed.weight"), B, 1, D, 1, st));
        CHK(launch_small_linear_wgrad(dXc + D, (int64_t)n * D, T.qpos, S, GP("encoder_joint_proj.weight"), B, D, S, st));
        CHK(colsum_d(ctx, dXc + D, (int64_t)n * D, GP("encoder_joint_proj.bias"), B, D, st));
        CHK(launch_gather_rows(dXc, T.cmap, T.dXg, B * Q, D, st));
        CHK(lin_wgrad(ctx, T.dXg, D, B * Q, D, T.actions, A, A, nullptr, 0, GP("encoder_action_proj.weight"),
                      GP("encoder_action_proj.bias"), st));
    }
    if (LS != 1.f) CHK(launch_scale(T.gbase, ctx->ptotal, 1.f / LS, st));
    T.grads_dirty = true;
    T.have_forward = false;
    return 0;
}

int train_zero_grad(actmi_ctx* ctx, hipStream_t st) {
    if (!ctx->train) { ctx->err = "handle was created without enable_training"; return ACTMI_E_STATE; }
    HIPCHK(hipMemsetAsync(ctx->train->gbase, 0, (size_t)ctx->ptotal * 4, st));
    ctx->train->grads_dirty = false;
    return 0;
}

int train_adamw_step(actmi_ctx* ctx, float lr, float lr_backbone, float wd, float b1, float b2, float eps, int64_t step,
                     hipStream_t st) {
    const int rc = train_adamw_range(ctx, lr, lr_backbone, wd, b1, b2, eps, step, 0, ctx->ptotal, st);
    if (rc != 0) return rc;
    return engine_prepare_weights(ctx, st, true); // conv repack, decoder constants, learned pos rows follow the new weights
}

// the update alone on the arena range [offset, offset + count) (64-float aligned): the sharded optimizer of data-parallel
// training runs it on the slices a rank owns; engine_prepare_weights follows once the updated parameters were all-gathered
int train_adamw_range(actmi_ctx* ctx, float lr, float lr_backbone, float wd, float b1, float b2, float eps, int64_t step,
                      int64_t offset, int64_t count, hipStream_t st) {
    if (!ctx->train) { ctx->err = "handle was created without enable_training"; return ACTMI_E_STATE; }
    if (offset < 0 || count < 0 || (offset & 63) || offset + count > ctx->ptotal) { ctx->err = "adamw range must be 64-float aligned and inside the arena"; return ACTMI_E_INVALID; }
    if (count == 0) return 0;
    TrainState& T = *ctx->train;
    // gated on the device by the handle's flag word: after a non-finite loss or a weight beyond its split scale the
    // update is skipped until the host has read and cleared the word (actmi_get_flags) -- weights and moments stay intact
    CHK(launch_adamw(ctx->pbase + offset, T.gbase + offset, T.mbase + offset, T.vbase + offset, T.group + (offset >> 6), count, lr,
                     lr_backbone, wd, b1, b2, eps, step, st, ctx->flags, ACTMI_FLAG_LOSS | ACTMI_FLAG_WEIGHT));
    return 0;
}
