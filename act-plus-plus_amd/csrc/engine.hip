// Host side of libactmi: parameter store in the reference's state_dict layout, weight preparation, and the
// inference graph of one ACT policy query (reference detr_vae.py:163-254,
// transformer.py:49-122, 211-224, 274-295).  No allocation and no host synchronisation on the forward path:
// every buffer is sized for cfg.max_batch at create time and all work is enqueued on the caller's stream,
// so the whole forward can be captured in a hipGraph by the caller.
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace {

thread_local std::string g_create_error;

#define HIPCHK(expr)                                                                            \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(_e);                       \
            return ACTMI_E_LAUNCH;                                                              \
        }                                                                                       \
    } while (0)

#define CHK(expr)                                                                               \
    do {                                                                                        \
        int _rc = (expr);                                                                       \
        if (_rc != 0) {                                                                         \
            if (ctx->err.empty()) ctx->err = std::string("failed: ") + #expr;                   \
            return _rc < -5 ? ACTMI_E_LAUNCH : _rc;                                             \
        }                                                                                       \
    } while (0)

int conv_out(int x, int k, int s, int p) { return (x + 2 * p - k) / s + 1; }

void add_param(actmi_ctx* c, const std::string& key, std::vector<int64_t> shape, bool is_buffer) {
    Param p;
    p.key = key;
    p.shape = shape;
    p.numel = 1;
    for (auto d : shape) p.numel *= d;
    p.off = c->ptotal;
    p.is_buffer = is_buffer;
    c->ptotal += (p.numel + 63) & ~int64_t(63);   // 256-byte aligned slots
    c->index[key] = (int)c->params.size();
    c->params.push_back(p);
}

void add_mha(actmi_ctx* c, const std::string& p, int D) {
    add_param(c, p + "in_proj_weight", {3 * D, D}, false);
    add_param(c, p + "in_proj_bias", {3 * D}, false);
    add_param(c, p + "out_proj.weight", {D, D}, false);
    add_param(c, p + "out_proj.bias", {D}, false);
}

void add_ffn_norms(actmi_ctx* c, const std::string& p, int D, int F, int nnorm) {
    add_param(c, p + "linear1.weight", {F, D}, false);
    add_param(c, p + "linear1.bias", {F}, false);
    add_param(c, p + "linear2.weight", {D, F}, false);
    add_param(c, p + "linear2.bias", {D}, false);
    for (int i = 1; i <= nnorm; ++i) {
        add_param(c, p + "norm" + std::to_string(i) + ".weight", {D}, false);
        add_param(c, p + "norm" + std::to_string(i) + ".bias", {D}, false);
    }
}

void add_fbn(actmi_ctx* c, const std::string& p, int n) {
    for (const char* s : {"weight", "bias", "running_mean", "running_var"}) add_param(c, p + s, {n}, true);
}

// state_dict spec in the registration order of the reference (detr_vae.py:49-105); mirrors actmi/weights.py
void build_spec(actmi_ctx* c) {
    const actmi_config& g = c->cfg;
    const int D = g.hidden_dim, F = g.dim_feedforward, Q = g.num_queries, S = g.state_dim, A = g.action_dim,
              L = g.latent_dim, w0 = g.base_width;
    add_param(c, "pos_table", {1, Q + 2, D}, true);
    for (int i = 0; i < g.enc_layers; ++i) {
        std::string p = "transformer.encoder.layers." + std::to_string(i) + ".";
        add_mha(c, p + "self_attn.", D);
        add_ffn_norms(c, p, D, F, 2);
    }
    for (int i = 0; i < g.dec_layers; ++i) {
        std::string p = "transformer.decoder.layers." + std::to_string(i) + ".";
        add_mha(c, p + "self_attn.", D);
        add_mha(c, p + "multihead_attn.", D);
        add_ffn_norms(c, p, D, F, 3);
    }
    add_param(c, "transformer.decoder.norm.weight", {D}, false);
    add_param(c, "transformer.decoder.norm.bias", {D}, false);
    if (g.has_cvae_encoder) {
        for (int i = 0; i < g.enc_layers; ++i) {
            std::string p = "encoder.layers." + std::to_string(i) + ".";
            add_mha(c, p + "self_attn.", D);
            add_ffn_norms(c, p, D, F, 2);
        }
    }
    add_param(c, "action_head.weight", {A, D}, false);
    add_param(c, "action_head.bias", {A}, false);
    add_param(c, "is_pad_head.weight", {1, D}, false);
    add_param(c, "is_pad_head.bias", {1}, false);
    add_param(c, "query_embed.weight", {Q, D}, false);
    add_param(c, "input_proj.weight", {D, 8 * w0, 1, 1}, false);
    add_param(c, "input_proj.bias", {D}, false);
    for (int cam = 0; cam < g.num_cams; ++cam) {
        std::string p = "backbones." + std::to_string(cam) + ".0.body.";
        add_param(c, p + "conv1.weight", {w0, 3, 7, 7}, false);
        add_fbn(c, p + "bn1.", w0);
        int cin = w0;
        for (int li = 1; li <= 4; ++li) {
            const int cout = w0 << (li - 1);
            for (int bi = 0; bi < 2; ++bi) {
                std::string bp = p + "layer" + std::to_string(li) + "." + std::to_string(bi) + ".";
                add_param(c, bp + "conv1.weight", {cout, cin, 3, 3}, false);
                add_fbn(c, bp + "bn1.", cout);
                add_param(c, bp + "conv2.weight", {cout, cout, 3, 3}, false);
                add_fbn(c, bp + "bn2.", cout);
                if (bi == 0 && li > 1) {
                    add_param(c, bp + "downsample.0.weight", {cout, cin, 1, 1}, false);
                    add_fbn(c, bp + "downsample.1.", cout);
                }
                cin = cout;
            }
        }
    }
    add_param(c, "input_proj_robot_state.weight", {D, S}, false);
    add_param(c, "input_proj_robot_state.bias", {D}, false);
    add_param(c, "cls_embed.weight", {1, D}, false);
    add_param(c, "encoder_action_proj.weight", {D, A}, false);
    add_param(c, "encoder_action_proj.bias", {D}, false);
    add_param(c, "encoder_joint_proj.weight", {D, S}, false);
    add_param(c, "encoder_joint_proj.bias", {D}, false);
    // VQ-ACT swaps the Gaussian latent for a [vq_class x vq_dim] code (detr_vae.py:50-60)
    const int Lp = g.vq ? g.vq_class * g.vq_dim : 2 * L, Li = g.vq ? g.vq_class * g.vq_dim : L;
    add_param(c, "latent_proj.weight", {Lp, D}, false);
    add_param(c, "latent_proj.bias", {Lp}, false);
    add_param(c, "latent_out_proj.weight", {D, Li}, false);
    add_param(c, "latent_out_proj.bias", {D}, false);
    add_param(c, "additional_pos_embed.weight", {2, D}, false);
}

int dev_alloc(actmi_ctx* ctx, float** p, int64_t nfloats) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, (size_t)(nfloats > 0 ? nfloats : 1) * sizeof(float));
    if (e != hipSuccess) {
        ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return ACTMI_E_NOMEM;
    }
    ctx->allocs.push_back(q);
    *p = reinterpret_cast<float*>(q);
    return 0;
}

}  // namespace

int ctx_gemm(actmi_ctx* ctx, GemmArgs a, hipStream_t st, int ws_half, LnFuse* ln) {
    if (ln) ln->done = false;
    a.prec = ctx->prec_override ? ctx->prec_override : ctx->gemm_prec;
    // ws_half 0 / 1: this launch belongs to one of two concurrent branches, each with its own half of the slice workspace
    const int64_t ws_part = (ctx->splitk_ws_floats / ctx->nbranch) & ~(int64_t)3;
    float* const ws = ctx->splitk_ws ? ctx->splitk_ws + (ws_half > 0 ? ws_half * ws_part : 0) : nullptr;
    const int64_t ws_floats = ws_half >= 0 ? ws_part : ctx->splitk_ws_floats;
    if (a.prec == ACTMI_PREC_F16X3 && a.tb == 0) {
        // B is a weight matrix: use its pre-split image (same offsets) where one exists
        if (a.Bw >= ctx->pbase && a.Bw < ctx->pbase + ctx->ptotal) {
            a.b_scale = engine_weight_scale(ctx, a.Bw);
            a.Bw = ctx->p16base + (a.Bw - ctx->pbase);
            a.b_split = 1;
        } else {
            for (const ConvLayer& cl : ctx->convs) {
                if (a.Bw >= cl.w && a.Bw < cl.w + (int64_t)ctx->cfg.num_cams * cl.cout * cl.K) {     // (a camera's slice of it)
                    a.Bw = cl.w16 + (a.Bw - cl.w); a.b_split = 1; a.b_scale = cl.w16_scale; a.k_tap_inner = cl.k_tap_inner ? 1 : 0; break;
                }
                if (cl.wf && a.Bw >= cl.wf && a.Bw < cl.wf + (int64_t)ctx->cfg.num_cams * cl.cout * (cl.K + cl.Kx)) {
                    a.Bw = cl.wf16 + (a.Bw - cl.wf); a.b_split = 1; a.b_scale = cl.wf16_scale; a.k_tap_inner = cl.k_tap_inner ? 1 : 0; break;
                }
            }
        }
    }
    // Small grids (B = 1-4 rollouts: layer3/4 convolutions, the K = 3200 FFN products): a launch of a few hundred tiles
    // leaves CUs idle and its lone workgroups latency-bound on a long K loop.  Split the contraction over blockIdx.z into
    // plain slices and let a combine pass sum them in a fixed order and apply the epilogue.  Thresholds from a sweep at
    // B = 1, 2, 4, 8 (tools/splitk_sweep.sh): aim at 1536 64x64-tile equivalents, keep >= 12 K tiles per split, leave
    // launches of >= 768 such tiles alone (B = 8 and the training batch never qualify).
    if (ctx->fwd_splitk && ctx->splitk_ws && a.splitk <= 1 && a.tb == 0 && a.ta == 0 && (a.mode == 0 || a.mode == 1) &&
        !a.rowmap && !a.C2 && !a.mask && a.drop_p == 0.f && a.res_mod == 0 && a.groups_inner == 0 && !a.stamps) {
        const int groups = a.groups > 0 ? a.groups : 1;
        const int64_t tiles = (int64_t)((a.M + 63) / 64) * ((a.N + 63) / 64) * groups * (ws_half >= 0 ? ctx->policy_mult : 1);
        const int nk = (a.K + 31) / 32;
        int S = tiles < ctx->sk_maxtiles ? (int)((ctx->sk_target + tiles - 1) / tiles) : 1;
        // B = 8: the 304-workgroup launches with a very long contraction (layer4's K = 4608 convolutions: 217 us unsplit,
        // 167 us + a 26 us combine pass split four ways; the K <= 3200 shapes do not pay for their combine pass --
        // profiles/r02_splitk_b8_sweep.json)
        if (S == 1 && tiles < ctx->sk_maxtiles_long && nk >= ctx->sk_long_nk) S = (int)((ctx->sk_target_long + tiles - 1) / tiles);
        // a product whose LayerNorm sums the slices itself pays no combine pass: a long contraction on a grid that leaves CUs
        // idle (FFN2: K = 3200 on 304 workgroups at B = 8) is split even where the general rule would not (ACTMI_LN_SPLIT)
        if (ln && S == 1 && ctx->ln_split > 1 && nk >= 64 && tiles < 2 * ctx->sk_maxtiles) S = ctx->ln_split;
        if (ln && S == 1 && ctx->ln_split_short > 1 && nk >= 16 && nk < 64 && tiles < 2 * ctx->sk_maxtiles) S = ctx->ln_split_short;
        if (S > 8) S = 8;
        if (S > nk / ctx->sk_minnk) S = nk / ctx->sk_minnk;
        while (S >= 2 && (S - 1) * ((nk + S - 1) / S) >= nk) --S;              // every split must own a K tile
        const int64_t slice = (int64_t)a.M * a.N;
        if (S >= 2 && slice * groups * S <= ws_floats && (a.N & 3) == 0) {
            GemmArgs p = a;
            p.scale = p.bias = p.res = nullptr;
            p.relu = 0;
            p.C = ws;
            p.ldc = a.N;
            p.gC = slice * S;
            p.splitk = S;
            p.split_stride = slice;
            int rc = launch_gemm(p, st, &ctx->err);
            if (rc) return rc;
            if (ln && groups == 1 && !a.scale && !a.relu && (!a.res || a.ldres == a.N)) {
                // the LayerNorm that follows reads the slices itself: sum in slice order + bias + residual, as the combine does
                rc = launch_layernorm(ws, a.res, 0, ln->w, ln->b, ln->w2, ln->b2, ln->out, a.M, a.N, ln->eps, st, &ctx->err, S, slice,
                                      a.bias, ln->extra);
                ln->done = rc == 0;
                return rc;
            }
            SplitCombineArgs c{};
            c.part = ws; c.nsplit = S; c.split_stride = slice; c.gP = slice * S; c.ldp = a.N;
            c.scale = a.scale; c.bias = a.bias; c.gSB = a.gSB;
            c.res = a.res; c.ldres = a.ldres; c.gRes = a.gRes;
            c.relu = a.relu;
            c.C = a.C; c.ldc = a.ldc; c.gC = a.gC;
            c.M = a.M; c.N = a.N; c.groups = groups;
            rc = launch_splitk_combine(c, st);
            if (rc) ctx->err = "splitk combine launch failed";
            return rc;
        }
    }
    return launch_gemm(a, st, &ctx->err);
}

// scale of the split image of the parameter that contains address w (parameters are laid out in increasing offset order)
float engine_weight_scale(const actmi_ctx* ctx, const float* w) {
    const int64_t off = w - ctx->pbase;
    size_t lo = 0, hi = ctx->params.size();
    while (hi - lo > 1) {
        const size_t mid = (lo + hi) / 2;
        if (ctx->params[mid].off <= off) lo = mid; else hi = mid;
    }
    return lo < ctx->pscale.size() ? ctx->pscale[lo] : W16_SCALE;
}

// Range guard of the f16x3 weight images: a fixed 2^8 assumed |W| < 255 and left the lo pieces of
// weights below ~2e-4 in the fp16 subnormals.  Here every parameter gets the power of two that brings its largest magnitude
// into [2^13, 2^14) -- capped at 2^12 so that ordinary 1e-2 .. 1 weights keep a scale of 2^12 -- measured on the device.  A
// parameter that is not finite, or so large that no representable scale fits, fails the finalize.  One host
// synchronisation; runs at finalize only (training steps keep the scales and raise ACTMI_FLAG_WEIGHT on overflow).
int engine_calibrate_weight_scales(actmi_ctx* ctx, hipStream_t st) {
    const int np = (int)ctx->params.size();
    ctx->pscale.assign(np, W16_SCALE);
    if (ctx->gemm_prec != ACTMI_PREC_F16X3) return 0;
    CHK(launch_seg_amax(ctx->pbase, ctx->poff_dev, ctx->pnumel_dev, np, ctx->pamax_dev, st));
    std::vector<unsigned> bits(np);
    HIPCHK(hipMemcpyAsync(bits.data(), ctx->pamax_dev, np * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    float gmax = 0.f;
    for (int i = 0; i < np; ++i) {
        float amax;
        memcpy(&amax, &bits[i], 4);
        if (!(amax <= 3.0e38f)) {                  // NaN or inf bits
            ctx->err = "parameter " + ctx->params[i].key + " is not finite";
            return ACTMI_E_INVALID;
        }
        float sc = 4096.f;                         // cap: an all-zero / tiny parameter needs no more
        if (amax > 0.f) {
            int e;
            frexpf(amax, &e);                      // amax = m * 2^e, m in [0.5, 1)
            const float fit = ldexpf(1.f, 14 - e); // amax * fit in [2^13, 2^14)
            sc = fit < 4096.f ? fit : 4096.f;
        }
        if (!(sc >= 1.17549435e-38f) || !(amax * sc < 65504.f)) {
            ctx->err = "parameter " + ctx->params[i].key + " is too large for the f16x3 path (use gemm_prec f32)";
            return ACTMI_E_INVALID;
        }
        ctx->pscale[i] = sc;
        if (!ctx->params[i].is_buffer && amax > gmax) gmax = amax;
    }
    HIPCHK(hipMemcpyAsync(ctx->pscale_dev, ctx->pscale.data(), np * sizeof(float), hipMemcpyHostToDevice, st));
    // one scale per convolution layer (its weight image spans the cameras): the smallest of the cameras' scales
    auto key_scale = [&](const std::string& key) {
        auto it = ctx->index.find(key);
        return it == ctx->index.end() ? W16_SCALE : ctx->pscale[it->second];
    };
    ctx->conv1_wscale = 4096.f;
    for (auto& cl : ctx->convs) cl.w16_scale = 4096.f;
    for (int cam = 0; cam < ctx->cfg.num_cams; ++cam) {
        const std::string p = "backbones." + std::to_string(cam) + ".0.body.";
        ctx->conv1_wscale = std::min(ctx->conv1_wscale, key_scale(p + "conv1.weight"));
        for (auto& cl : ctx->convs) cl.w16_scale = std::min(cl.w16_scale, key_scale(p + cl.name + ".weight"));
    }
    // weights split on the fly in the backward GEMMs share one static scale
    ctx->bwd_wscale = W16_SCALE;
    while (gmax * ctx->bwd_wscale >= 16384.f && ctx->bwd_wscale > 1.f / 65536.f) ctx->bwd_wscale *= 0.5f;
    HIPCHK(hipStreamSynchronize(st));
    return 0;
}

namespace {

MhaW mha_w(actmi_ctx* c, const std::string& p) {
    MhaW m;
    m.in_w = c->P(p + "in_proj_weight");
    m.in_b = c->P(p + "in_proj_bias");
    m.out_w = c->P(p + "out_proj.weight");
    m.out_b = c->P(p + "out_proj.bias");
    return m;
}

void resolve_layers(actmi_ctx* c) {
    const actmi_config& g = c->cfg;
    auto enc = [&](const std::string& p) {
        EncW e;
        e.attn = mha_w(c, p + "self_attn.");
        e.l1w = c->P(p + "linear1.weight"); e.l1b = c->P(p + "linear1.bias");
        e.l2w = c->P(p + "linear2.weight"); e.l2b = c->P(p + "linear2.bias");
        e.n1w = c->P(p + "norm1.weight"); e.n1b = c->P(p + "norm1.bias");
        e.n2w = c->P(p + "norm2.weight"); e.n2b = c->P(p + "norm2.bias");
        return e;
    };
    c->enc.clear(); c->cvae.clear(); c->dec.clear();
    for (int i = 0; i < g.enc_layers; ++i) c->enc.push_back(enc("transformer.encoder.layers." + std::to_string(i) + "."));
    if (g.has_cvae_encoder)
        for (int i = 0; i < g.enc_layers; ++i) c->cvae.push_back(enc("encoder.layers." + std::to_string(i) + "."));
    for (int i = 0; i < g.dec_layers; ++i) {
        std::string p = "transformer.decoder.layers." + std::to_string(i) + ".";
        DecW d;
        d.self_attn = mha_w(c, p + "self_attn.");
        d.cross = mha_w(c, p + "multihead_attn.");
        d.l1w = c->P(p + "linear1.weight"); d.l1b = c->P(p + "linear1.bias");
        d.l2w = c->P(p + "linear2.weight"); d.l2b = c->P(p + "linear2.bias");
        d.n1w = c->P(p + "norm1.weight"); d.n1b = c->P(p + "norm1.bias");
        d.n2w = c->P(p + "norm2.weight"); d.n2b = c->P(p + "norm2.bias");
        d.n3w = c->P(p + "norm3.weight"); d.n3b = c->P(p + "norm3.bias");
        c->dec.push_back(d);
    }
}

GemmArgs linear_args(const float* A, int64_t lda, int M, int K, const float* W, int N, const float* bias, float* C,
                     int64_t ldc) {
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.lda = lda; a.M = M; a.K = K; a.N = N; a.Bw = W; a.ldb = K; a.bias = bias; a.C = C; a.ldc = ldc;
    a.groups = 1;
    return a;
}

}  // namespace

float* actmi_ctx::P(const std::string& key) {
    auto it = index.find(key);
    if (it == index.end()) return nullptr;
    return pbase + params[it->second].off;
}

// ------------------------------------------------------------------------------------------------
// create / destroy
// ------------------------------------------------------------------------------------------------

int engine_create(const actmi_config* cfg, actmi_ctx** out) {
    if (!cfg || !out) { g_create_error = "null argument"; return ACTMI_E_INVALID; }
    const actmi_config& g = *cfg;
    if (g.struct_size != (uint32_t)sizeof(actmi_config)) {
        // the first field is readable whatever the caller's struct looks like; nothing else is trusted before this check
        g_create_error = "actmi_config.struct_size is " + std::to_string(g.struct_size) + ", this library expects " +
                         std::to_string(sizeof(actmi_config)) + " (binding built against a different include/actmi.h)";
        return ACTMI_E_INVALID;
    }
    if (g.num_cams < 1 || g.max_batch < 1 || g.hidden_dim % g.nheads || (g.hidden_dim & 3) || (g.dim_feedforward & 3) ||
        (g.base_width & 3) || g.base_width > 64 || g.enc_layers < 1 || g.dec_layers < 1) {
        g_create_error = "unsupported configuration";
        return ACTMI_E_INVALID;
    }
    const int hd = g.hidden_dim / g.nheads;
    if (hd != 16 && hd != 32 && hd != 64) { g_create_error = "head_dim must be 16, 32 or 64"; return ACTMI_E_INVALID; }
    if (g.vq && (g.vq_class < 1 || g.vq_dim < 1 || ((g.vq_class * g.vq_dim) & 3))) {
        g_create_error = "vq needs positive vq_class, vq_dim with vq_class*vq_dim a multiple of 4";
        return ACTMI_E_INVALID;
    }
    actmi_ctx* ctx = new actmi_ctx();
    ctx->cfg = g;
    if (hipGetDevice(&ctx->device) != hipSuccess) { g_create_error = "hipGetDevice failed"; delete ctx; return ACTMI_E_LAUNCH; }
    ctx->ptotal = 0;
    build_spec(ctx);
    // geometry
    ctx->H1 = conv_out(g.image_h, 7, 2, 3); ctx->W1 = conv_out(g.image_w, 7, 2, 3);
    ctx->H2 = conv_out(ctx->H1, 3, 2, 1); ctx->W2 = conv_out(ctx->W1, 3, 2, 1);
    int h = ctx->H2, w = ctx->W2;
    for (int i = 0; i < 3; ++i) { h = conv_out(h, 3, 2, 1); w = conv_out(w, 3, 2, 1); }
    ctx->fh = h; ctx->fw = w;
    ctx->P_ = h * w;
    ctx->N = 2 + g.num_cams * h * w;
    if (h < 1 || w < 1) { g_create_error = "image too small"; delete ctx; return ACTMI_E_INVALID; }

    auto fail = [&](int rc) { g_create_error = ctx->err; engine_destroy(ctx); return rc; };
    int rc;
    if ((rc = dev_alloc(ctx, &ctx->pbase, ctx->ptotal))) return fail(rc);
    if (hipMemset(ctx->pbase, 0, ctx->ptotal * sizeof(float)) != hipSuccess) { ctx->err = "hipMemset failed"; return fail(ACTMI_E_LAUNCH); }
    {
        // forward precision of this handle: fp16-split products unless ACTMI_GEMM_PREC=f32 asks for the native fp32 MFMA
        const char* e = getenv("ACTMI_GEMM_PREC");
        ctx->gemm_prec = (e && e[0] == 'f' && e[1] == '3') ? ACTMI_PREC_F32 : ACTMI_PREC_F16X3;
        { const char* tp = getenv("ACTMI_TRAIN_PREC"); if (tp && tp[0] == 'b') ctx->train_prec = ACTMI_PREC_BF16; }
        if (ctx->ptotal & 3) { ctx->err = "parameter arena not a multiple of 4 floats"; return fail(ACTMI_E_LAUNCH); }
        if ((rc = dev_alloc(ctx, &ctx->p16base, ctx->ptotal))) return fail(rc);
        { const char* vp = getenv("ACTMI_CONV1_VPOOL"); ctx->conv1_vpool = !(vp && vp[0] == '0'); }
        const char* sk = getenv("ACTMI_FWD_SPLITK");
        ctx->fwd_splitk = !(sk && sk[0] == '0');
        if (const char* e2 = getenv("ACTMI_FWD_SPLITK_TARGET")) ctx->sk_target = atoi(e2);
        if (const char* e2 = getenv("ACTMI_FWD_SPLITK_MINNK")) ctx->sk_minnk = atoi(e2) > 0 ? atoi(e2) : 1;
        if (const char* e2 = getenv("ACTMI_FWD_SPLITK_MAXTILES")) ctx->sk_maxtiles = atoi(e2);
        if (const char* e2 = getenv("ACTMI_FWD_SPLITK_LONG_NK")) ctx->sk_long_nk = atoi(e2) > 0 ? atoi(e2) : 1 << 30;
        // slices of split contractions (ctx_gemm checks the fit); 256 MB covers 4-way splits of the B = 8 launches
        ctx->splitk_ws_floats = (int64_t)(getenv("ACTMI_FWD_SPLITK_WS_MB") ? atoi(getenv("ACTMI_FWD_SPLITK_WS_MB")) : 256) << 18;
        if ((rc = dev_alloc(ctx, &ctx->splitk_ws, ctx->splitk_ws_floats))) return fail(rc);
    }
    {
        // tables of the weight range guard: parameter index of every 64-float slot, offsets / sizes for the amax kernel
        const int np = (int)ctx->params.size();
        std::vector<int> seg((size_t)(ctx->ptotal / 64), 0);
        std::vector<int64_t> off(np), numel(np);
        for (int i = 0; i < np; ++i) {
            off[i] = ctx->params[i].off; numel[i] = ctx->params[i].numel;
            const int64_t g0 = ctx->params[i].off / 64, g1 = (i + 1 < np ? ctx->params[i + 1].off : ctx->ptotal) / 64;
            for (int64_t g = g0; g < g1; ++g) seg[(size_t)g] = i;
        }
        float *f0 = nullptr, *f1 = nullptr, *f2 = nullptr, *f3 = nullptr, *f4 = nullptr, *f5 = nullptr;
        if ((rc = dev_alloc(ctx, &f0, (int64_t)seg.size()))) return fail(rc);
        if ((rc = dev_alloc(ctx, &f1, 2 * np))) return fail(rc);
        if ((rc = dev_alloc(ctx, &f2, 2 * np))) return fail(rc);
        if ((rc = dev_alloc(ctx, &f3, np))) return fail(rc);
        if ((rc = dev_alloc(ctx, &f4, np))) return fail(rc);
        if ((rc = dev_alloc(ctx, &f5, 4))) return fail(rc);
        ctx->pseg64 = reinterpret_cast<int*>(f0);
        ctx->poff_dev = reinterpret_cast<int64_t*>(f1);
        ctx->pnumel_dev = reinterpret_cast<int64_t*>(f2);
        ctx->pamax_dev = reinterpret_cast<unsigned*>(f3);
        ctx->pscale_dev = f4;
        ctx->flags = reinterpret_cast<uint32_t*>(f5);
        ctx->pscale.assign(np, W16_SCALE);
        if (hipMemcpy(ctx->pseg64, seg.data(), seg.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(ctx->poff_dev, off.data(), np * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(ctx->pnumel_dev, numel.data(), np * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(ctx->pscale_dev, ctx->pscale.data(), np * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemset(ctx->flags, 0, 16) != hipSuccess) {
            ctx->err = "hipMemcpy failed (weight scale tables)";
            return fail(ACTMI_E_LAUNCH);
        }
    }
    resolve_layers(ctx);

    // ---- conv layer table + packed weights
    const int C = g.num_cams, w0 = g.base_width;
    {
        int cin = w0, H = ctx->H2, W = ctx->W2;
        for (int li = 1; li <= 4; ++li) {
            const int cout = w0 << (li - 1);
            for (int bi = 0; bi < 2; ++bi) {
                const int s = (bi == 0 && li > 1) ? 2 : 1;
                std::string bp = "layer" + std::to_string(li) + "." + std::to_string(bi) + ".";
                ConvLayer c1{bp + "conv1", bp + "bn1.", cin, cout, 3, s, 1, H, W, conv_out(H, 3, s, 1), conv_out(W, 3, s, 1)};
                ConvLayer c2{bp + "conv2", bp + "bn2.", cout, cout, 3, 1, 1, c1.Ho, c1.Wo, c1.Ho, c1.Wo};
                ctx->convs.push_back(c1);
                ctx->convs.push_back(c2);
                if (bi == 0 && li > 1) {
                    ConvLayer ds{bp + "downsample.0", bp + "downsample.1.", cin, cout, 1, s, 0, H, W, c1.Ho, c1.Wo};
                    ctx->convs.push_back(ds);
                }
                cin = cout; H = c1.Ho; W = c1.Wo;
            }
        }
        for (auto& cl : ctx->convs) cl.K = cl.k * cl.k * cl.cin;
        // conv2 of every block with a downsample branch carries the branch in its own contraction (inference, f16x3)
        for (size_t i = 0; i + 2 < ctx->convs.size(); ++i) {
            ConvLayer& k2 = ctx->convs[i + 1];
            const ConvLayer& ds = ctx->convs[i + 2];
            if (ds.k == 1 && ds.stride == 2 && k2.k == 3 && k2.stride == 1 && ds.cout == k2.cout && ds.Ho == k2.Ho && ds.Wo == k2.Wo &&
                ds.name.find("downsample") != std::string::npos && (k2.cin % 32) == 0 && (ds.cin % 32) == 0) {
                k2.ds_index = (int)i + 2;
                k2.Kx = ds.cin;
                if ((rc = dev_alloc(ctx, &k2.wf, (int64_t)C * k2.cout * (k2.K + k2.Kx)))) return fail(rc);
                if ((rc = dev_alloc(ctx, &k2.wf16, (int64_t)C * k2.cout * (k2.K + k2.Kx)))) return fail(rc);
                if ((rc = dev_alloc(ctx, &k2.bias_f, (int64_t)C * k2.cout))) return fail(rc);
            }
        }
        for (auto& cl : ctx->convs) {
            if ((rc = dev_alloc(ctx, &cl.w, (int64_t)C * cl.cout * cl.K))) return fail(rc);
            if ((rc = dev_alloc(ctx, &cl.w16, (int64_t)C * cl.cout * cl.K))) return fail(rc);
            if ((rc = dev_alloc(ctx, &cl.scale, (int64_t)C * cl.cout))) return fail(rc);
            if ((rc = dev_alloc(ctx, &cl.bias, (int64_t)C * cl.cout))) return fail(rc);
        }
    }
    const int D = g.hidden_dim, F = g.dim_feedforward, Q = g.num_queries, N = ctx->N, B = g.max_batch;
    if ((rc = dev_alloc(ctx, &ctx->conv1_w, (int64_t)C * w0 * 148))) return fail(rc);
    if (ctx->gemm_prec == ACTMI_PREC_F16X3 && (rc = dev_alloc(ctx, &ctx->conv1_wimg, (int64_t)C * conv1_wimg_bytes() / 4))) return fail(rc);
    if ((rc = dev_alloc(ctx, &ctx->conv1_scale, (int64_t)C * w0))) return fail(rc);
    if ((rc = dev_alloc(ctx, &ctx->conv1_bias, (int64_t)C * w0))) return fail(rc);
    if ((rc = dev_alloc(ctx, &ctx->lut, 768))) return fail(rc);
    if ((rc = dev_alloc(ctx, &ctx->pos_tokens, (int64_t)N * D))) return fail(rc);
    if ((rc = dev_alloc(ctx, &ctx->dec_t1, D))) return fail(rc);
    if ((rc = dev_alloc(ctx, &ctx->dec_q, (int64_t)Q * D))) return fail(rc);
    if ((rc = dev_alloc(ctx, &ctx->tmp_vec, 4 * D))) return fail(rc);
    {
        float* rm = nullptr;
        if ((rc = dev_alloc(ctx, &rm, (int64_t)B * C * ctx->P_))) return fail(rc);
        ctx->rowmap = reinterpret_cast<int*>(rm);
        ctx->rowmap_B = -1;
    }
    // ---- activations (camera-major NHWC maps, token-major [B][N][D])
    const int64_t n1 = (int64_t)C * B * ctx->H1 * ctx->W1 * w0;
    const int64_t n2 = (int64_t)C * B * ctx->H2 * ctx->W2 * w0;
    if ((rc = dev_alloc(ctx, &ctx->act1, n1))) return fail(rc);
    for (int i = 0; i < 3; ++i)
        if ((rc = dev_alloc(ctx, &ctx->buf[i], n2))) return fail(rc);
    const int64_t BN_ = (int64_t)B * N;
    if ((rc = dev_alloc(ctx, &ctx->X, BN_ * D))) return fail(rc);
    if ((rc = dev_alloc(ctx, &ctx->X1, BN_ * D))) return fail(rc);
    if ((rc = dev_alloc(ctx, &ctx->XP, BN_ * D))) return fail(rc);
    if ((rc = dev_alloc(ctx, &ctx->act_scale_dev, (int64_t)ctx->convs.size() + 4))) return fail(rc);
    { const char* e1 = getenv("ACTMI_ACT_CALIB"); ctx->act_calib = !(e1 && e1[0] == '0'); }
    { const char* e1 = getenv("ACTMI_LN_XP"); ctx->ln_xp = !(e1 && e1[0] == '0'); }
    { const char* e1 = getenv("ACTMI_LN_HEAD"); ctx->ln_head = !(e1 && e1[0] == '0'); }
    if ((rc = dev_alloc(ctx, &ctx->Y, BN_ * D))) return fail(rc);
    if ((rc = dev_alloc(ctx, &ctx->ATT, BN_ * D))) return fail(rc);
    if ((rc = dev_alloc(ctx, &ctx->QKV, BN_ * 3 * D))) return fail(rc);
    if ((rc = dev_alloc(ctx, &ctx->Hb, BN_ * F))) return fail(rc);
    const int64_t BQ = (int64_t)B * Q;
    if ((rc = dev_alloc(ctx, &ctx->dO, BQ * D))) return fail(rc);
    if ((rc = dev_alloc(ctx, &ctx->dY, BQ * D))) return fail(rc);
    if ((rc = dev_alloc(ctx, &ctx->dT2, BQ * D))) return fail(rc);
    if ((rc = dev_alloc(ctx, &ctx->dH, BQ * F))) return fail(rc);
    if ((rc = dev_alloc(ctx, &ctx->hs, BQ * D))) return fail(rc);
    {
        const int64_t rows = std::max<int64_t>((int64_t)8 * B * Q, (int64_t)2 * B * N);
        ctx->attn_ws_floats = rows * (D + 2 * g.nheads);
        if ((rc = dev_alloc(ctx, &ctx->attn_ws, ctx->attn_ws_floats))) return fail(rc);
    }
    {
        // second stream for the downsample branch of the ResNet blocks (ACTMI_DS_FORK=0 keeps everything on one stream).
        // Not used while the per-launch profiler is on (its events bracket launches on one stream) or when forward
        // contractions are being split (the side branch would share the slice workspace)
        const char* e = getenv("ACTMI_DS_FORK");
        ctx->ds_fork = !(e && e[0] == '0');
        { const char* ef = getenv("ACTMI_FUSE_DS"); ctx->fuse_ds = !(ef && ef[0] == '0'); }
        if (hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) {
            ctx->err = "cannot create the side stream";
            return fail(ACTMI_E_LAUNCH);
        }
        if (const char* e8 = getenv("ACTMI_LN_SPLIT")) ctx->ln_split = atoi(e8);
        if (const char* e9 = getenv("ACTMI_LN_SPLIT_SHORT")) ctx->ln_split_short = atoi(e9);
        const char* e5 = getenv("ACTMI_CAM_PIPE");
        ctx->cam_pipe = !(e5 && e5[0] == '0');             // default on; ACTMI_CAM_PIPE=0: one branch (every launch spans all cameras)
        if (const char* e6 = getenv("ACTMI_BRANCHES")) ctx->nbranch = atoi(e6);
        if (ctx->nbranch < 2) ctx->nbranch = 2;
        if (ctx->nbranch > 4) ctx->nbranch = 4;
        if (ctx->cam_pipe) {
            bool ok = hipEventCreateWithFlags(&ctx->ev_pfork, hipEventDisableTiming) == hipSuccess;
            for (int i = 0; ok && i < ctx->nbranch - 1; ++i)
                ok = hipStreamCreateWithFlags(&ctx->pipe_streams[i], hipStreamNonBlocking) == hipSuccess &&
                     hipEventCreateWithFlags(&ctx->ev_pjoins[i], hipEventDisableTiming) == hipSuccess;
            if (!ok) { ctx->err = "cannot create the branch streams"; return fail(ACTMI_E_LAUNCH); }
            ctx->pipe_stream = ctx->pipe_streams[0];
        }
    }
    ctx->finalized = false;
    if (g.enable_training && (rc = train_create(ctx))) return fail(rc);
    *out = ctx;
    return 0;
}

int engine_destroy(actmi_ctx* ctx) {
    if (!ctx) return 0;
    for (void* p : ctx->allocs) (void)hipFree(p);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->ev_pfork) (void)hipEventDestroy(ctx->ev_pfork);
    for (int i = 0; i < 3; ++i) {
        if (ctx->ev_pjoins[i]) (void)hipEventDestroy(ctx->ev_pjoins[i]);
        if (ctx->pipe_streams[i]) (void)hipStreamDestroy(ctx->pipe_streams[i]);
    }
    if (ctx->train && ctx->train->ev_phase1) (void)hipEventDestroy(ctx->train->ev_phase1);
    delete ctx->train;
    delete ctx;
    return 0;
}

const char* engine_create_error() { return g_create_error.c_str(); }

// ------------------------------------------------------------------------------------------------
// finalize: weight preparation
// ------------------------------------------------------------------------------------------------

// device-side weight preparation; runs at finalize and after every optimizer step (all asynchronous on `st`)
int engine_prepare_weights(actmi_ctx* ctx, hipStream_t st, bool after_step) {
    const actmi_config& g = ctx->cfg;
    const int C = g.num_cams, w0 = g.base_width, D = g.hidden_dim, Q = g.num_queries;
    // conv weights OIHW -> [cam][O][(r,s,c)]: one launch per layer over the cameras (same-named parameters of consecutive
    // backbones are a constant stride apart in the arena)
    {
        const std::string p0 = "backbones.0.0.body.", p1 = "backbones." + std::to_string(C > 1 ? 1 : 0) + ".0.body.";
        const int64_t cam_stride = ctx->P(p1 + "conv1.weight") - ctx->P(p0 + "conv1.weight");
        CHK(launch_repack_conv_w(ctx->P(p0 + "conv1.weight"), ctx->conv1_w, C, w0, 3, 7, 7, cam_stride, (int64_t)w0 * 148, 148, st));
        for (auto& cl : ctx->convs)
            CHK(launch_repack_conv_w(ctx->P(p0 + cl.name + ".weight"), cl.w, C, cl.cout, cl.cin, cl.k, cl.k, cam_stride,
                                     (int64_t)cl.cout * cl.K, cl.K, st));
    }
    // FrozenBN -> scale / bias: buffers the optimizer never touches (reference backbone.py:21-57), so not redone after a step
    if (!after_step)
        for (int cam = 0; cam < C; ++cam) {
            std::string p = "backbones." + std::to_string(cam) + ".0.body.";
            CHK(launch_bn_fold(ctx->P(p + "bn1.weight"), ctx->P(p + "bn1.bias"), ctx->P(p + "bn1.running_mean"),
                               ctx->P(p + "bn1.running_var"), ctx->conv1_scale + cam * w0, ctx->conv1_bias + cam * w0, w0, st));
            for (auto& cl : ctx->convs)
                CHK(launch_bn_fold(ctx->P(p + cl.bn + "weight"), ctx->P(p + cl.bn + "bias"), ctx->P(p + cl.bn + "running_mean"),
                                   ctx->P(p + cl.bn + "running_var"), cl.scale + cam * cl.cout, cl.bias + cam * cl.cout,
                                   cl.cout, st));
        }
    if (ctx->gemm_prec == ACTMI_PREC_F16X3) {
        // per-parameter scales (engine_calibrate_weight_scales); a weight that has outgrown its scale since the last
        // finalize raises ACTMI_FLAG_WEIGHT instead of silently becoming inf
        CHK(launch_split16_map(ctx->pbase, ctx->p16base, ctx->ptotal, ctx->pseg64, ctx->pscale_dev, ctx->flags, st));
        CHK(launch_conv1_wimg(ctx->conv1_w, ctx->conv1_wimg, C, w0, st, ctx->conv1_wscale));
        static const bool tap_inner_on = !(getenv("ACTMI_K_TAP_INNER") && getenv("ACTMI_K_TAP_INNER")[0] == '0');
        for (ConvLayer& cl : ctx->convs) {
            // K order of the image: channel blocks outer, taps inner, for the convolutions of the implicit-GEMM kernel (L2 reuse of
            // the input patch); the direct kernel of layer1 reads (r, s, c)
            const bool direct = cl.k == 3 && cl.stride == 1 && cl.pad == 1 && cl.cin == 64 && cl.cout == 64;
            cl.k_tap_inner = tap_inner_on && cl.k == 3 && (cl.cin % 32) == 0 && !direct &&
                             (int64_t)C * cl.cout * (cl.K + cl.Kx) <= ctx->splitk_ws_floats;
            if (cl.k_tap_inner) {
                CHK(launch_permute_conv_k(cl.w, ctx->splitk_ws, (int64_t)C * cl.cout, cl.k * cl.k, cl.cin, cl.K, st));
                CHK(launch_split16(ctx->splitk_ws, cl.w16, (int64_t)C * cl.cout * cl.K, cl.w16_scale, st, ctx->flags));
            } else {
                CHK(launch_split16(cl.w, cl.w16, (int64_t)C * cl.cout * cl.K, cl.w16_scale, st, ctx->flags));
            }
        }
        // blocks with a downsample branch: [bn2.scale * conv2.w | bn_ds.scale * ds.w] and the summed bias (FrozenBN statistics
        // are buffers: only the weights change under training, so the fold is redone with them)
        for (const ConvLayer& cl : ctx->convs)
            if (cl.wf) {
                const ConvLayer& ds = ctx->convs[cl.ds_index];
                CHK(launch_fold_cat_w(cl.w, cl.scale, cl.bias, ds.w, ds.scale, ds.bias, cl.wf, cl.bias_f, C, cl.cout, cl.K, cl.Kx, st));
                // (at finalize the scale is measured right after this and the image split again: no overflow report from the
                // provisional one)
                const float* src = cl.wf;
                if (cl.k_tap_inner) {
                    CHK(launch_permute_conv_k(cl.wf, ctx->splitk_ws, (int64_t)C * cl.cout, cl.k * cl.k, cl.cin, cl.K + cl.Kx, st));
                    src = ctx->splitk_ws;
                }
                CHK(launch_split16(src, cl.wf16, (int64_t)C * cl.cout * (cl.K + cl.Kx), cl.wf16_scale, st, after_step ? ctx->flags : nullptr));
            }
    }
    // learned rows of the token position table (transformer.py:91-92)
    HIPCHK(hipMemcpyAsync(ctx->pos_tokens, ctx->P("additional_pos_embed.weight"), 2 * D * sizeof(float),
                          hipMemcpyDeviceToDevice, st));
    // decoder layer 0, constant part: tgt = 0 => self-attention output is
    // out_proj(b_v) + b_o for every query; t1 = norm1 of it; q = (t1 + query_embed) Wq^T + bq.
    {
        const DecW& d = ctx->dec[0];
        GemmArgs a = linear_args(d.self_attn.in_b + 2 * D, D, 1, D, d.self_attn.out_w, D, d.self_attn.out_b, ctx->tmp_vec, D);
        CHK(ctx_gemm(ctx, a, st));
        CHK(launch_layernorm(ctx->tmp_vec, nullptr, 0, d.n1w, d.n1b, nullptr, nullptr, ctx->dec_t1, 1, D, 1e-5f, st, &ctx->err));
        GemmArgs q = linear_args(ctx->P("query_embed.weight"), D, Q, D, d.cross.in_w, D, d.cross.in_b, ctx->dec_q, D);
        q.A_add = ctx->dec_t1; q.ld_add = D; q.add_mod = 1; q.add_ncols = D;
        CHK(ctx_gemm(ctx, q, st));
    }
    return 0;
}

int engine_finalize(actmi_ctx* ctx, hipStream_t st) {
    ctx->err.clear();
    const actmi_config& g = ctx->cfg;
    const int C = g.num_cams, D = g.hidden_dim;
    // 2. u8 -> normalised float LUT with the reference's arithmetic:
    //    x = float(v / 255.0 in f64)  (imitate_episodes.py:212), (x - mean) / std in f32 (policy.py:268-272)
    {
        const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
        std::vector<float> lut(768);
        for (int c = 0; c < 3; ++c)
            for (int v = 0; v < 256; ++v) {
                const float x = (float)((double)v / 255.0);
                lut[c * 256 + v] = (x - mean[c]) / stdv[c];
            }
        HIPCHK(hipMemcpyAsync(ctx->lut, lut.data(), 768 * sizeof(float), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    // 3. position table per token: rows 0,1 = additional_pos_embed (transformer.py:91-92); rows 2.. =
    //    PositionEmbeddingSine(normalize=True) (position_encoding.py:30-52), identical for every camera
    {
        const int fh = ctx->fh, fw = ctx->fw, N = ctx->N, npf = D / 2;
        std::vector<float> pos((size_t)N * D, 0.f);
        HIPCHK(hipMemcpy(pos.data(), ctx->P("additional_pos_embed.weight"), 2 * D * sizeof(float), hipMemcpyDeviceToHost));
        const float eps = 1e-6f, scale = (float)(2.0 * M_PI);
        std::vector<float> dim_t(npf);
        for (int k = 0; k < npf; ++k) dim_t[k] = powf(10000.f, (2.f * (float)(k / 2)) / (float)npf);
        for (int hh = 0; hh < fh; ++hh)
            for (int cam = 0; cam < C; ++cam)
                for (int ww = 0; ww < fw; ++ww) {
                    float* row = &pos[(size_t)(2 + hh * (fw * C) + cam * fw + ww) * D];
                    const float y = (float)(hh + 1) / ((float)fh + eps) * scale;
                    const float x = (float)(ww + 1) / ((float)fw + eps) * scale;
                    for (int k = 0; k < npf; ++k) {
                        const float py = y / dim_t[k], px = x / dim_t[k];
                        row[k] = (k & 1) ? cosf(py) : sinf(py);
                        row[npf + k] = (k & 1) ? cosf(px) : sinf(px);
                    }
                }
        HIPCHK(hipMemcpy(ctx->pos_tokens, pos.data(), pos.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    CHK(engine_calibrate_weight_scales(ctx, st));
    CHK(engine_prepare_weights(ctx, st));
    if (ctx->gemm_prec == ACTMI_PREC_F16X3) {
        // range guard of the fused conv2 + downsample images: FrozenBN scales are folded into those weights, so their magnitude
        // is only known now -- measure each fused matrix on the device (max|w| * scale in [2^13, 2^14), capped at 2^12 like every
        // other image) and split again with that scale; training keeps it and raises ACTMI_FLAG_WEIGHT on overflow
        std::vector<ConvLayer*> fl;
        for (auto& cl : ctx->convs) if (cl.wf) fl.push_back(&cl);
        if (!fl.empty() && 2 * fl.size() <= (size_t)(4 * ctx->cfg.hidden_dim)) {
            HIPCHK(hipMemsetAsync(ctx->tmp_vec, 0, 2 * fl.size() * sizeof(float), st));
            for (size_t i = 0; i < fl.size(); ++i) {
                const int64_t Kf = fl[i]->K + fl[i]->Kx;
                CHK(launch_pow2_scale(fl[i]->wf, Kf, ctx->cfg.num_cams * fl[i]->cout, (int)Kf, ctx->tmp_vec + 2 * i, st));
            }
            std::vector<float> sc(2 * fl.size());
            HIPCHK(hipMemcpyAsync(sc.data(), ctx->tmp_vec, sc.size() * sizeof(float), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            for (size_t i = 0; i < fl.size(); ++i) {
                float v = sc[2 * i];
                if (!(v > 0.f) || !(v <= 3.0e38f)) { ctx->err = "fused weights of " + fl[i]->name + " are not finite"; return ACTMI_E_INVALID; }
                fl[i]->wf16_scale = v < 4096.f ? v : 4096.f;
                const float* src = fl[i]->wf;
                if (fl[i]->k_tap_inner) {
                    CHK(launch_permute_conv_k(fl[i]->wf, ctx->splitk_ws, (int64_t)ctx->cfg.num_cams * fl[i]->cout, fl[i]->k * fl[i]->k, fl[i]->cin,
                                              fl[i]->K + fl[i]->Kx, st));
                    src = ctx->splitk_ws;
                }
                CHK(launch_split16(src, fl[i]->wf16, (int64_t)ctx->cfg.num_cams * fl[i]->cout * (fl[i]->K + fl[i]->Kx),
                                   fl[i]->wf16_scale, st, ctx->flags));
            }
        }
    }
    HIPCHK(hipStreamSynchronize(st));
    CHK(engine_calibrate_activations(ctx, st));
    ctx->finalized = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// forward pieces
// ------------------------------------------------------------------------------------------------

// Activation range guard of the f16x3 forward.  Activations are split into fp16 pieces on
// their way into LDS: a map whose magnitudes sit far below 1 loses the lo piece to fp16 subnormals (absolute floor 2^-25) and
// one beyond 65504 overflows.  The trunk's maps pass through FrozenBatchNorm2d (backbone.py:21-57), whose frozen statistics
// can leave them at any magnitude in a trained checkpoint, so actmi_finalize runs ONE calibration forward of the trunk on a
// synthetic image and gives every convolution whose input lies outside [2^-4, 2^11] a power-of-two input pre-scale that brings
// its largest magnitude to ~2^9..2^10 (64x headroom for hotter frames); inside that window the scale stays 1 and the hot
// (unmasked) kernel flavour runs.  Power-of-two scaling is exact; it is undone through the epilogue's alpha.
int engine_measure_act_scale(actmi_ctx* ctx, const float* x, int64_t rows, int cols, hipStream_t st, float* out) {
    *out = 1.f;
    if (rows <= 0 || rows > 0x7fffffff) return 0;
    float* slot = ctx->tmp_vec + 2;             // [scale, bits word]: the word must be zero before the first use
    HIPCHK(hipMemsetAsync(slot, 0, 2 * sizeof(float), st));
    CHK(launch_pow2_scale(x, cols, (int)rows, cols, slot, st));
    float s = 1.f;
    HIPCHK(hipMemcpyAsync(&s, slot, sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    // s brings max|x| into [2^13, 2^14): max|x| < 2^-4  <=>  s > 2^17;  max|x| >= 2^11  <=>  s <= 2^2
    if (s > 131072.f || s <= 4.f) *out = s * (1.f / 16.f);
    return 0;
}

int engine_calibrate_activations(actmi_ctx* ctx, hipStream_t st) {
    for (auto& cl : ctx->convs) cl.a_scale = 1.f;
    ctx->ip_a_scale = 1.f;
    if (ctx->gemm_prec != ACTMI_PREC_F16X3 || !ctx->act_calib) return 0;
    const actmi_config& g = ctx->cfg;
    const size_t nbytes = (size_t)g.num_cams * g.image_h * g.image_w * 3;
    std::vector<unsigned char> img(nbytes);
    uint32_t x = 0x9E3779B9u;                   // a fixed noise frame covering the whole u8 range
    for (size_t i = 0; i < nbytes; ++i) { x = x * 1664525u + 1013904223u; img[i] = (unsigned char)(x >> 24); }
    void* dimg = nullptr;
    if (hipMalloc(&dimg, nbytes) != hipSuccess) { ctx->err = "hipMalloc (calibration frame)"; return ACTMI_E_NOMEM; }
    int rc = hipMemcpy(dimg, img.data(), nbytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : ACTMI_E_LAUNCH;
    if (rc == 0) {
        ctx->calibrating = true;
        const std::string keep = ctx->stop_stage;
        ctx->stop_stage.clear();
        rc = engine_backbone(ctx, dimg, ACTMI_IMG_U8_NHWC, 1, st);
        ctx->stop_stage = keep;
        ctx->calibrating = false;
        if (hipStreamSynchronize(st) != hipSuccess && rc == 0) rc = ACTMI_E_LAUNCH;
    }
    (void)hipFree(dimg);
    ctx->dbg.clear();
    return rc;
}

// nb concurrent branches of one forward: branch 0 on the caller's stream, branch i > 0 on ctx->pipe_streams[i - 1], forked from
// and joined back into the caller's stream with events (parallel branches of the graph under capture).  Whatever fails -- a
// launch inside a branch or the event plumbing itself -- every stream that WAS forked is still joined before the first error
// is returned (an early return left forked streams unjoined, which under hipGraph capture voids the capture with
// a secondary StreamCaptureUnjoined error that hid the real cause, and in eager mode let branch work race the next call).
template <class Body>
static int run_branches(actmi_ctx* ctx, int nb, hipStream_t st, Body body) {
    ctx->policy_mult = nb;
    int rc = 0;
    hipError_t he = hipEventRecord(ctx->ev_pfork, st);
    const char* he_where = "hipEventRecord(fork)";
    bool forked[4] = {false, false, false, false};
    for (int i = nb - 1; i >= 0; --i) {
        if (rc != 0 || he != hipSuccess) break;
        hipStream_t bs = i ? ctx->pipe_streams[i - 1] : st;
        if (i) {
            he = hipStreamWaitEvent(bs, ctx->ev_pfork, 0);
            if (he != hipSuccess) { he_where = "hipStreamWaitEvent(fork)"; break; }
            forked[i] = true;
        }
        rc = body(i, bs);               // leaves its message in ctx->err (first error wins: CHK / HIPCHK)
    }
    for (int i = 1; i < nb; ++i) {      // join every forked branch, error or not
        if (!forked[i]) continue;
        hipError_t e = hipEventRecord(ctx->ev_pjoins[i - 1], ctx->pipe_streams[i - 1]);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, ctx->ev_pjoins[i - 1], 0);
        if (e != hipSuccess && he == hipSuccess) { he = e; he_where = "branch join"; }
    }
    ctx->policy_mult = 1;
    if (rc != 0) return rc;
    if (he != hipSuccess) {
        if (ctx->err.empty()) ctx->err = std::string(he_where) + ": " + hipGetErrorString(he);
        return ACTMI_E_LAUNCH;
    }
    return 0;
}

// multi-camera ResNet18 trunk + input_proj -> token rows 2.. of X   (backbone.py:66-71, detr_vae.py:180-185)
int engine_backbone(actmi_ctx* ctx, const void* image, int fmt, int B, hipStream_t st) {
    const actmi_config& g = ctx->cfg;
    const int C = g.num_cams, w0 = g.base_width, D = g.hidden_dim;
    Conv1Args c1;
    c1.image = image; c1.fmt = fmt; c1.lut = ctx->lut; c1.w = ctx->conv1_w; c1.scale = ctx->conv1_scale;
    c1.bias = ctx->conv1_bias; c1.out = ctx->act1; c1.B = B; c1.C = C; c1.H = g.image_h; c1.W = g.image_w;
    c1.Ho = ctx->H1; c1.Wo = ctx->W1; c1.Cout = w0;
    c1.prec = ctx->gemm_prec;
    c1.wimg = reinterpret_cast<const unsigned char*>(ctx->conv1_wimg);
    c1.wscale = ctx->conv1_wscale;
    // inference never needs conv1's own map: the stem emits the vertical half of the max pool (half the bytes) and a
    // row-wise pass finishes it.  Same maxima, so the result is bit-identical to conv1 -> 3x3 pool.
    const bool vpool = ctx->conv1_vpool && ctx->stop_stage != "conv1" && ctx->gemm_prec == ACTMI_PREC_F16X3 && (ctx->H1 & 1) == 0 &&
                       (w0 & 3) == 0 && ctx->H2 == ctx->H1 / 2;
    // stem (conv1 + pool) of the cameras [c0, c0 + nc) on stream ss
    auto run_stem = [&](int c0, int nc, hipStream_t ss) -> int {
        Conv1Args cc = c1;
        cc.cam0 = c0; cc.ncam = nc;
        if (vpool) {
            cc.vpool = 1;
            CHK(launch_conv1(cc, ss, &ctx->err));
            const int64_t a_cam = (int64_t)B * ctx->H2 * ctx->W1 * w0, p_cam = (int64_t)B * ctx->H2 * ctx->W2 * w0;
            CHK(launch_hpool(ctx->act1 + c0 * a_cam, ctx->buf[0] + c0 * p_cam, nc * B * ctx->H2, ctx->W1, w0, ctx->W2, ss));
        } else {
            CHK(launch_conv1(cc, ss, &ctx->err));
            const int64_t a_cam = (int64_t)B * ctx->H1 * ctx->W1 * w0, p_cam = (int64_t)B * ctx->H2 * ctx->W2 * w0;
            CHK(launch_maxpool(ctx->act1 + c0 * a_cam, ctx->buf[0] + c0 * p_cam, nc * B, ctx->H1, ctx->W1, w0, ctx->H2, ctx->W2, ss));
        }
        return 0;
    };
    // the stem inside the branches (ACTMI_STEM_BRANCH=1): one branch's conv1 (bound by its own instruction stream) beside the
    // other's pool / layer1 launches
    static const bool stem_in_branch = getenv("ACTMI_STEM_BRANCH") && getenv("ACTMI_STEM_BRANCH")[0] == '1';
    const bool pipe_early = stem_in_branch && ctx->cam_pipe && ctx->pipe_stream && C >= 2 && !prof_enabled() && ctx->stop_stage.empty() &&
                            !ctx->calibrating;
    if (!pipe_early) CHK(run_stem(0, C, st));
    ctx->dbg.clear();
    ctx->dbg["conv1"] = {ctx->act1, (int64_t)C * B * ctx->H1 * ctx->W1 * w0};
    ctx->dbg["maxpool"] = {ctx->buf[0], (int64_t)C * B * ctx->H2 * ctx->W2 * w0};
    if (ctx->stop_stage == "conv1" || ctx->stop_stage == "maxpool") return 1;
    // one convolution of the trunk for the cameras [c0, c0 + nc) (feature maps and weights are camera-major, so a camera range
    // is a pointer offset + a group count); half >= 0: one of two concurrent branches (own half of the slice workspace)
    auto run_conv_on = [&](const ConvLayer& cl, const float* in, float* out, const float* res, int relu, hipStream_t cs, int c0,
                           int nc, int half) -> int {
        // (in / out / res already point at the range's first camera: run_layers)
        const int64_t in_cam = (int64_t)B * cl.H * cl.W * cl.cin;
        const int64_t w_cam = (int64_t)cl.cout * cl.K;
        const float* scale = cl.scale + (int64_t)c0 * cl.cout;
        const float* bias = cl.bias + (int64_t)c0 * cl.cout;
        const int cl_index = (int)(&cl - ctx->convs.data());
        if (ctx->calibrating) {
            // calibration forward (actmi_finalize): measure this layer's input, fix its pre-scale, THEN run the layer with it
            // (so that a map far outside the fp16 range does not poison the measurements downstream); one host sync per layer
            float sc = 1.f;
            CHK(engine_measure_act_scale(ctx, in, (int64_t)nc * B * cl.H * cl.W, cl.cin, cs, &sc));
            const_cast<ConvLayer&>(cl).a_scale = sc;
            HIPCHK(hipMemcpyAsync(ctx->act_scale_dev + cl_index, &cl.a_scale, sizeof(float), hipMemcpyHostToDevice, cs));
            HIPCHK(hipStreamSynchronize(cs));
        }
        if (ctx->gemm_prec == ACTMI_PREC_F16X3 && cl.k == 3 && cl.stride == 1 && cl.pad == 1 && cl.cin == 64 && cl.cout == 64) {
            // layer1: direct convolution over an LDS-resident patch (the im2col GEMM is L2-traffic bound at 64 channels)
            Conv3Args c3;
            c3.x = in; c3.w16 = cl.w16 + c0 * w_cam; c3.scale = scale; c3.bias = bias; c3.res = res; c3.out = out;
            c3.G = nc; c3.B = B; c3.H = cl.H; c3.W = cl.W; c3.relu = relu; c3.w_scale = cl.w16_scale;
            if (cl.a_scale != 1.f) c3.x_scale_dev = ctx->act_scale_dev + cl_index;
            return launch_conv3x3_c64(c3, cs, &ctx->err);
        }
        GemmArgs a;
        memset(&a, 0, sizeof(a));
        a.mode = 1;
        a.A = in; a.H = cl.H; a.W = cl.W; a.Cin = cl.cin; a.KH = a.KW = cl.k; a.stride = cl.stride; a.pad = cl.pad;
        a.Ho = cl.Ho; a.Wo = cl.Wo; a.img_stride = (int64_t)cl.H * cl.W * cl.cin;
        a.M = B * cl.Ho * cl.Wo; a.N = cl.cout; a.K = cl.K;
        a.Bw = cl.w + c0 * w_cam; a.ldb = cl.K; a.scale = scale; a.bias = bias; a.res = res; a.ldres = cl.cout; a.relu = relu;
        a.C = out; a.ldc = cl.cout;
        a.groups = nc;
        a.gA = in_cam; a.gB = w_cam; a.gSB = cl.cout;
        a.gC = (int64_t)a.M * cl.cout; a.gRes = a.gC;
        if (ctx->gemm_prec == ACTMI_PREC_F16X3 && cl.a_scale != 1.f) a.a_scale = cl.a_scale;
        return ctx_gemm(ctx, a, cs, half);
    };
    // the side branch needs the split-K workspace for itself: only taken when the main stream's launches do not split
    // (not while the per-launch profiler brackets launches with events, nor for the debug early-outs)
    const bool pipe = ctx->cam_pipe && ctx->pipe_stream && C >= 2 && !prof_enabled() && ctx->stop_stage.empty() && !ctx->calibrating;
    const bool fork_ds = ctx->side_stream != nullptr && ctx->ds_fork && !pipe && !ctx->calibrating;
    float* final_cur = nullptr;
    // layer1 .. layer4 for the cameras [c0, c0 + nc) on stream ls
    auto run_layers = [&](int c0, int nc, hipStream_t ls, int half) -> int {
        // the range's maps live at the offset of its first camera in the POOLED map (the largest per-camera block): every
        // later map of the range fits behind it without reaching the next range's block, whatever layer the other branch is in
        const int64_t hb = (int64_t)c0 * B * ctx->H2 * ctx->W2 * w0;
        float *cur = ctx->buf[0] + hb, *s1 = ctx->buf[1] + hb, *s2 = ctx->buf[2] + hb;
        auto run_conv = [&](const ConvLayer& cl, const float* in, float* out, const float* res, int relu) -> int {
            return run_conv_on(cl, in, out, res, relu, ls, c0, nc, half);
        };
        // conv2 of a downsample block with the branch in its contraction: y = relu([W2' | Wd'] [y1 taps ; x at stride 2] + b)
        auto run_conv_fused = [&](const ConvLayer& cl, const float* y1, const float* x, float* out) -> int {
            const ConvLayer& ds = ctx->convs[cl.ds_index];
            const int Kf = cl.K + cl.Kx;
            GemmArgs a;
            memset(&a, 0, sizeof(a));
            a.mode = 1;
            a.A = y1; a.H = cl.H; a.W = cl.W; a.Cin = cl.cin; a.KH = a.KW = cl.k; a.stride = cl.stride; a.pad = cl.pad;
            a.Ho = cl.Ho; a.Wo = cl.Wo; a.img_stride = (int64_t)cl.H * cl.W * cl.cin;
            a.M = B * cl.Ho * cl.Wo; a.N = cl.cout; a.K = Kf;
            a.Ax = x; a.kx_begin = cl.K; a.Hx = ds.H; a.Wx = ds.W; a.Cx = ds.cin; a.stride_x = ds.stride;
            a.gAx = (int64_t)B * ds.H * ds.W * ds.cin;
            a.Bw = cl.wf + (int64_t)c0 * cl.cout * Kf; a.ldb = Kf; a.bias = cl.bias_f + (int64_t)c0 * cl.cout; a.relu = 1;
            a.C = out; a.ldc = cl.cout;
            a.groups = nc;
            a.gA = (int64_t)B * cl.H * cl.W * cl.cin; a.gB = (int64_t)cl.cout * Kf; a.gSB = cl.cout;
            a.gC = (int64_t)a.M * cl.cout;
            return ctx_gemm(ctx, a, ls, half);
        };
        size_t ci = 0;
        for (int li = 1; li <= 4; ++li) {
            for (int bi = 0; bi < 2; ++bi) {
                const ConvLayer& k1 = ctx->convs[ci++];
                const ConvLayer& k2 = ctx->convs[ci++];
                const bool has_ds = (bi == 0 && li > 1);
                if (has_ds && ctx->fuse_ds && k2.wf && ctx->gemm_prec == ACTMI_PREC_F16X3 && !ctx->calibrating && k2.a_scale == 1.f &&
                    ctx->convs[k2.ds_index].a_scale == 1.f) {
                    // the downsample branch rides in conv2's contraction (second source = the block input at stride 2):
                    // two launches instead of three, and the branch's map is neither written nor read back
                    ++ci;                                        // (the downsample layer's own entry)
                    CHK(run_conv(k1, cur, s1, nullptr, 1));
                    CHK(run_conv_fused(k2, s1, cur, s2));
                    std::swap(cur, s2);                          // x stays live until conv2 has read it: the output goes to s2
                } else if (has_ds) {
                    const ConvLayer& ds = ctx->convs[ci++];
                    if (fork_ds) {
                        // the 1x1 / stride-2 downsample (23-50 us, HBM bound, few workgroups) only needs the block input: it
                        // runs on a second stream beside the block's first 3x3 convolution (fork / join through events: in a
                        // captured graph these are two parallel branches) and fills CUs that launch leaves idle
                        HIPCHK(hipEventRecord(ctx->ev_fork, ls));
                        HIPCHK(hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
                        // from here the side stream is forked: join it whatever happens, then report the first error
                        const int r1 = run_conv_on(ds, cur, s2, nullptr, 0, ctx->side_stream, c0, nc, half);
                        const hipError_t e1 = hipEventRecord(ctx->ev_join, ctx->side_stream);
                        const int r2 = r1 == 0 ? run_conv(k1, cur, s1, nullptr, 1) : 0;
                        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(ls, ctx->ev_join, 0) : e1;
                        CHK(r1);
                        CHK(r2);
                        HIPCHK(e1);
                        HIPCHK(e2);
                    } else {
                        CHK(run_conv(k1, cur, s1, nullptr, 1));
                        CHK(run_conv(ds, cur, s2, nullptr, 0));
                    }
                    CHK(run_conv(k2, s1, cur, s2, 1));       // x is dead: reuse its buffer for the block output
                } else {
                    CHK(run_conv(k1, cur, s1, nullptr, 1));
                    CHK(run_conv(k2, s1, s2, cur, 1));
                    std::swap(cur, s2);
                }
                if (bi == 1 && c0 == 0) {
                    const std::string nm = "layer" + std::to_string(li);
                    ctx->dbg[nm] = {cur, (int64_t)C * B * k2.Ho * k2.Wo * k2.cout};
                    if (ctx->stop_stage == nm) { final_cur = cur; return 1; }    // debug early-out: buffers rotate, views alias
                }
            }
        }
        final_cur = cur;
        // input_proj (1x1 convolution, detr_vae.py:184) of the range's layer4 maps, rows scattered to their tokens
        GemmArgs ip = linear_args(cur, 8 * w0, nc * B * ctx->P_, 8 * w0, ctx->P("input_proj.weight"), D, ctx->P("input_proj.bias"),
                                  ctx->X, D);
        ip.rowmap = ctx->rowmap + (int64_t)c0 * B * ctx->P_;
        if (ctx->calibrating) {
            CHK(engine_measure_act_scale(ctx, cur, (int64_t)nc * B * ctx->P_, 8 * w0, ls, &ctx->ip_a_scale));
            HIPCHK(hipStreamSynchronize(ls));
        }
        if (ctx->gemm_prec == ACTMI_PREC_F16X3 && ctx->ip_a_scale != 1.f) ip.a_scale = ctx->ip_a_scale;
        return ctx_gemm(ctx, ip, ls, half);
    };
    if (ctx->rowmap_B != B) {
        CHK(launch_build_rowmap(ctx->rowmap, B, C, ctx->fh, ctx->fw, ctx->N, st));
        ctx->rowmap_B = B;
    }
    if (pipe) {
        // two camera halves as two parallel branches: when one half's launch runs out of workgroups (layer3: 300 per half on
        // 512 slots) the other half's current launch fills the CUs, and no launch boundary drains the whole chip
        // branch i takes the cameras [i C / nb, (i+1) C / nb); branch 0 runs on the caller's stream
        const int nb = C < ctx->nbranch ? C : ctx->nbranch;
        const int rc = run_branches(ctx, nb, st, [&](int i, hipStream_t bs) -> int {
            const int c0 = i * C / nb, c1 = (i + 1) * C / nb;
            int r = pipe_early ? run_stem(c0, c1 - c0, bs) : 0;
            if (r == 0) r = run_layers(c0, c1 - c0, bs, i);
            return r;
        });
        if (rc != 0) return rc;
    } else {
        const int rc = run_layers(0, C, st, -1);
        if (rc != 0) return rc;
    }
    (void)final_cur;
    return 0;
}

// The transformer's activation buffers as seen by one branch: the samples [b0, b0 + nb) of the batch (token-major [B][N][D]
// layouts: a batch range is a pointer offset).  half >= 0: one of two concurrent branches (own half of the slice workspace).
struct TView {
    float *X, *XP, *QKV, *ATT, *Y, *X1, *Hb, *dO, *dY, *dT2, *dH, *hs, *attn_ws;
    int64_t attn_ws_floats;
    int half;
};
static TView make_view(const actmi_ctx* ctx, int b0, int nb, int half) {
    const actmi_config& g = ctx->cfg;
    const int64_t D = g.hidden_dim, F = g.dim_feedforward, Q = g.num_queries, N = ctx->N;
    const int64_t ws_per = ctx->attn_ws_floats / g.max_batch;
    TView v;
    v.X = ctx->X + b0 * N * D; v.XP = ctx->XP + b0 * N * D; v.QKV = ctx->QKV + b0 * N * 3 * D; v.ATT = ctx->ATT + b0 * N * D; v.Y = ctx->Y + b0 * N * D;
    v.X1 = ctx->X1 + b0 * N * D; v.Hb = ctx->Hb + b0 * N * F;
    v.dO = ctx->dO + b0 * Q * D; v.dY = ctx->dY + b0 * Q * D; v.dT2 = ctx->dT2 + b0 * Q * D; v.dH = ctx->dH + b0 * Q * F;
    v.hs = ctx->hs + b0 * Q * D;
    v.attn_ws = ctx->attn_ws + b0 * ws_per; v.attn_ws_floats = (half < 0) ? ctx->attn_ws_floats : nb * ws_per;
    v.half = half;
    return v;
}

// one post-norm encoder layer on x [B*n][D] in place (transformer.py:211-224)
// xp_in: V.XP already holds x + pos (written by the LayerNorm that produced x); xp_out: this layer's last LayerNorm writes
// x + pos of ITS output into V.XP for the next attention block
int engine_encoder_layer(actmi_ctx* ctx, const EncW& w, const TView& V, const float* pos, int B, int n, const uint8_t* kpm,
                         hipStream_t st, bool xp_in, bool xp_out) {
    float* x = V.X;
    const actmi_config& g = ctx->cfg;
    const int D = g.hidden_dim, F = g.dim_feedforward, M = B * n, hd = D / g.nheads;
    GemmArgs qkv = linear_args(x, D, M, D, w.attn.in_w, 3 * D, w.attn.in_b, V.QKV, 3 * D);
    if (xp_in) { qkv.A_alt = V.XP; qkv.alt_ncols = 2 * D; }                      // q = k = x + pos (a matrix already), v = x
    else { qkv.A_add = pos; qkv.ld_add = D; qkv.add_mod = n; qkv.add_ncols = 2 * D; }     // q = k = x + pos, v = x
    CHK(ctx_gemm(ctx, qkv, st, V.half));
    AttnArgs at;
    memset(&at, 0, sizeof(at));
    at.Q = V.QKV; at.q_bs = (int64_t)n * 3 * D; at.q_rs = 3 * D;
    at.K = V.QKV + D; at.k_bs = at.q_bs; at.k_rs = 3 * D;
    at.V = V.QKV + 2 * D; at.v_bs = at.q_bs; at.v_rs = 3 * D;
    at.O = V.ATT; at.o_bs = (int64_t)n * D; at.o_rs = D;
    at.kpm = kpm; at.kpm_bs = n;
    at.B = B; at.H = g.nheads; at.Nq = n; at.Nk = n; at.HD = hd; at.scale = 1.0f / sqrtf((float)hd);
    at.ws = V.attn_ws; at.ws_floats = V.attn_ws_floats;
    at.prec = ctx->gemm_prec;
    CHK(launch_attention(at, st, &ctx->err));
    GemmArgs op = linear_args(V.ATT, D, M, D, w.attn.out_w, D, w.attn.out_b, V.Y, D);
    op.res = x; op.ldres = D;
    LnFuse ln1{w.n1w, w.n1b, nullptr, nullptr, V.X1, 1e-5f, false};
    CHK(ctx_gemm(ctx, op, st, V.half, &ln1));
    if (!ln1.done) CHK(launch_layernorm(V.Y, nullptr, 0, w.n1w, w.n1b, nullptr, nullptr, V.X1, M, D, 1e-5f, st, &ctx->err));
    GemmArgs f1 = linear_args(V.X1, D, M, D, w.l1w, F, w.l1b, V.Hb, F);
    f1.relu = 1;
    CHK(ctx_gemm(ctx, f1, st, V.half));
    GemmArgs f2 = linear_args(V.Hb, F, M, F, w.l2w, D, w.l2b, V.Y, D);
    f2.res = V.X1; f2.ldres = D;
    LnExtra ex;
    if (xp_out) { ex.y2 = V.XP; ex.add2 = pos; ex.add2_mod = n; }
    LnFuse ln2{w.n2w, w.n2b, nullptr, nullptr, x, 1e-5f, false, xp_out ? &ex : nullptr};
    CHK(ctx_gemm(ctx, f2, st, V.half, &ln2));
    if (!ln2.done)
        CHK(launch_layernorm(V.Y, nullptr, 0, w.n2w, w.n2b, nullptr, nullptr, x, M, D, 1e-5f, st, &ctx->err, 1, 0, nullptr,
                             xp_out ? &ex : nullptr));
    return 0;
}

// decoder layer 0 with the constant query path + heads (transformer.py:274-295,175; detr_vae.py:245,252)
int engine_decoder_infer(actmi_ctx* ctx, const TView& V, int B, float* a_hat, hipStream_t st, bool xp_in) {
    const actmi_config& g = ctx->cfg;
    const int D = g.hidden_dim, F = g.dim_feedforward, Q = g.num_queries, N = ctx->N, hd = D / g.nheads;
    const DecW& d = ctx->dec[0];
    float* KV = V.QKV;   // [B*N][2D]
    GemmArgs kv = linear_args(V.X, D, B * N, D, d.cross.in_w + (int64_t)D * D, 2 * D, d.cross.in_b + D, KV, 2 * D);
    if (xp_in) { kv.A_alt = V.XP; kv.alt_ncols = D; }                                   // k = memory + pos (a matrix already), v = memory
    else { kv.A_add = ctx->pos_tokens; kv.ld_add = D; kv.add_mod = N; kv.add_ncols = D; }      // k = memory + pos, v = memory
    CHK(ctx_gemm(ctx, kv, st, V.half));
    AttnArgs at;
    memset(&at, 0, sizeof(at));
    at.Q = ctx->dec_q; at.q_bs = 0; at.q_rs = D;
    at.K = KV; at.k_bs = (int64_t)N * 2 * D; at.k_rs = 2 * D;
    at.V = KV + D; at.v_bs = at.k_bs; at.v_rs = 2 * D;
    at.O = V.dO; at.o_bs = (int64_t)Q * D; at.o_rs = D;
    at.B = B; at.H = g.nheads; at.Nq = Q; at.Nk = N; at.HD = hd; at.scale = 1.0f / sqrtf((float)hd);
    at.ws = V.attn_ws; at.ws_floats = V.attn_ws_floats;
    at.prec = ctx->gemm_prec;
    CHK(launch_attention(at, st, &ctx->err));
    const int M = B * Q;
    GemmArgs op = linear_args(V.dO, D, M, D, d.cross.out_w, D, d.cross.out_b, V.dY, D);
    op.res = ctx->dec_t1; op.ldres = D; op.res_mod = 1;
    CHK(ctx_gemm(ctx, op, st, V.half));
    CHK(launch_layernorm(V.dY, nullptr, 0, d.n2w, d.n2b, nullptr, nullptr, V.dT2, M, D, 1e-5f, st, &ctx->err));
    GemmArgs f1 = linear_args(V.dT2, D, M, D, d.l1w, F, d.l1b, V.dH, F);
    f1.relu = 1;
    CHK(ctx_gemm(ctx, f1, st, V.half));
    GemmArgs f2 = linear_args(V.dH, F, M, F, d.l2w, D, d.l2b, V.dY, D);
    f2.res = V.dT2; f2.ldres = D;
    // norm3 + decoder.norm; the action head (detr_vae.py:252) is computed from the finished row in the same kernel, with the
    // default-on output guard: an operand that left the fp16 range of the f16x3 products surfaces as inf / NaN in a_hat and
    // raises the flag (read at the caller's next natural synchronisation: actmi_get_flags)
    LnExtra hx;
    const bool head_in_ln = ctx->ln_head && g.action_dim <= 64;
    if (head_in_ln) {
        hx.head_out = a_hat; hx.head_w = ctx->P("action_head.weight"); hx.head_b = ctx->P("action_head.bias"); hx.head_n = g.action_dim;
        hx.flag = ctx->flags; hx.flag_bit = ACTMI_FLAG_OUTPUT;
    }
    LnFuse ln3{d.n3w, d.n3b, ctx->P("transformer.decoder.norm.weight"), ctx->P("transformer.decoder.norm.bias"), V.hs, 1e-5f, false,
               head_in_ln ? &hx : nullptr};
    CHK(ctx_gemm(ctx, f2, st, V.half, &ln3));
    if (!ln3.done)
        CHK(launch_layernorm(V.dY, nullptr, 0, d.n3w, d.n3b, ctx->P("transformer.decoder.norm.weight"),
                             ctx->P("transformer.decoder.norm.bias"), V.hs, M, D, 1e-5f, st, &ctx->err, 1, 0, nullptr,
                             head_in_ln ? &hx : nullptr));
    if (!head_in_ln) {
        GemmArgs ah = linear_args(V.hs, D, M, D, ctx->P("action_head.weight"), g.action_dim, ctx->P("action_head.bias"),
                                  a_hat, g.action_dim);
        ah.finite_flag = ctx->flags; ah.finite_bit = ACTMI_FLAG_OUTPUT;
        CHK(ctx_gemm(ctx, ah, st, V.half));
    }
    if (V.half <= 0) ctx->dbg["hs"] = {ctx->hs, (int64_t)ctx->last_B * Q * D};
    return 0;
}

int engine_forward_infer(actmi_ctx* ctx, const float* qpos, const void* image, int fmt, int B, float* a_hat,
                         hipStream_t st, const float* vq_sample) {
    ctx->err.clear();
    if (!ctx->finalized) { ctx->err = "forward before finalize"; return ACTMI_E_STATE; }
    if (B < 1 || B > ctx->cfg.max_batch) { ctx->err = "batch exceeds max_batch"; return ACTMI_E_INVALID; }
    if (fmt != ACTMI_IMG_U8_NHWC && fmt != ACTMI_IMG_F32_NCHW) { ctx->err = "bad image format"; return ACTMI_E_INVALID; }
    const actmi_config& g = ctx->cfg;
    const int D = g.hidden_dim, N = ctx->N;
    // actmi_set_forward_phase: the step as two halves a caller can capture into two graphs -- the trunk with the token assembly
    // (the only reader of `image` and `qpos`, and the HBM-heavy part) and the transformer -- so that the host-to-device copy of the
    // NEXT frame hangs on an ordinary stream event between them and runs beside the transformer.  Phase 2 continues from the tokens
    // phase 1 left in ctx->X.
    if (ctx->fwd_phase != 2) {
        const int rc = engine_backbone(ctx, image, fmt, B, st);
        if (rc == 1) return 0;
        if (rc != 0) return rc;
        // token 0: latent_input = latent_out_proj(0) = bias (detr_vae.py:158-159), or latent_out_proj(code) for VQ-ACT
        // (detr_vae.py:155-156); token 1: proprio (detr_vae.py:213)
        float* fill_dst = nullptr;
        if (g.vq && vq_sample) {
            const int K = g.vq_class * g.vq_dim;
            GemmArgs lz = linear_args(vq_sample, K, B, K, ctx->P("latent_out_proj.weight"), D, ctx->P("latent_out_proj.bias"),
                                      ctx->X, (int64_t)N * D);
            CHK(ctx_gemm(ctx, lz, st));
        } else {
            fill_dst = ctx->X;                   // token 0 = the bias row, written by the proprio projection's launch
        }
        CHK(launch_small_linear(qpos, g.state_dim, ctx->P("input_proj_robot_state.weight"),
                                ctx->P("input_proj_robot_state.bias"), ctx->X + D, (int64_t)N * D, B, D, g.state_dim, st, fill_dst,
                                ctx->P("latent_out_proj.bias"), 0));
        if (ctx->fwd_phase == 1) { ctx->last_B = B; return 0; }
    } else if (B != ctx->last_B) {
        ctx->err = "forward phase 2 without a phase 1 of the same batch before it";
        return ACTMI_E_STATE;
    }
    ctx->dbg["src"] = {ctx->X, (int64_t)B * N * D};
    if (ctx->stop_stage == "src") return 0;
    ctx->last_B = B;
    auto run_transformer = [&](int b0, int nb, hipStream_t ts, int half) -> int {
        const TView V = make_view(ctx, b0, nb, half);
        // x + pos as a matrix of its own (written by each layer's last LayerNorm) wherever the consumer's column split falls on
        // a tile boundary: the packed QKV products of layers 1.. and the decoder's KV product are then plain GEMMs
        const bool xp_qkv = ctx->ln_xp && ((2 * D) % 128) == 0, xp_kv = ctx->ln_xp && (D % 128) == 0;
        for (int l = 0; l < g.enc_layers; ++l)
            CHK(engine_encoder_layer(ctx, ctx->enc[l], V, ctx->pos_tokens, nb, N, nullptr, ts, l > 0 && xp_qkv,
                                     l + 1 < g.enc_layers ? xp_qkv : xp_kv));
        return engine_decoder_infer(ctx, V, nb, a_hat + (int64_t)b0 * g.num_queries * g.action_dim, ts, xp_kv);
    };
    // encoder + decoder as two concurrent branches over the two halves of the batch (samples are independent): the 304-workgroup
    // launches (out-proj, FFN2: 59 % of the 512 residency slots) of one half run beside the other half's launches
    const bool tpipe = ctx->cam_pipe && ctx->pipe_stream && B >= 2 && !prof_enabled() && ctx->stop_stage.empty();
    ctx->dbg["memory"] = {ctx->X, (int64_t)B * N * D};
    if (tpipe) {
        const int nb = B < ctx->nbranch ? B : ctx->nbranch;
        const int rc = run_branches(ctx, nb, st, [&](int i, hipStream_t bs) -> int {
            const int b0 = i * B / nb, b1 = (i + 1) * B / nb;
            return run_transformer(b0, b1 - b0, bs, i);
        });
        if (rc != 0) return rc;
    } else {
        CHK(run_transformer(0, B, st, -1));
    }
    return 0;
}
