// extern "C" surface of libactmi (declared in include/actmi.h).  Nothing here throws across the boundary.
#include "engine.h"

#include <cstring>
#include <vector>

namespace {
thread_local std::string g_op_error;
inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
int bad(actmi_ctx* h, const std::string& m, int code = ACTMI_E_INVALID) { h->err = m; return code; }
// a handle is bound to the device it was created on: calls made while another device is current switch to it for
// their duration (one process driving several GPUs; torch's current device is whatever the caller left it at)
struct DevGuard {
    int prev = -1, want;
    explicit DevGuard(int dev) : want(dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != want) (void)hipSetDevice(want);
    }
    ~DevGuard() { if (prev >= 0 && prev != want) (void)hipSetDevice(prev); }
};
// entry of every call on a handle: no stale message from an earlier failure, handle's device current
#define ENTER(h) (h)->err.clear(); DevGuard _dev_guard((h)->device)
}  // namespace

extern "C" {

int actmi_version(void) { return ACTMI_VERSION; }

int actmi_create(const actmi_config* cfg, actmi_handle* out) {
    try {
        return engine_create(cfg, out);
    } catch (const std::exception& e) {
        return ACTMI_E_NOMEM;
    }
}

int actmi_destroy(actmi_handle h) {
    if (!h) return 0;
    DevGuard g(h->device);
    return engine_destroy(h);
}

const char* actmi_last_error(actmi_handle h) { return h ? h->err.c_str() : engine_create_error(); }

int actmi_num_params(actmi_handle h) { return h ? (int)h->params.size() : ACTMI_E_INVALID; }

int actmi_param_info(actmi_handle h, int index, const char** key, int64_t* shape4, int* ndim, int* is_buffer) {
    if (!h || index < 0 || index >= (int)h->params.size()) return ACTMI_E_INVALID;
    const Param& p = h->params[index];
    if (key) *key = p.key.c_str();
    if (ndim) *ndim = (int)p.shape.size();
    if (shape4) for (size_t i = 0; i < 4; ++i) shape4[i] = i < p.shape.size() ? p.shape[i] : 1;
    if (is_buffer) *is_buffer = p.is_buffer ? 1 : 0;
    return 0;
}

int actmi_set_param(actmi_handle h, const char* key, const void* src, const int64_t* shape, int ndim, int is_device) {
    if (!h || !key || !src) return ACTMI_E_INVALID;
    ENTER(h);
    auto it = h->index.find(key);
    if (it == h->index.end()) return bad(h, std::string("unknown state_dict key: ") + key);
    const Param& p = h->params[it->second];
    if (shape) {
        if (ndim != (int)p.shape.size()) return bad(h, std::string("rank mismatch for ") + key);
        for (int i = 0; i < ndim; ++i)
            if (shape[i] != p.shape[i]) return bad(h, std::string("shape mismatch for ") + key);
    }
    // ordered against work in flight on any stream (a parameter must not change under a running kernel)
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(h->pbase + p.off, src, p.numel * sizeof(float),
                             is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
    if (e != hipSuccess) return bad(h, std::string("hipMemcpy: ") + hipGetErrorString(e), ACTMI_E_LAUNCH);
    h->finalized = false;
    return 0;
}

int actmi_get_param(actmi_handle h, const char* key, void* dst, int64_t nbytes, int is_device) {
    if (!h || !key || !dst) return ACTMI_E_INVALID;
    ENTER(h);
    auto it = h->index.find(key);
    if (it == h->index.end()) return bad(h, std::string("unknown state_dict key: ") + key);
    const Param& p = h->params[it->second];
    if (nbytes != p.numel * (int64_t)sizeof(float)) return bad(h, std::string("size mismatch for ") + key);
    // ordered against an optimizer step in flight on any stream (no checkpoint of half-updated weights)
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(dst, h->pbase + p.off, nbytes, is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost);
    if (e != hipSuccess) return bad(h, std::string("hipMemcpy: ") + hipGetErrorString(e), ACTMI_E_LAUNCH);
    return 0;
}

int actmi_param_ptr(actmi_handle h, const char* key, void** dev_ptr, int64_t* numel) {
    if (!h || !key) return ACTMI_E_INVALID;
    ENTER(h);
    auto it = h->index.find(key);
    if (it == h->index.end()) return bad(h, std::string("unknown state_dict key: ") + key);
    const Param& p = h->params[it->second];
    if (dev_ptr) *dev_ptr = h->pbase + p.off;
    if (numel) *numel = p.numel;
    return 0;
}

int actmi_finalize(actmi_handle h, void* stream) {
    if (!h) return ACTMI_E_INVALID;
    ENTER(h);
    return engine_finalize(h, S(stream));
}

int actmi_forward_infer(actmi_handle h, const float* qpos, const void* image, int image_fmt, int B, float* a_hat,
                        void* stream) {
    if (!h) return ACTMI_E_INVALID;
    ENTER(h);
    if (!qpos || !image || !a_hat) return bad(h, "null pointer");
    return engine_forward_infer(h, qpos, image, image_fmt, B, a_hat, S(stream), nullptr);
}

int actmi_set_forward_phase(actmi_handle h, int phase) {
    if (!h) return ACTMI_E_INVALID;
    if (phase < 0 || phase > 2) { h->err = "forward phase must be 0 (whole), 1 (trunk) or 2 (transformer)"; return ACTMI_E_INVALID; }
    h->fwd_phase = phase;
    return ACTMI_OK;
}

int actmi_forward_infer_vq(actmi_handle h, const float* qpos, const void* image, int image_fmt, int B,
                           const float* vq_sample, float* a_hat, void* stream) {
    if (!h) return ACTMI_E_INVALID;
    ENTER(h);
    if (!qpos || !image || !a_hat || !vq_sample) return bad(h, "null pointer");
    if (!h->cfg.vq) return bad(h, "handle was not created with vq = 1");
    return engine_forward_infer(h, qpos, image, image_fmt, B, a_hat, S(stream), vq_sample);
}

int actmi_forward_train(actmi_handle h, const float* qpos, const void* image, int image_fmt, const float* actions,
                        const uint8_t* is_pad, const float* eps, uint64_t dropout_seed, float dropout_p, int B, float* losses,
                        float* a_hat, float* mu, float* logvar, void* stream) {
    if (!h) return ACTMI_E_INVALID;
    ENTER(h);
    if (!qpos || !image || !actions || !is_pad) return bad(h, "null pointer");
    return train_forward(h, qpos, image, image_fmt, actions, is_pad, eps, dropout_seed, dropout_p, B, losses, a_hat, mu, logvar,
                         S(stream));
}
int actmi_backward(actmi_handle h, float loss_scale, void* stream) {
    if (!h) return ACTMI_E_INVALID;
    ENTER(h);
    return train_backward(h, loss_scale, S(stream));
}
int actmi_zero_grad(actmi_handle h, void* stream) {
    if (!h) return ACTMI_E_INVALID;
    ENTER(h);
    return train_zero_grad(h, S(stream));
}
int actmi_adamw_step(actmi_handle h, float lr, float lr_backbone, float weight_decay, float beta1, float beta2, float eps,
                     int64_t step, void* stream) {
    if (!h) return ACTMI_E_INVALID;
    ENTER(h);
    return train_adamw_step(h, lr, lr_backbone, weight_decay, beta1, beta2, eps, step, S(stream));
}
int actmi_adamw_step_range(actmi_handle h, float lr, float lr_backbone, float weight_decay, float beta1, float beta2, float eps,
                           int64_t step, int64_t offset, int64_t count, void* stream) {
    if (!h) return ACTMI_E_INVALID;
    ENTER(h);
    return train_adamw_range(h, lr, lr_backbone, weight_decay, beta1, beta2, eps, step, offset, count, S(stream));
}
int actmi_refresh_weights(actmi_handle h, void* stream) {
    if (!h) return ACTMI_E_INVALID;
    ENTER(h);
    if (!h->finalized) return bad(h, "refresh_weights before finalize", ACTMI_E_STATE);
    return engine_prepare_weights(h, S(stream), true);
}
int actmi_param_arena(actmi_handle h, void** dev_ptr, int64_t* nfloats) {
    if (!h) return ACTMI_E_INVALID;
    if (dev_ptr) *dev_ptr = h->pbase;
    if (nfloats) *nfloats = h->ptotal;
    return 0;
}
int actmi_grad_ptr(actmi_handle h, const char* key, void** dev_ptr, int64_t* numel) {
    if (!h || !key) return ACTMI_E_INVALID;
    ENTER(h);
    if (!h->train) return bad(h, "handle was created without enable_training", ACTMI_E_STATE);
    auto it = h->index.find(key);
    if (it == h->index.end()) return bad(h, std::string("unknown state_dict key: ") + key);
    const Param& p = h->params[it->second];
    if (dev_ptr) *dev_ptr = h->train->gbase + p.off;
    if (numel) *numel = p.numel;
    return 0;
}

int actmi_grad_arena(actmi_handle h, void** dev_ptr, int64_t* nfloats) {
    if (!h) return ACTMI_E_INVALID;
    ENTER(h);
    if (!h->train) return bad(h, "handle was created without enable_training", ACTMI_E_STATE);
    if (dev_ptr) *dev_ptr = h->train->gbase;
    if (nfloats) *nfloats = h->ptotal;
    return 0;
}

int actmi_grad_phase_range(actmi_handle h, int phase, int64_t* offset, int64_t* count) {
    if (!h || !offset || !count || (phase != 1 && phase != 2)) return ACTMI_E_INVALID;
    ENTER(h);
    if (!h->train) return bad(h, "handle was created without enable_training", ACTMI_E_STATE);
    // phase 1 = pos_table + transformer.* (registration order puts them first); everything after is phase 2
    int64_t split = h->ptotal;
    for (const Param& p : h->params)
        if (p.key != "pos_table" && p.key.rfind("transformer.", 0) != 0) { split = p.off; break; }
    *offset = phase == 1 ? 0 : split;
    *count = phase == 1 ? split : h->ptotal - split;
    return 0;
}

int actmi_wait_grad_phase(actmi_handle h, int phase, void* stream) {
    if (!h || phase != 1) return ACTMI_E_INVALID;
    ENTER(h);
    if (!h->train || !h->train->ev_phase1) return bad(h, "handle was created without enable_training", ACTMI_E_STATE);
    hipError_t e = hipStreamWaitEvent(S(stream), h->train->ev_phase1, 0);
    if (e != hipSuccess) return bad(h, std::string("hipStreamWaitEvent: ") + hipGetErrorString(e), ACTMI_E_LAUNCH);
    return 0;
}

int actmi_ensemble_step(float* ring, int32_t* tcount, const float* chunk, double k, double* out, uint8_t* populated, int E,
                        int Q, int A, void* stream) {
    if (!ring || !tcount || !chunk || !out || Q < 1 || A < 1 || A > 64) return ACTMI_E_INVALID;
    return launch_ensemble(ring, tcount, chunk, k, out, populated, E, Q, A, S(stream));
}

int actmi_op_gemm(const actmi_gemm_desc* d, void* stream) {
    if (!d) return ACTMI_E_INVALID;
    g_op_error.clear();
    return launch_gemm(*d, S(stream), &g_op_error);
}

int actmi_op_split16(const float* src, float* dst, int64_t nfloats, float scale, void* stream) {
    g_op_error.clear();
    const int rc = launch_split16(src, dst, nfloats, scale, S(stream));
    if (rc != 0) g_op_error = "split16: nfloats must be a multiple of 4 and both pointers 16-byte aligned";
    return rc == 0 ? 0 : ACTMI_E_INVALID;
}

int actmi_op_permute_conv_k(const float* src, float* dst, int64_t rows, int taps, int cin, int ld, void* stream) {
    g_op_error.clear();
    const int rc = launch_permute_conv_k(src, dst, rows, taps, cin, ld, S(stream));
    if (rc != 0) g_op_error = "permute_conv_k: cin must be a multiple of 32, ld >= taps*cin, src != dst";
    return rc == 0 ? 0 : (rc == -2 ? ACTMI_E_INVALID : ACTMI_E_LAUNCH);
}




int actmi_op_pow2_scale(const float* x, int64_t ld, int M, int N, float* out, void* stream) {
    g_op_error.clear();
    return launch_pow2_scale(x, ld, M, N, out, S(stream)) == 0 ? 0 : ACTMI_E_LAUNCH;
}

int actmi_op_splitk_combine(const float* part, int nsplit, int64_t split_stride, int64_t ldp, int M, int N, const float* scale,
                            const float* bias, const float* res, int64_t ldres, int relu, float* out, int64_t ldc, void* stream) {
    g_op_error.clear();
    if (!part || !out || nsplit < 1 || M < 0 || N < 0 || ldp < N || ldc < N || (res && ldres < N) || relu < 0 || relu > 2) {
        g_op_error = "splitk_combine: bad argument";
        return ACTMI_E_INVALID;
    }
    SplitCombineArgs c{};
    c.part = part; c.nsplit = nsplit; c.split_stride = split_stride; c.ldp = ldp;
    c.scale = scale; c.bias = bias; c.res = res; c.ldres = ldres; c.relu = relu;
    c.C = out; c.ldc = ldc; c.M = M; c.N = N; c.groups = 1;
    return launch_splitk_combine(c, S(stream)) == 0 ? 0 : ACTMI_E_LAUNCH;
}

int actmi_op_sample_onehot(const float* logits, int n, int V, float temperature, uint64_t seed, float* probs, float* code,
                           void* stream) {
    g_op_error.clear();
    if (!logits || !code || n < 1 || V < 1 || !(temperature > 0.f)) { g_op_error = "sample_onehot: bad argument"; return ACTMI_E_INVALID; }
    return launch_vq_code(logits, nullptr, seed, probs, code, n, 1, V, S(stream), temperature) == 0 ? 0 : ACTMI_E_LAUNCH;
}

int actmi_op_attention(const actmi_attn_desc* d, void* stream) {
    if (!d) return ACTMI_E_INVALID;
    g_op_error.clear();
    return launch_attention(*d, S(stream), &g_op_error);
}

int actmi_op_attention_bwd(const actmi_attn_bwd_desc* d, void* stream) {
    g_op_error.clear();
    if (!d || !d->o || !d->delta_ws) { g_op_error = "attention_bwd: null descriptor / output / scratch"; return ACTMI_E_INVALID; }
    const int64_t D = (int64_t)d->H * d->HD;
    if (launch_attn_delta(d->d_o, d->o, d->delta_ws, d->B, d->H, d->Nq, d->HD, S(stream)) != 0) { g_op_error = "attention_bwd: delta launch failed"; return ACTMI_E_LAUNCH; }
    AttnBwdArgs a{};
    a.Q = d->q; a.K = d->k; a.V = d->v; a.dO = d->d_o; a.lse = d->lse; a.delta = d->delta_ws; a.dO_scale = d->do_scale;
    a.dQ = d->dq; a.dK = d->dk; a.dV = d->dv;
    a.q_bs = d->q_bs; a.q_rs = d->q_rs; a.k_bs = d->k_bs; a.k_rs = d->k_rs; a.v_bs = d->v_bs; a.v_rs = d->v_rs;
    a.do_bs = (int64_t)d->Nq * D; a.do_rs = D;
    a.dq_bs = d->dq_bs; a.dq_rs = d->dq_rs; a.dk_bs = d->dk_bs; a.dk_rs = d->dk_rs; a.dv_bs = d->dv_bs; a.dv_rs = d->dv_rs;
    a.kpm = d->kpm; a.kpm_bs = d->kpm_bs;
    a.B = d->B; a.H = d->H; a.Nq = d->Nq; a.Nk = d->Nk; a.HD = d->HD;
    a.scale = 1.0f / sqrtf((float)d->HD); a.drop_p = d->drop_p; a.drop_seed = d->drop_seed; a.amax_out = d->amax_out;
    const int rc = launch_attention_bwd(a, S(stream), &g_op_error);
    return rc == 0 ? ACTMI_OK : (rc == -2 ? ACTMI_E_SHAPE : ACTMI_E_LAUNCH);
}

int actmi_op_layernorm(const float* x, const float* res, int res_mod, const float* w, const float* b, const float* w2,
                       const float* b2, float* y, int M, int D, float eps, void* stream) {
    g_op_error.clear();
    return launch_layernorm(x, res, res_mod, w, b, w2, b2, y, M, D, eps, S(stream), &g_op_error);
}

int actmi_op_maxpool3x3s2(const float* in, float* out, int nimg, int H, int W, int C, void* stream) {
    return launch_maxpool(in, out, nimg, H, W, C, (H + 2 - 3) / 2 + 1, (W + 2 - 3) / 2 + 1, S(stream));
}

int actmi_op_conv1(const void* image, int image_fmt, const float* w_oihw, const float* scale, const float* bias,
                   float* out, float* workspace, int B, int C, int H, int W, int Cout, int prec, void* stream) {
    g_op_error.clear();
    float* wp = workspace;
    float* lut = workspace + (int64_t)C * Cout * 148;
    int rc = launch_repack_conv_w(w_oihw, wp, C, Cout, 3, 7, 7, (int64_t)Cout * 147, (int64_t)Cout * 148, 148, S(stream));
    if (rc) return rc;
    float hl[768];
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    for (int c = 0; c < 3; ++c)
        for (int v = 0; v < 256; ++v) hl[c * 256 + v] = ((float)((double)v / 255.0) - mean[c]) / stdv[c];
    if (hipMemcpyAsync(lut, hl, sizeof(hl), hipMemcpyHostToDevice, S(stream)) != hipSuccess) return ACTMI_E_LAUNCH;
    if (hipStreamSynchronize(S(stream)) != hipSuccess) return ACTMI_E_LAUNCH;
    Conv1Args a;
    a.image = image; a.fmt = image_fmt; a.lut = lut; a.w = wp; a.scale = scale; a.bias = bias; a.out = out;
    a.B = B; a.C = C; a.H = H; a.W = W; a.Ho = (H + 6 - 7) / 2 + 1; a.Wo = (W + 6 - 7) / 2 + 1; a.Cout = Cout;
    a.prec = prec;
    return launch_conv1(a, S(stream), &g_op_error);
}

// workspace layout of the prepared stem: [C*Cout*148 repacked weights][768 lut][C*Cout ones][C*Cout zeros][C * wimg bytes]
static int64_t conv1_ws_off_lut(int C, int Cout) { return (int64_t)C * Cout * 148; }
int64_t actmi_op_conv1_workspace_floats(int C, int Cout) {
    return conv1_ws_off_lut(C, Cout) + 768 + 2 * (int64_t)C * Cout + (int64_t)C * ((conv1_wimg_bytes() + 3) / 4) + 16;
}
int actmi_op_conv1_prepare(const float* w_oihw, float* workspace, int C, int Cout, int lut_mode, void* stream) {
    g_op_error.clear();
    if (!w_oihw || !workspace || C < 1 || Cout < 1 || Cout > 64) { g_op_error = "conv1_prepare: bad argument"; return ACTMI_E_INVALID; }
    float* lut = workspace + conv1_ws_off_lut(C, Cout);
    float* ones = lut + 768;
    float* zeros = ones + (int64_t)C * Cout;
    float* wimg = zeros + (int64_t)C * Cout;
    wimg = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(wimg) + 15) & ~(uintptr_t)15);
    int rc = launch_repack_conv_w(w_oihw, workspace, C, Cout, 3, 7, 7, (int64_t)Cout * 147, (int64_t)Cout * 148, 148, S(stream));
    if (rc) return ACTMI_E_LAUNCH;
    std::vector<float> host(768 + 2 * (size_t)C * Cout);
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    for (int c = 0; c < 3; ++c)
        for (int v = 0; v < 256; ++v) {
            const float x = (float)((double)v / 255.0);
            host[c * 256 + v] = lut_mode == 0 ? (x - mean[c]) / stdv[c] : x;
        }
    for (int i = 0; i < C * Cout; ++i) { host[768 + i] = 1.f; host[768 + C * Cout + i] = 0.f; }
    if (hipMemcpyAsync(lut, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice, S(stream)) != hipSuccess) return ACTMI_E_LAUNCH;
    rc = launch_conv1_wimg(workspace, wimg, C, Cout, S(stream), 256.f);
    if (rc) return ACTMI_E_LAUNCH;
    if (hipStreamSynchronize(S(stream)) != hipSuccess) return ACTMI_E_LAUNCH;
    return 0;
}
int actmi_op_conv1_prepared(const void* image_u8, const float* workspace, const float* scale, const float* bias, float* out, int B,
                            int C, int H, int W, int Cout, int relu, void* stream) {
    g_op_error.clear();
    if (!image_u8 || !workspace || !out) { g_op_error = "conv1_prepared: null pointer"; return ACTMI_E_INVALID; }
    const float* lut = workspace + conv1_ws_off_lut(C, Cout);
    const float* ones = lut + 768;
    const float* zeros = ones + (int64_t)C * Cout;
    const float* wimg = zeros + (int64_t)C * Cout;
    wimg = reinterpret_cast<const float*>((reinterpret_cast<uintptr_t>(wimg) + 15) & ~(uintptr_t)15);
    Conv1Args a;
    a.image = image_u8; a.fmt = ACTMI_IMG_U8_NHWC; a.lut = lut; a.w = workspace; a.scale = scale ? scale : ones; a.bias = bias ? bias : zeros;
    a.out = out; a.B = B; a.C = C; a.H = H; a.W = W; a.Ho = (H + 6 - 7) / 2 + 1; a.Wo = (W + 6 - 7) / 2 + 1; a.Cout = Cout;
    a.prec = ACTMI_PREC_F16X3; a.wimg = reinterpret_cast<const unsigned char*>(wimg); a.wscale = 256.f;
    a.relu_floor = relu ? 0.f : -__builtin_inff();
    return launch_conv1(a, S(stream), &g_op_error);
}

int actmi_op_conv3x3_c64(const float* x, const float* w16, float w_scale, const float* scale, const float* bias,
                         const float* res, float* out, int G, int B, int H, int W, int relu, void* stream) {
    g_op_error.clear();
    Conv3Args a;
    a.x = x; a.w16 = w16; a.scale = scale; a.bias = bias; a.res = res; a.out = out;
    a.G = G; a.B = B; a.H = H; a.W = W; a.relu = relu; a.w_scale = w_scale;
    return launch_conv3x3_c64(a, S(stream), &g_op_error);
}


int actmi_op_wgrad3x3_c64(const float* dy, const float* x, float* dw, float* ws, int64_t ws_floats, const float* dy_scale_dev, int G,
                          int B, int H, int W, void* stream) {
    g_op_error.clear();
    int nwg = 0;
    int rc = launch_wgrad3x3_c64(dy, x, ws, ws_floats, dy_scale_dev, G, B, H, W, &nwg, S(stream));
    if (rc != 0) { g_op_error = "wgrad3x3_c64: bad arguments or launch failure"; return rc == -2 ? ACTMI_E_INVALID : ACTMI_E_LAUNCH; }
    SplitCombineArgs c{};
    const int64_t slice = (int64_t)64 * 576;
    c.part = ws; c.nsplit = nwg; c.split_stride = slice; c.gP = slice * nwg; c.ldp = 576;
    c.C = dw; c.ldc = 576; c.gC = slice; c.M = 64; c.N = 576; c.groups = G;
    rc = launch_splitk_combine(c, S(stream));
    if (rc != 0) { g_op_error = "wgrad3x3_c64: combine launch failed"; return ACTMI_E_LAUNCH; }
    return 0;
}

int actmi_op_wgrad7x7s2(const float* dy, const float* x4, float* dw, float* ws, int64_t ws_floats, const float* dy_scale_dev, int G,
                        int B, int H, int W, void* stream) {
    g_op_error.clear();
    int nwg = 0;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    int rc = launch_wgrad7x7s2(dy, x4, ws, ws_floats, dy_scale_dev, G, B, H, W, Ho, Wo, &nwg, S(stream));
    if (rc != 0) { g_op_error = "wgrad7x7s2: bad arguments or launch failure"; return rc == -2 ? ACTMI_E_INVALID : ACTMI_E_LAUNCH; }
    SplitCombineArgs c{};
    const int64_t slice = (int64_t)64 * 196;
    c.part = ws; c.nsplit = nwg; c.split_stride = slice; c.gP = slice * nwg; c.ldp = 196;
    c.C = dw; c.ldc = 196; c.gC = slice; c.M = 64; c.N = 196; c.groups = G;
    rc = launch_splitk_combine(c, S(stream));
    if (rc != 0) { g_op_error = "wgrad7x7s2: combine launch failed"; return ACTMI_E_LAUNCH; }
    return 0;
}

const char* actmi_op_last_error(void) { return g_op_error.c_str(); }

int actmi_set_gemm_prec(actmi_handle h, int prec) {
    if (!h) return ACTMI_E_INVALID;
    ENTER(h);
    if (prec != ACTMI_PREC_F32 && prec != ACTMI_PREC_F16X3) { h->err = "prec must be ACTMI_PREC_F32 or ACTMI_PREC_F16X3"; return ACTMI_E_INVALID; }
    h->gemm_prec = prec;
    h->finalized = false;
    return 0;
}

int actmi_set_train_prec(actmi_handle h, int prec) {
    if (!h) return ACTMI_E_INVALID;
    ENTER(h);
    if (prec != 0 && prec != ACTMI_PREC_BF16 && prec != ACTMI_PREC_F16X3 && prec != ACTMI_PREC_F32) { h->err = "train prec must be 0 or ACTMI_PREC_*"; return ACTMI_E_INVALID; }
    h->train_prec = (prec == h->gemm_prec) ? 0 : prec;
    if (h->train_prec != 0 && h->train_prec != ACTMI_PREC_BF16) { h->err = "only ACTMI_PREC_BF16 differs from the handle precision"; h->train_prec = 0; return ACTMI_E_INVALID; }
    return 0;
}

int actmi_get_flags(actmi_handle h, uint32_t* host_flags, int clear, void* stream) {
    if (!h || !host_flags) return ACTMI_E_INVALID;
    ENTER(h);
    hipError_t e = hipMemcpyAsync(host_flags, h->flags, sizeof(uint32_t), hipMemcpyDeviceToHost, S(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(S(stream));
    if (e == hipSuccess && clear && *host_flags) e = hipMemsetAsync(h->flags, 0, sizeof(uint32_t), S(stream));
    if (e != hipSuccess) return bad(h, std::string("get_flags: ") + hipGetErrorString(e), ACTMI_E_LAUNCH);
    return 0;
}

int actmi_flags_ptr(actmi_handle h, void** dev_ptr) {
    if (!h || !dev_ptr) return ACTMI_E_INVALID;
    *dev_ptr = h->flags;
    return 0;
}

int actmi_debug_stop_after(actmi_handle h, const char* stage) {
    if (!h) return ACTMI_E_INVALID;
    h->stop_stage = stage ? stage : "";
    return 0;
}

int actmi_debug_tensor(actmi_handle h, const char* name, const float** dev_ptr, int64_t* numel) {
    if (!h || !name) return ACTMI_E_INVALID;
    ENTER(h);
    auto it = h->dbg.find(name);
    if (it == h->dbg.end()) return bad(h, std::string("no debug tensor ") + name);
    if (dev_ptr) *dev_ptr = it->second.ptr;
    if (numel) *numel = it->second.numel;
    return 0;
}

// ---- ops of the latent-prior training step (prior.hip; the LayerNorm backward, column sum and batch sum are the training
// engine's own kernels) ------------------------------------------------------------------------------------------
#define OPCHK(cond, msg) do { g_op_error.clear(); if (!(cond)) { g_op_error = msg; return ACTMI_E_INVALID; } } while (0)
#define OPRC(call, what) do { const int rc_ = (call); if (rc_ != 0) { g_op_error = what; return rc_ == -2 ? ACTMI_E_SHAPE : ACTMI_E_LAUNCH; } return ACTMI_OK; } while (0)

int actmi_op_gelu(const float* x, float* y, int64_t n, void* stream) {
    OPCHK(x && y && n >= 0, "gelu: bad argument");
    OPRC(launch_gelu(x, y, n, S(stream)), "gelu launch failed");
}
int actmi_op_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
    OPCHK(x && dy && dx && n >= 0, "gelu_bwd: bad argument");
    OPRC(launch_gelu_bwd(x, dy, dx, n, S(stream)), "gelu_bwd launch failed");
}
int actmi_op_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream) {
    OPCHK(x && y && n >= 0 && p >= 0.f && p < 1.f, "dropout: bad argument (0 <= p < 1)");
    OPRC(launch_dropout(x, y, n, p, seed, S(stream)), "dropout launch failed");
}
int actmi_op_small_attention(const float* qkv, float* out, int n, int T, int H, int HD, int causal, float drop_p, uint64_t seed,
                             void* stream) {
    OPCHK(qkv && out, "small_attention: null pointer");
    OPRC(launch_small_attention(qkv, out, n, T, H, HD, causal, drop_p, seed, S(stream)),
         "small_attention: needs 1 <= T <= 64, 1 <= head_dim <= 64, 0 <= drop_p < 1");
}
int actmi_op_small_attention_bwd(const float* qkv, const float* dout, float* dqkv, int n, int T, int H, int HD, int causal,
                                 float drop_p, uint64_t seed, void* stream) {
    OPCHK(qkv && dout && dqkv, "small_attention_bwd: null pointer");
    OPRC(launch_small_attention_bwd(qkv, dout, dqkv, n, T, H, HD, causal, drop_p, seed, S(stream)),
         "small_attention_bwd: needs 1 <= T <= 64, 1 <= head_dim <= 64, 0 <= drop_p < 1");
}
int actmi_op_soft_ce_dim1(const float* logits, const float* target, int B, int T, int V, float* loss, float* dlogits, float* ws,
                          void* stream) {
    OPCHK(logits && target && loss && ws, "soft_ce_dim1: null pointer");
    OPRC(launch_soft_ce_dim1(logits, target, B, T, V, loss, dlogits, ws, S(stream)), "soft_ce_dim1: bad shape");
}
int actmi_op_argmax_l1(const float* logits, const float* target, int rows, int V, float* out, float* ws, void* stream) {
    OPCHK(logits && target && out && ws, "argmax_l1: null pointer");
    OPRC(launch_argmax_l1(logits, target, rows, V, out, ws, S(stream)), "argmax_l1: bad shape");
}
int actmi_op_layernorm_bwd(const float* x, const float* w, const float* dy, const float* dx_add, float* dx, float* dw, float* db,
                           int M, int D, float eps, float* ws, int64_t ws_floats, void* stream) {
    OPCHK(x && w && dy && dx && dw && db && M >= 0, "layernorm_bwd: null pointer");
    OPRC(launch_ln_bwd(x, w, dy, dx_add, dx, dw, db, M, D, eps, S(stream), ws, ws_floats), "layernorm_bwd: D must be a multiple of 4 and <= 2048");
}
int actmi_op_colsum(const float* src, int64_t ld, float* out, int M, int N, float* ws, int64_t ws_floats, void* stream) {
    OPCHK(src && out && M >= 0 && N >= 0, "colsum: bad argument");
    OPRC(launch_colsum(src, ld, out, M, N, S(stream), ws, ws_floats), "colsum launch failed");
}
int actmi_op_sum_batch(const float* src, int64_t batch_stride, int64_t ld, float* dst, int B, int R, int D, int accumulate,
                       void* stream) {
    OPCHK(src && dst && B >= 1, "sum_batch: bad argument");
    OPRC(launch_sum_batch(src, batch_stride, ld, dst, B, R, D, accumulate, S(stream)), "sum_batch launch failed");
}
int actmi_op_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float weight_decay, float beta1, float beta2,
                   float eps, int64_t step, void* stream) {
    OPCHK(p && g && m && v && n >= 0 && step >= 1, "adamw: bad argument (step counts from 1)");
    OPRC(launch_adamw_flat(p, g, m, v, n, lr, weight_decay, beta1, beta2, eps, step, S(stream)), "adamw launch failed");
}
#undef OPCHK
#undef OPRC

}  // extern "C"
