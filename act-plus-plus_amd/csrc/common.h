// Shared declarations for libactmi (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <cstring>
#include "actmi.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef __HIPCC__
// Running maximum of |x| bits shared by a whole launch (the operand-scale slots of the backward pass): `v` is the calling
// wave's maximum, already reduced over its lanes; call from ONE lane.  Device-scope atomics resolve at the memory side and
// serialise on the address (~6 ns each: 150k waves cost a millisecond), so a wave first looks at the current value with a
// device-scope load and only issues the atomic when it would raise it -- after the first few waves almost none do.
__device__ __forceinline__ void amax_commit(unsigned* bits, unsigned v) {
    if (v > __hip_atomic_load(bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(bits, v);
}
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ACTMI_WAVE 64

struct ActmiError {
    int code;
    std::string msg;
};

// ---- GEMM / implicit-GEMM convolution (gemm.hip): descriptor is part of the C ABI (include/actmi.h)
typedef actmi_gemm_desc GemmArgs;

int launch_gemm(const GemmArgs& a, hipStream_t st, std::string* err);

// sums the slices of a sliced split-K product (GemmArgs::split_stride) in split order and applies the forward epilogue
// v = sum * scale[n] + bias[n] (+ res[m][n]) -> ReLU / GELU (misc.hip)
struct SplitCombineArgs {
    const float* part;        // slice s of group g: part + g*gP + s*split_stride, rows of ldp floats
    int nsplit;
    int64_t split_stride, gP, ldp;
    const float* scale;       // per-n or NULL (group stride gSB)
    const float* bias;
    int64_t gSB;
    const float* res;         // optional [M][ldres] (group stride gRes)
    int64_t ldres, gRes;
    int relu;                 // 0 none, 1 ReLU, 2 GELU
    float* C;
    int64_t ldc, gC;
    int M, N, groups;
};
int launch_splitk_combine(const SplitCombineArgs& a, hipStream_t st);

// ---- conv1 7x7/s2 + FrozenBN + ReLU (conv1.hip) -----------------------------------------------
struct Conv1Args {
    const void* image;    // u8 NHWC [B][C][H][W][3] or f32 NCHW [B][C][3][H][W]
    int fmt;              // 0 = U8_NHWC, 1 = F32_NCHW
    const float* lut;     // [3][256] normalised values for u8 input
    const float* w;       // [C][Cout][KPAD] (r,s,c) order, zero padded
    const float* scale;   // [C][Cout]
    const float* bias;    // [C][Cout]
    float* out;           // camera-major NHWC [C][B][Ho][Wo][Cout]
    int B, C, H, W, Ho, Wo, Cout;
    int prec = 0;         // ACTMI_PREC_* (0 = environment / native fp32)
    const unsigned char* wimg = nullptr;   // f16x3 only, optional: launch_conv1_wimg's image of w (C x conv1_wimg_bytes())
    int vpool = 0;        // f16x3 only: write max over conv rows (2a-1, 2a, 2a+1) -> out [C][B][Ho/2][Wo][Cout] (pool's vertical half)
    int vpool_nseg = 1;   // set by the launcher
    float wscale = 256.f; // f16x3 only: power of two the split weight image was built with (undone in the epilogue)
    int cam0 = 0, ncam = 0;   // camera range of this launch (ncam = 0: all C); image / w / scale / bias / out stay whole-tensor pointers
    float relu_floor = 0.f;   // f16x3, plain (non-vpool) form: out = max(acc * scale + bias, relu_floor); -inf = no ReLU
};
int64_t conv1_wimg_bytes();
int launch_conv1_wimg(const float* w, void* img, int C, int Cout, hipStream_t st, float wscale = 256.f);
int launch_conv1(const Conv1Args& a, hipStream_t st, std::string* err);

// ---- direct 3x3 / stride 1 / pad 1 convolution, 64 -> 64 channels, f16x3 (conv3.hip) ----------
struct Conv3Args {
    const float* x;       // camera-major NHWC [G][B][H][W][64]
    const float* w16;     // fp16-split image of the weights [G][64 cout][(r,s,c) = 576], built with scale w_scale
    const float* scale;   // [G][64] folded FrozenBN
    const float* bias;    // [G][64]
    const float* res;     // optional residual, same shape as out
    float* out;           // [G][B][H][W][64]
    int G, B, H, W, relu;
    float w_scale;
    // data-gradient use (training): the input is a gradient map with a device-side power-of-two scale (applied before the
    // fp16 split, undone in the result); the result is zeroed where mask <= 0 (ReLU backward; same shape as out) and then
    // multiplied by post_scale[G][64] (the FrozenBN scale of the layer below); amax_out collects the bits of max |out|
    const float* x_scale_dev = nullptr;
    const float* mask = nullptr;
    const float* post_scale = nullptr;
    unsigned* amax_out = nullptr;
};
int launch_conv3x3_c64(const Conv3Args& a, hipStream_t st, std::string* err);

// ---- 3x3/s2/p1 max pool NHWC (pool.hip) -----------------------------------------------------
int launch_maxpool(const float* in, float* out, int nimg, int H, int W, int C, int Ho, int Wo, hipStream_t st);
// horizontal half of the 3x3/s2/p1 pool on a vertically pooled map: out[h][pw] = max(in[h][2pw-1 .. 2pw+1])
int launch_hpool(const float* in, float* out, int nrows, int W, int C, int Wo, hipStream_t st);
int launch_maxpool_idx(const float* in, float* out, uint8_t* arg, int nimg, int H, int W, int C, int Ho, int Wo, hipStream_t st);
int launch_maxpool_bwd_idx(const uint8_t* arg, const float* dy, float* dx, int nimg, int H, int W, int C, int Ho, int Wo,
                           hipStream_t st, const float* relu_x = nullptr, const float* bn_scale = nullptr, int imgs_per_group = 1,
                           unsigned* amax_bits = nullptr);

// ---- LayerNorm (layernorm.hip) ----------------------------------------------------------------
// y = LN(x + res[m % res_mod]) * w + b ; optional second LN (w2,b2) applied on top.
// optional extra outputs computed from the finished row while it is still in registers
struct LnExtra {
    float* y2 = nullptr;             // y2[row] = y[row] + add2[row % add2_mod]  (add2_mod = 0: add2[row])
    const float* add2 = nullptr;
    int add2_mod = 0;
    float* head_out = nullptr;       // head_out[row][n] = y[row] . head_w[n] + head_b[n], n < head_n  (fp32 FMA)
    const float* head_w = nullptr;   // [head_n][D]
    const float* head_b = nullptr;   // [head_n] or NULL
    int head_n = 0;
    uint32_t* flag = nullptr;        // OR flag_bit into *flag when a head output is NaN / infinite
    uint32_t flag_bit = 0;
};
int launch_layernorm(const float* x, const float* res, int res_mod, const float* w, const float* b,
                     const float* w2, const float* b2, float* y, int M, int D, float eps, hipStream_t st,
                     std::string* err, int nsplit = 1, int64_t split_stride = 0, const float* bias = nullptr,
                     const LnExtra* extra = nullptr);

// ---- attention (attn.hip) -----------------------------------------------------------------------
typedef actmi_attn_desc AttnArgs;
int launch_attention(const AttnArgs& a, hipStream_t st, std::string* err);
// attention backward without materialised scores (attn_bwd.hip): dQ, dK, dV from Q, K, V, dO, the forward's log-sum-exp and
// delta[b][h][q] = dO[q] . O[q].  dO_scale (optional, device): the power of two dO is multiplied with on its way into fp16 pieces.
struct AttnBwdArgs {
    const float *Q, *K, *V, *dO, *lse, *delta, *dO_scale;
    float *dQ, *dK, *dV;
    int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs;
    const uint8_t* kpm; int64_t kpm_bs;
    int B, H, Nq, Nk, HD;
    float scale, drop_p; uint64_t drop_seed;
    unsigned* amax_out;            // optional: bits of the largest |value| written (integer atomicMax)
};
int launch_attention_bwd(const AttnBwdArgs& a, hipStream_t st, std::string* err);

// ---- small kernels (misc.hip) --------------------------------------------------------------------
int launch_small_linear(const float* x, int64_t ldx, const float* w, const float* b, float* y, int64_t ldy,
                        int M, int N, int K, hipStream_t st, float* fill_dst = nullptr, const float* fill_src = nullptr,
                        int64_t fill_src_bs = 0);
int launch_fill_rows(float* dst, int64_t ld, int64_t batch_stride, const float* src, int64_t src_bs, int B, int D,
                     hipStream_t st);
int launch_repack_conv_w(const float* w_oihw, float* w_ohwi, int G, int O, int I, int KH, int KW, int64_t g_in,
                         int64_t g_out, int kpad, hipStream_t st);
int launch_ensemble(float* ring, int* tcount, const float* chunk, double k, double* out, uint8_t* populated, int E,
                    int Q, int A, hipStream_t st);
int launch_bn_fold(const float* w, const float* b, const float* rm, const float* rv, float* scale, float* bias, int n,
                   hipStream_t st);
int launch_build_rowmap(int* map, int B, int C, int fh, int fw, int N, hipStream_t st);
int launch_permute_conv_k(const float* src, float* dst, int64_t rows, int taps, int cin, int ld, hipStream_t st);
// [s2 * w2 | sd * wd] per output row + summed bias: the fused weights of a block's conv2 + downsample (misc.hip)
int launch_fold_cat_w(const float* w2, const float* s2, const float* b2, const float* wd, const float* sd, const float* bd,
                      float* out, float* bias, int G, int N, int K2, int Kd, hipStream_t st);

// ---- per-launch event profiler (prof.hip) ------------------------------------------------------------
bool prof_enabled();
void prof_begin(const char* name, double flops, double bytes, hipStream_t st);
void prof_end(hipStream_t st);

// ---- backward-pass kernels (bwd.hip, misc.hip) ---------------------------------------------------------
// ws (optional): scratch for the deterministic form -- per-block dw / db partials summed in block order instead of float atomics
int launch_ln_bwd(const float* x, const float* w, const float* dy, const float* dx_add, float* dx, float* dw, float* db,
                  int M, int D, float eps, hipStream_t st, float* ws = nullptr, int64_t ws_floats = 0, unsigned* dx_amax = nullptr);
int launch_maxpool_bwd(const float* x, const float* dy, float* dx, int nimg, int H, int W, int C, int Ho, int Wo,
                       hipStream_t st);
int launch_colsum(const float* src, int64_t ld, float* out, int M, int N, hipStream_t st, float* ws = nullptr, int64_t ws_floats = 0);
int launch_sum_batch(const float* src, int64_t bs, int64_t ld, float* dst, int B, int R, int D, int accumulate,
                     hipStream_t st);
int launch_attn_delta(const float* dO, const float* O, float* delta, int B, int H, int Nq, int HD, hipStream_t st);
int launch_attn_probs(float* S, const float* lse, const uint8_t* kpm, int64_t kpm_bs, int G, int H, int Nq, int Nk, int ldp,
                      hipStream_t st);
int launch_zero_cols(float* x, int64_t rows, int ld, int c0, hipStream_t st);
int launch_attn_ds(const float* P, float* dP, const float* delta, float scale, int G, int Nq, int Nk, int ldp,
                   hipStream_t st, unsigned* amax_bits = nullptr);
// losses: [3] results followed by >= 513 floats of scratch (block partials of the l1 sum: fixed-order total)
int launch_losses(const float* a_hat, const float* actions, const uint8_t* is_pad, const float* latent_info, float* losses,
                  int B, int Q, int A, int L, float kl_weight, hipStream_t st);
int launch_bcast_add_rows(float* dst, const float* vec, int R, int D, hipStream_t st);
int launch_l1_bwd(const float* a_hat, const float* actions, const uint8_t* is_pad, float* d_a_hat, int B, int Q, int A,
                  float gscale, hipStream_t st);
int launch_reparam(const float* latent_info, const float* eps, float* z, float* mu_out, float* logvar_out, int B, int L,
                   hipStream_t st);
int launch_reparam_kl_bwd(const float* latent_info, const float* eps, const float* dz, float* d_latent_info, int B, int L,
                          float klw_scaled, hipStream_t st);
int launch_vq_code(const float* logits, const float* code_in, uint64_t seed, float* probs, float* code, int B, int VC, int VD,
                   hipStream_t st, float temperature = 1.f);
int launch_vq_bwd(const float* probs, const float* g, float* dlogits, int B, int VC, int VD, hipStream_t st);
int launch_adamw(float* p, const float* g, float* m, float* v, const uint8_t* group, int64_t n, float lr, float lr_bb,
                 float wd, float b1, float b2, float eps, int64_t step, hipStream_t st, const uint32_t* flags = nullptr,
                 uint32_t skip_mask = 0);
int launch_repack_dgrad_w(const float* wf, float* wd, int G, int O, int I, int KK, hipStream_t st, int flip = 0);
int launch_unpack_wgrad(const float* gp, float* g_oihw, int O, int I, int KH, int KW, int kpad, int ipack, hipStream_t st, int G = 1,
                        int64_t g_gp = 0, int64_t g_out = 0);
int launch_relu_bn_bwd(const float* x, const float* add, const float* mask, const float* scale, float* y_plain,
                       float* y_scaled, int G, int64_t per_group, int C, hipStream_t st, unsigned* amax_bits = nullptr);
int launch_normalize_pad(const void* image, int fmt, const float* lut, float* out, int B, int C, int H, int W,
                         hipStream_t st);
int launch_gather_rows(const float* src, const int* map, float* dst, int M, int D, hipStream_t st);
int launch_small_linear_wgrad(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dW, int M, int N, int K,
                              hipStream_t st);
int launch_cvae_maps(int* map, uint8_t* kpm, const uint8_t* is_pad, int B, int Q, hipStream_t st);
int launch_axpy(float* dst, const float* src, int64_t n, hipStream_t st);
int launch_scale(float* x, int64_t n, float s, hipStream_t st);
// direct weight gradient of the 64 -> 64 channel 3x3 / s1 / p1 convolutions (wgrad3.hip): per-workgroup partials
// [groups][*nwg_out][64][576] into ws, to be summed by launch_splitk_combine
int launch_wgrad3x3_c64(const float* dy, const float* x, float* ws, int64_t ws_floats, const float* dy_scale_dev, int groups, int B,
                        int H, int W, int* nwg_out, hipStream_t st);
// direct weight gradient of the stem (7x7 / s2 / p3, 4-channel-padded image -> 64 channels; wgrad7.hip): per-workgroup partials
// [groups][*nwg_out][64][196]
int launch_wgrad7x7s2(const float* dy, const float* x4, float* ws, int64_t ws_floats, const float* dy_scale_dev, int groups, int B,
                      int H, int W, int Ho, int Wo, int* nwg_out, hipStream_t st);
int launch_pow2_scale(const float* x, int64_t ld, int M, int N, float* out, hipStream_t st);
int launch_pow2_from_bits(float* out, hipStream_t st);      // the scale from bits a producing kernel left in out[1]
int launch_split16(const float* src, float* dst, int64_t nfloats, float scale, hipStream_t st, uint32_t* flag = nullptr);
// range guard of the f16x3 weight images (misc.hip): per-segment max |x| (as float bits), the parameter arena split with one
// power-of-two scale per parameter, and the finite check of an output
int launch_seg_amax(const float* base, const int64_t* off, const int64_t* numel, int nseg, unsigned* out_bits, hipStream_t st);
int launch_split16_map(const float* src, float* dst, int64_t nfloats, const int* seg_of_group64, const float* seg_scale,
                       uint32_t* flag, hipStream_t st);
int launch_check_finite(const float* x, int64_t n, uint32_t* flag, uint32_t bit, hipStream_t st);
int launch_dropout_bwd(const float* dy, float* dz, uint64_t seed, float p, int64_t n, hipStream_t st);
// ---- latent-prior training pieces (prior.hip) ----------------------------------------------------------------
int launch_gelu(const float* x, float* y, int64_t n, hipStream_t st);
int launch_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, hipStream_t st);
int launch_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, hipStream_t st);
int launch_small_attention(const float* qkv, float* out, int n, int T, int H, int HD, int causal, float drop_p, uint64_t seed,
                           hipStream_t st);
int launch_small_attention_bwd(const float* qkv, const float* dout, float* dqkv, int n, int T, int H, int HD, int causal, float drop_p,
                               uint64_t seed, hipStream_t st);
int launch_soft_ce_dim1(const float* logits, const float* target, int B, int T, int V, float* loss, float* dlogits, float* ws,
                        hipStream_t st);
int launch_argmax_l1(const float* logits, const float* target, int rows, int V, float* out, float* ws, hipStream_t st);
int launch_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float wd, float b1, float b2, float eps,
                      int64_t step, hipStream_t st);
int launch_attn_drop(const float* P, float* Pd, uint64_t seed, float p, int G, int Nq, int Nk, int ldp, hipStream_t st);
int launch_attn_ds_drop(const float* P, float* dP, const float* delta, float scale, uint64_t seed, float p, int G, int Nq,
                        int Nk, int ldp, hipStream_t st);
