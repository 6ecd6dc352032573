// Internal context of libactmi (not part of the C ABI).
#pragma once
#include "common.h"

#include <map>
#include <string>
#include <unordered_map>
#include <vector>

struct Param {
    std::string key;
    std::vector<int64_t> shape;
    int64_t numel;
    int64_t off;       // float offset into the parameter arena
    bool is_buffer;
};

// default power-of-two scale of a split weight image: the lo pieces of weights around 1e-2 stay normal fp16 numbers.  The
// handle replaces it per parameter at finalize (engine_calibrate_weight_scales) from the parameter's largest magnitude.
static constexpr float W16_SCALE = 256.f;
// flag word of a handle (actmi_get_flags): bit 0 = a forward output was not finite, bit 1 = a weight left the fp16 range of
// its split image, bit 2 = a training loss was not finite
// (values: ACTMI_FLAG_OUTPUT / _WEIGHT / _LOSS in include/actmi.h)

struct ConvLayer {
    std::string name, bn;
    int cin, cout, k, stride, pad, H, W, Ho, Wo;
    int K = 0;
    float* w = nullptr;       // [cam][cout][K], K index (r,s,c)
    float* w16 = nullptr;     // the same, fp16-split (f16x3 GEMM), built with w16_scale
    float w16_scale = W16_SCALE;
    float* scale = nullptr;   // [cam][cout]
    float* bias = nullptr;
    // conv2 of a block with a downsample branch (inference, f16x3): the branch rides in this convolution's contraction
    // (gemm.hip second source).  wf = [cam][cout][K + Kx] = [scale * w | ds.scale * ds.w] (FrozenBN folded), bias_f = bias + ds.bias
    int ds_index = -1;        // index of the block's downsample layer in ctx->convs, or -1
    int Kx = 0;               // = ds.cin
    float *wf = nullptr, *wf16 = nullptr, *bias_f = nullptr;
    float wf16_scale = W16_SCALE;
    // power-of-two pre-scale of this layer's INPUT activations before their fp16 split (inference, f16x3): 1 unless the
    // calibration forward of actmi_finalize found the input far from the fp16 range (FrozenBN statistics of a trained
    // checkpoint can leave a map at 1e-5 or 1e4); undone through the epilogue's alpha
    float a_scale = 1.f;
    // the split images (w16, wf16) store their K index channel-block-major, taps inner (actmi_gemm_desc.k_tap_inner): every 3x3
    // convolution that runs on the implicit-GEMM kernel with Cin % 32 == 0; the direct kernels (layer1) keep (r, s, c)
    bool k_tap_inner = false;
};

struct MhaW { float *in_w, *in_b, *out_w, *out_b; };
struct EncW { MhaW attn; float *l1w, *l1b, *l2w, *l2b, *n1w, *n1b, *n2w, *n2b; };
struct DecW { MhaW self_attn, cross; float *l1w, *l1b, *l2w, *l2b, *n1w, *n1b, *n2w, *n2b, *n3w, *n3b; };

struct DbgView { const float* ptr; int64_t numel; };

// saved activations of one encoder layer (training)
struct EncSave { float *x_in, *QKV, *lse, *ATT, *Y1, *X1, *Hb, *Y2; };
struct BlockSave { int c1, c2, ds; float *y1, *out; };

struct TrainState {
    int B = 0, fmt = 0;
    bool have_forward = false;
    const float* qpos = nullptr;
    float *gbase = nullptr, *mbase = nullptr, *vbase = nullptr;
    uint8_t* group = nullptr;
    // backbone
    float *xn4 = nullptr, *pool = nullptr, *g_act1 = nullptr, *gbuf[4] = {nullptr, nullptr, nullptr, nullptr};
    float* conv1_gw = nullptr;
    std::vector<BlockSave> blocks;
    std::vector<float*> conv_gw, conv_wd;
    std::vector<float*> conv_wd16;                 // layer1 (64 -> 64, 3x3 / s1): split image of the flipped data-gradient weights, else NULL
    // transformer
    std::vector<EncSave> en, cv;
    float *mem = nullptr, *Xc = nullptr, *cv_out = nullptr;
    int* cmap = nullptr;
    uint8_t* ckpm = nullptr;
    float *latent_info = nullptr, *z = nullptr, *eps = nullptr, *d_latent_info = nullptr, *dz = nullptr;
    float *sa_tmp = nullptr, *t1 = nullptr, *qin = nullptr, *dq = nullptr, *KV = nullptr, *lse_c = nullptr, *Oc = nullptr,
          *Y2pre = nullptr, *T2 = nullptr, *Hd = nullptr, *Y3pre = nullptr, *T3 = nullptr, *hs = nullptr, *a_hat = nullptr,
          *actions = nullptr, *losses = nullptr;
    uint8_t* is_pad = nullptr;
    // general decoder self-attention path (dropout > 0)
    float drop_p = 0.f;
    uint64_t drop_seed = 0;
    float* vq_probs = nullptr;         // VQ-ACT: softmax of the latent logits [B][vq_class*vq_dim]
    bool have_eps = true;              // false: no eps / code was supplied (VQ: draw the code on the device)
    uint8_t* pool_arg = nullptr;       // stem max-pool argmax codes (maxpool_idx_kernel)
    bool grads_dirty = false;
    hipEvent_t ev_phase1 = nullptr;    // recorded inside train_backward once the transformer.* gradients are final
    float* det_ws = nullptr;           // slices / partials of the fixed-order reductions of the backward pass
    int64_t det_ws_floats = 0;
    float* scale_slots = nullptr;      // [SCALE_SLOTS][2]: device-computed operand scales of the f16x3 backward GEMMs
    int scale_next = 0;
    const float* amax_key_ptr = nullptr; int64_t amax_key_ld = 0; int amax_key_m = 0, amax_key_n = 0;   // one-shot reuse
    const float* amax_key_slot = nullptr;
    const float* amax_pre_ptr = nullptr; float* amax_pre_slot = nullptr;   // a producer kernel left this tensor's amax bits          // the gradient arena holds gradients of an earlier backward (no zero_grad since)
    float *qkd = nullptr, *sO = nullptr, *lse_s = nullptr, *saB = nullptr, *T1B = nullptr, *dqB = nullptr, *gT1 = nullptr,
          *dsaB = nullptr, *dqkB = nullptr, *dvB = nullptr, *dqk_d = nullptr, *tmpQD = nullptr;
    // backward scratch
    float *gA = nullptr, *gB = nullptr, *gC = nullptr, *gH = nullptr, *gQKV = nullptr, *Pbuf = nullptr, *dPbuf = nullptr,
          *delta = nullptr, *dXg = nullptr, *tmp2BD = nullptr, *tmpD = nullptr, *dqb = nullptr;
    int* pos_rows = nullptr;
};

struct actmi_ctx {
    actmi_config cfg;
    int device = 0;                    // HIP device the handle was created on (all its memory lives there)
    std::string err;
    std::vector<Param> params;
    std::unordered_map<std::string, int> index;
    std::vector<void*> allocs;
    float* pbase = nullptr;
    float* p16base = nullptr;          // fp16-split image of the parameter arena (B operands of the f16x3 GEMM)
    int gemm_prec = 0;                 // ACTMI_PREC_* used by the forward GEMMs of this handle
    int train_prec = 0;                // ACTMI_PREC_BF16: the GEMMs of the TRAINING step form one bf16 product per fp32 product
                                       // (opt-in speed mode, actmi_set_train_prec / ACTMI_TRAIN_PREC=bf16); 0 = gemm_prec
    int prec_override = 0;             // set for the duration of train_forward / train_backward (PrecScope)
    int fwd_phase = 0;                 // actmi_set_forward_phase: 0 whole inference forward, 1 trunk + token assembly only, 2 transformer only
    // range guard of the f16x3 forward: one power-of-two scale per parameter for its split image, chosen at
    // finalize so that max|w| * scale lands in [2^13, 2^14) (capped at 2^12); device copies for the split kernel
    std::vector<float> pscale;         // per parameter (index = position in params)
    float* pscale_dev = nullptr;
    int* pseg64 = nullptr;             // parameter index of every 64-float slot of the arena
    int64_t *poff_dev = nullptr, *pnumel_dev = nullptr;
    unsigned* pamax_dev = nullptr;
    float conv1_wscale = W16_SCALE;
    float bwd_wscale = W16_SCALE;      // static scale of weights used as on-the-fly B operands of the backward GEMMs
    uint32_t* flags = nullptr;         // device flag word (ACTMI_FLAG_*)
    float* splitk_ws = nullptr;        // slices of the forward GEMMs whose contraction is split to fill the chip
    int64_t splitk_ws_floats = 0;
    int conv1_vpool = 1;               // inference: conv1 emits the vertical half of the max pool (ACTMI_CONV1_VPOOL=0: off)
    int fwd_splitk = 1;                // 0: never split a forward contraction (ACTMI_FWD_SPLITK=0)
    int sk_target = 1536, sk_minnk = 12, sk_maxtiles = 768;     // split heuristic (tuning aids ACTMI_FWD_SPLITK_*)
    int sk_target_long = 4864, sk_maxtiles_long = 1300, sk_long_nk = 128;   // B = 8: very long contractions (layer4, K = 4608)
    hipStream_t side_stream = nullptr; // downsample branch of the ResNet blocks (engine_backbone)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool ds_fork = true;
    // two camera halves of the ResNet trunk as two parallel branches (second stream): the tail of one half's launch is
    // filled by the other half's next launch (default on, ACTMI_CAM_PIPE=0 disables; engine_backbone)
    hipStream_t pipe_stream = nullptr;             // branch 1 (non-null = branches available)
    hipStream_t pipe_streams[3] = {nullptr, nullptr, nullptr};     // branches 1 .. nbranch-1 (pipe_streams[0] == pipe_stream)
    hipEvent_t ev_pfork = nullptr, ev_pjoin = nullptr;
    hipEvent_t ev_pjoins[3] = {nullptr, nullptr, nullptr};
    int nbranch = 2;                   // ACTMI_BRANCHES (2 .. 4)
    bool cam_pipe = false;
    int ln_split_short = 1;            // the same for 16 <= K/32 < 64 (out-proj: K = 512) (ACTMI_LN_SPLIT_SHORT)
    int ln_split = 3;                  // split factor of a long-K product followed by a slice-summing LayerNorm (ACTMI_LN_SPLIT)
    int last_B = 0;                    // batch of the forward in flight (debug views)
    int policy_mult = 1;               // split-K policy counts the tiles of the WHOLE camera set while a half is being launched
    bool act_calib = true;             // activation pre-scales measured at finalize (ACTMI_ACT_CALIB=0: off)
    bool calibrating = false;          // engine_backbone is running the calibration forward
    float ip_a_scale = 1.f;            // the same for input_proj's operand (the layer4 maps)
    float* act_scale_dev = nullptr;    // device copies [convs.size() + 1] for the kernels that take a device scale (conv3.hip)
    bool fuse_ds = true;               // downsample branch inside conv2's contraction (ACTMI_FUSE_DS=0: three launches as before)
    int64_t ptotal = 0;
    bool finalized = false;
    // geometry
    int H1, W1, H2, W2, fh, fw, P_, N;
    // prepared weights
    std::vector<ConvLayer> convs;
    float *conv1_w = nullptr, *conv1_scale = nullptr, *conv1_bias = nullptr, *lut = nullptr;
    float* conv1_wimg = nullptr;       // f16x3: conv1's LDS weight image per camera (launch_conv1_wimg), rebuilt with the weights
    float *pos_tokens = nullptr, *dec_t1 = nullptr, *dec_q = nullptr, *tmp_vec = nullptr;
    int* rowmap = nullptr;
    int rowmap_B = -1;
    std::vector<EncW> enc, cvae;
    std::vector<DecW> dec;
    // activations
    float *act1 = nullptr, *buf[3] = {nullptr, nullptr, nullptr};
    float *X = nullptr, *X1 = nullptr, *Y = nullptr, *ATT = nullptr, *QKV = nullptr, *Hb = nullptr;
    float* XP = nullptr;               // x + pos of the encoder stream, written by the LayerNorm that produces x (ACTMI_LN_XP=0: off)
    bool ln_xp = true, ln_head = true; // LayerNorm extras: x + pos second output, action head in the decoder's last LayerNorm
    float *dO = nullptr, *dY = nullptr, *dT2 = nullptr, *dH = nullptr, *hs = nullptr;
    float* attn_ws = nullptr;          // split-KV partials (attn.hip)
    int64_t attn_ws_floats = 0;
    std::map<std::string, DbgView> dbg;
    std::string stop_stage;   // debug: return from the forward right after this stage
    TrainState* train = nullptr;

    float* P(const std::string& key);
};

int engine_create(const actmi_config* cfg, actmi_ctx** out);
// forward GEMMs of a handle go through here: applies the handle's precision and swaps in pre-split weights
// LayerNorm that follows a product (y = LN(C), optionally a second LN on top): when the product's contraction is split, the
// LayerNorm kernel sums the slices itself (no combine pass); done tells the caller whether that happened
struct LnFuse {
    const float *w, *b, *w2, *b2;
    float* out;
    float eps;
    bool done;
    const LnExtra* extra = nullptr;    // extra outputs of that LayerNorm (x + pos for the next attention block, the action head)
};
int ctx_gemm(actmi_ctx* ctx, GemmArgs a, hipStream_t st, int ws_half = -1, LnFuse* ln = nullptr);
// precision of the GEMMs issued while a training call is running (restored on every exit path)
struct PrecScope {
    actmi_ctx* c;
    explicit PrecScope(actmi_ctx* ctx) : c(ctx) { c->prec_override = c->train_prec; }
    ~PrecScope() { c->prec_override = 0; }
};
int engine_destroy(actmi_ctx* ctx);
const char* engine_create_error();
int engine_finalize(actmi_ctx* ctx, hipStream_t st);
int engine_prepare_weights(actmi_ctx* ctx, hipStream_t st, bool after_step = false);
int engine_calibrate_weight_scales(actmi_ctx* ctx, hipStream_t st);
int engine_calibrate_activations(actmi_ctx* ctx, hipStream_t st);
int engine_measure_act_scale(actmi_ctx* ctx, const float* x, int64_t rows, int cols, hipStream_t st, float* out);
int engine_backbone(actmi_ctx* ctx, const void* image, int fmt, int B, hipStream_t st);
float engine_weight_scale(const actmi_ctx* ctx, const float* w);
int train_create(actmi_ctx* ctx);
int train_forward(actmi_ctx* ctx, const float* qpos, const void* image, int fmt, const float* actions, const uint8_t* is_pad,
                  const float* eps, uint64_t dropout_seed, float dropout_p, int B, float* losses, float* a_hat_out,
                  float* mu_out, float* logvar_out, hipStream_t st);
int train_backward(actmi_ctx* ctx, float loss_scale, hipStream_t st);
int train_zero_grad(actmi_ctx* ctx, hipStream_t st);
int train_adamw_step(actmi_ctx* ctx, float lr, float lr_backbone, float wd, float b1, float b2, float eps, int64_t step,
                     hipStream_t st);
int train_adamw_range(actmi_ctx* ctx, float lr, float lr_backbone, float wd, float b1, float b2, float eps, int64_t step,
                      int64_t offset, int64_t count, hipStream_t st);
int engine_forward_infer(actmi_ctx* ctx, const float* qpos, const void* image, int fmt, int B, float* a_hat,
                         hipStream_t st, const float* vq_sample /* [B][vq_class*vq_dim] or null */);
