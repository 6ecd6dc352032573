/*
 * libactmi -- C ABI of the MI355X-native ACT policy path (gfx950).
 *
 * This header is the drop-in boundary below the reference's Python policy adaptor.  The reference owns no
 * native code: its ACT arithmetic is reached through `ACTPolicy.__call__(qpos, image, actions, is_pad)`
 * (reference policy.py:264-332) which calls `DETRVAE.forward` (detr/models/detr_vae.py:163-254); the
 * optimizer is `torch.optim.AdamW` built at detr/main.py:102-110 and the temporal ensemble is inline
 * in `eval_bc` (imitate_episodes.py:402-411).  Each entry point below names the reference interface it
 * replaces.  Plain pointers and sizes only; no torch types.  All pointers are DEVICE pointers valid on
 * the handle's device unless a parameter is documented as host memory; `stream` is a hipStream_t passed
 * as void*.  Every function returns 0 on success or a negative ACTMI_E_* code and never throws;
 * `actmi_last_error` returns the message.  A handle is bound to one device and is not re-entrant.
 */
#ifndef ACTMI_H
#define ACTMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACTMI_VERSION 110

#define ACTMI_OK 0
#define ACTMI_E_INVALID (-1)   /* bad argument / unknown key / shape mismatch */
#define ACTMI_E_SHAPE (-2)     /* operand shape or alignment not supported by a kernel */
#define ACTMI_E_LAUNCH (-3)    /* HIP launch or runtime error */
#define ACTMI_E_STATE (-4)     /* call order (e.g. forward before finalize, backward before forward_train) */
#define ACTMI_E_NOMEM (-5)

#define ACTMI_IMG_U8_NHWC 0    /* [B][C][H][W][3] uint8, the on-disk / simulator format (utils.py:104) */
#define ACTMI_IMG_F32_NCHW 1   /* [B][C][3][H][W] float32 in [0,1], the ACTPolicy.__call__ contract */

typedef struct actmi_ctx* actmi_handle;

/* Model hyper-parameters: reference imitate_episodes.py:78-94 + detr/main.py:12-89 defaults. */
typedef struct actmi_config {
    /* ABI guard: set to sizeof(actmi_config) as the CALLER's binding declares it.  actmi_create rejects any other value
     * with ACTMI_E_INVALID ("actmi_config.struct_size ..."), so a binding written against an older header (fewer trailing
     * fields) fails loudly instead of having the library read past the end of its struct. */
    uint32_t struct_size;
    int32_t num_cams;
    int32_t image_h, image_w;
    int32_t base_width;        /* resnet18 stem width (64) */
    int32_t hidden_dim, nheads, dim_feedforward;
    int32_t enc_layers, dec_layers;
    int32_t num_queries, state_dim, action_dim, latent_dim;
    int32_t has_cvae_encoder;  /* 0 when --no_encoder */
    int32_t max_batch;         /* workspace is sized for this batch at create time */
    int32_t enable_training;   /* allocate gradient / optimizer / saved-activation storage */
    float kl_weight;
    /* VQ-ACT (detr_vae.py:50-60): latent_proj emits vq_class*vq_dim logits and latent_out_proj takes the
     * [vq_class x vq_dim] one-hot code (actmi_forward_infer_vq; in training `eps` carries the sampled code or is NULL to
     * draw it on the device, `mu` / `logvar` return probs / code, kl = 0 and the straight-through estimator of
     * detr_vae.py:137-145 routes the code's gradient to the softmax). */
    int32_t vq, vq_class, vq_dim;
} actmi_config;

/* GEMM / implicit-GEMM convolution descriptor (also the unit-test entry of the MFMA kernel):
 * C[rowmap(m)][n] = act((sum_k A'[m][k] * Bw[n][k]) * scale[n] + bias[n] + res[m % res_mod][n]).
 * Replaces ATen addmm / cuDNN convolution as launched by nn.Linear, nn.MultiheadAttention projections and
 * torchvision Conv2d+FrozenBatchNorm2d (backbone.py:47-57). */
#define ACTMI_PREC_F32 1
#define ACTMI_PREC_F16X3 2
#define ACTMI_PREC_BF16 3      /* one bf16 MFMA product per fp32 product, fp32 accumulate: ~1e-2 relative; training speed mode only */

typedef struct actmi_gemm_desc {
    const float* A;
    int64_t lda;
    /* A operand form.  0: row-major [M][K] (contraction contiguous).  1: NHWC implicit im2col (forward conv),
     * K index = (r*KW+s)*Cin+c.  2: conv data-gradient gather: rows are INPUT pixels (b,hi,wi), A = dY NHWC
     * [img][Ho][Wo][Cout], K index = (r*KW+s)*Cout+n.  (ta=1 selects the transposed form instead.) */
    int32_t mode;
    int32_t H, W, Cin, KH, KW, stride, pad, Ho, Wo;
    int64_t img_stride;
    const float* A_add;        /* optional: A'[m][k] = A[m][k] + A_add[m % add_mod][k] for columns n < add_ncols */
    int64_t ld_add;
    int32_t add_mod;
    int32_t add_ncols;
    const float* Bw;           /* [N][K] row-major (torch Linear layout) unless tb */
    int64_t ldb;
    const float* scale;        /* per-n or NULL */
    const float* bias;         /* per-n or NULL */
    const float* res;          /* residual or NULL */
    int64_t ldres;
    int32_t res_mod;
    int32_t relu;              /* activation: 0 none, 1 ReLU, 2 GELU (exact erf form, nn.GELU default) */
    float* C;
    int64_t ldc;
    const int32_t* rowmap;
    int32_t M, N, K;
    int32_t groups;            /* blockIdx.z; per-group element offsets below */
    int64_t gA, gB, gSB, gC, gRes;
    /* ---- extensions used by the backward pass ---- */
    int32_t ta;                /* 1: A stored [K][M] (out index contiguous), lda = row stride of that storage */
    int32_t tb;                /* 1: B stored [K][N]; 2: conv weight-gradient gather: B[(r,s,c)][m=(b,ho,wo)] read from
                                  the NHWC input X (H,W,Cin,...,Ho,Wo describe the forward conv), out index = (r*KW+s)*Cin+c */
    const int32_t* a_rowmap;   /* optional gather of A rows (mode 0, ta 0) */
    const float* B_add;        /* tb=1 only: B'[k][n] = B[k][n] + B_add[k % badd_mod][n] */
    int64_t ld_badd;
    int32_t badd_mod;
    int32_t splitk;            /* >1: contraction split over blockIdx.z, results atomically added into C (C pre-zeroed) */
    const float* mask;         /* epilogue: v = 0 where mask[m][n] <= 0 (ReLU backward), same ld as C unless ldmask */
    int64_t ldmask;
    float* C2;                 /* optional second output C2[m][n] = v * scale2[n] (ld = ldc) */
    const float* scale2;
    float alpha;               /* acc multiplier; 0 means 1 */
    int32_t groups_inner;      /* >0: group g = (g / groups_inner, g % groups_inner) with the second-level strides below */
    int64_t gA2, gB2, gC2, gRes2;
    int64_t gMask, gC2out;     /* per-group strides of mask / C2 (first level) */
    /* dropout on the (scaled, biased, ReLU'd when no residual) result before the residual is added; the mask is a pure
     * function of (drop_seed, output element index), kept values are scaled by 1/(1-drop_p); drop_p = 0 disables */
    float drop_p;
    uint64_t drop_seed;
    /* diagnostic: when non-NULL, thread 0 of every block writes 4 shader-clock stamps (s_memtime) here:
     * [block][0] entry, [1] first LDS stage ready, [2] K loop done, [3] epilogue done.  Never set on the product path. */
    uint64_t* stamps;
    /* how each fp32 product is formed: 0 = default (environment ACTMI_GEMM_PREC=f32|f16x3, else the library default),
     * ACTMI_PREC_F32 = native fp32 MFMA, ACTMI_PREC_F16X3 = exact two-piece fp16 split of both operands, three fp16
     * MFMA products, fp32 accumulation (fp32-grade results, needs finite |x| < 65504) */
    int32_t prec;
    /* f16x3 only: Bw already holds split weights (actmi_op_split16: every aligned group of 4 floats replaced by
     * 4 hi halfs + 4 lo halfs, same addressing as the fp32 matrix) */
    int32_t b_split;
    /* f16x3 only: power-of-two operand pre-scales, undone through alpha (0 means 1).  With b_split, b_scale says what the
     * image was built with (actmi_op_split16); otherwise the kernel multiplies the operand on its way into LDS.  Use them
     * for operands whose typical magnitude is below ~1e-2 (weights: 256) so that the lo pieces stay normal fp16 numbers;
     * |x| * scale must stay below 65504. */
    float b_scale;
    float a_scale;
    /* f16x3 only: optional device-resident power-of-two scales (one float each, see actmi_op_pow2_scale), multiplied
     * with a_scale / b_scale; read by the kernel, so they can be produced on the same stream just before the launch */
    const float* a_scale_dev;
    const float* b_scale_dev;
    /* splitk > 1 with split_stride != 0: split s stores its partial product plainly at C + s*split_stride (elements; no
     * atomics, C need not be zeroed, the fast epilogue applies); actmi_op_splitk_combine sums the slices in a fixed order
     * and applies scale / bias / residual / activation.  Every split must own at least one K tile of 32. */
    int64_t split_stride;
    /* optional: the launch takes the integer maximum of the bits of |v| over every value it stores into *amax_out
     * (non-negative floats order like unsigned integers; the word must be zero or hold an earlier maximum): what
     * actmi_op_pow2_scale computes with a pass of its own, for the GEMM that reads this output next.  Meaningless with an
     * atomic split-K (partial sums are stored). */
    uint32_t* amax_out;
    /* fused attention-backward epilogues (training path; fast epilogue forms only, no row map / C2 / dropout):
     *   epi = 1: C = exp(v - epi_row[m]), zero where epi_colkill[n] != 0   (v = alpha * acc: softmax probabilities from the
     *            scores and the saved log-sum-exp; epi_colkill = key padding mask, optional)
     *   epi = 2: C = res[m][n] * (v - epi_row[m]) * epi_scale              (dS = P * (dP - delta) * scale; res is a factor here,
     *            not an addend)
     * epi_row: per-row vector of the group, group strides gRow (first level) / gRow2 (second level, with groups_inner);
     * epi_colkill: bytes per column, first-level group stride gColkill (second level shares it). */
    int32_t epi;
    float epi_scale;
    const float* epi_row;
    int64_t gRow, gRow2;
    const uint8_t* epi_colkill;
    int64_t gColkill;
    /* tile shape: 0 = chosen per launch shape, 1 = 128x128, 2 = 128x64, 3 = 64x64 (tuning aid for launches the shape model
     * misjudges: the K = 64 products of the attention backward are all epilogue) */
    int32_t tile_hint;
    /* optional: OR finite_bit into *finite_flag when any value this launch stores is NaN or infinite (the handle's output
     * guard rides on the last product of the forward instead of a pass of its own) */
    uint32_t* finite_flag;
    uint32_t finite_bit;
    /* mode 1 only, optional SECOND SOURCE of the contraction: for k >= kx_begin (= KH*KW*Cin)
     * A'[m = (img, ho, wo)][k] = Ax[img][ho * stride_x][wo * stride_x][k - kx_begin], Ax an NHWC map [img][Hx][Wx][Cx], K = kx_begin
     * + Cx; Bw rows carry the matching Cx extra columns.  A ResNet block's 1x1 / stride-2 downsample convolution (+ FrozenBN,
     * folded into the weights) rides in its second 3x3 convolution this way (torchvision BasicBlock, backbone.py:66-71).
     * Needs prec f16x3, b_split, Cin % 32 == 0, Cx % 32 == 0, no operand pre-scales / fused epilogues; gAx = group stride. */
    const float* Ax;
    int32_t kx_begin, Hx, Wx, Cx, stride_x;
    int64_t gAx;
    /* mode 0, ta 0, optional: the output columns n < alt_ncols (a multiple of 128) contract the rows of A_alt (same shape / lda /
     * group stride as A) instead of A -- nn.MultiheadAttention's packed projection with q = k = x + pos, v = x
     * (transformer.py:216-217) when x + pos already exists as a matrix */
    const float* A_alt;
    int32_t alt_ncols;
    /* mode 1, Cin % 32 == 0: the K index of the Bw rows is ((c / 32) * KH*KW + r*KW + s) * 32 + c % 32 (channel blocks outer,
     * taps inner) instead of (r*KW + s) * Cin + c -- the order of the engine's split images of the 3x3 convolutions, chosen for
     * L2 reuse of the input patch (actmi_op_permute_conv_k builds it from the reference order).  With a second source (Ax) the
     * extra Cx columns still follow the KH*KW*Cin permuted ones. */
    int32_t k_tap_inner;
} actmi_gemm_desc;

/* Multi-head attention descriptor: softmax(scale * q k^T [+ key padding mask]) v per (batch, head).
 * Replaces nn.MultiheadAttention's bmm/softmax/bmm (transformer.py:217-218, 282-289). */
typedef struct actmi_attn_desc {
    const float* Q; int64_t q_bs, q_rs;
    const float* K; int64_t k_bs, k_rs;
    const float* V; int64_t v_bs, v_rs;
    float* O; int64_t o_bs, o_rs;
    const uint8_t* kpm; int64_t kpm_bs;
    float* lse;
    int32_t B, H, Nq, Nk, HD;
    float scale;
    /* optional split-KV workspace: when the (query block x head x batch) grid is too small to fill 256 CUs the key range
     * is split over `nsplit` blocks whose partial (max, sum, O) are merged by a second kernel.  ws_floats >=
     * nsplit * B * Nq * (H*HD + 2*H); ws = NULL disables splitting. */
    float* ws;
    int64_t ws_floats;
    /* dropout on the attention weights (nn.MultiheadAttention dropout): mask = f(drop_seed, ((b*H+h)*Nq+q)*Nk+key) */
    float drop_p;
    uint64_t drop_seed;
    /* product precision, as in actmi_gemm_desc.prec (0 = ACTMI_GEMM_PREC from the environment, else native fp32) */
    int32_t prec;
    /* 1: causal mask, key j visible to query i only for j <= i (needs Nq == Nk) */
    int32_t causal;
} actmi_attn_desc;

int actmi_version(void);

/* ---- lifecycle ------------------------------------------------------------------------------------ */
/* replaces build_ACT_model_and_optimizer (detr/main.py:92-112) without touching sys.argv */
int actmi_create(const actmi_config* cfg, actmi_handle* out);
int actmi_destroy(actmi_handle h);
const char* actmi_last_error(actmi_handle h);   /* h may be NULL: last error of a failed create */

/* state_dict wire format (policy.py:344-348 serialize/deserialize): keys WITHOUT the "model." prefix,
 * float32, shapes as in the reference state_dict.  src/dst are host pointers unless is_device != 0. */
int actmi_num_params(actmi_handle h);
int actmi_param_info(actmi_handle h, int index, const char** key, int64_t* shape4, int* ndim, int* is_buffer);
int actmi_set_param(actmi_handle h, const char* key, const void* src, const int64_t* shape, int ndim, int is_device);
int actmi_get_param(actmi_handle h, const char* key, void* dst, int64_t nbytes, int is_device);
/* device address of a parameter's fp32 master copy inside the handle's arena (nn.Module.parameters() views, policy.py:243;
 * READ-ONLY for the caller: derived weights are rebuilt only by actmi_set_param + actmi_finalize / actmi_adamw_step) */
int actmi_param_ptr(actmi_handle h, const char* key, void** dev_ptr, int64_t* numel);
/* fold FrozenBN, repack conv weights, build position tables and the constant decoder query path.
 * Must be called after the last set_param and before the first forward. */
int actmi_finalize(actmi_handle h, void* stream);

/* ---- inference: ACTPolicy.__call__(qpos, image) -> a_hat (policy.py:322-332) ----------------------- */
int actmi_forward_infer(actmi_handle h, const float* qpos /*[B][S]*/, const void* image, int image_fmt, int B,
                        float* a_hat /*[B][Q][A]*/, void* stream);

/* The inference forward in two halves, for callers that feed frames from the host: phase 1 = multi-camera ResNet trunk + token
 * assembly (the only reader of `image` and `qpos`, and the HBM-heavy part of the step), phase 2 = transformer + action head (reads
 * the tokens phase 1 left in the handle; `image`, `qpos` are ignored).  actmi_forward_infer runs the selected phase until it is set
 * back to 0 (whole step, the default).  Captured into two hipGraphs, an ordinary stream event between them releases the
 * host-to-device copy of the NEXT frame beside the transformer instead of beside the stem (actmi/engine.py:InferPipeline).
 * The reference has no counterpart (its eval loop copies, then computes: imitate_episodes.py:286-300). */
int actmi_set_forward_phase(actmi_handle h, int phase);

/* VQ-ACT inference, ACTPolicy.__call__(qpos, image, vq_sample=code) (policy.py:322-332, detr_vae.py:155-156): the latent
 * token is latent_out_proj(vq_sample[b]) instead of latent_out_proj(0).  vq_sample: [B][vq_class*vq_dim] f32 device. */
int actmi_forward_infer_vq(actmi_handle h, const float* qpos, const void* image, int image_fmt, int B,
                           const float* vq_sample, float* a_hat, void* stream);

/* ---- training: ACTPolicy.__call__(qpos, image, actions, is_pad) -> {l1, kl, loss} (policy.py:288-320) -- */
int actmi_forward_train(actmi_handle h, const float* qpos, const void* image, int image_fmt,
                        const float* actions /*[B][Q][A]*/, const uint8_t* is_pad /*[B][Q]*/,
                        const float* eps /*[B][L]*/, uint64_t dropout_seed, float dropout_p, int B,
                        float* losses /*[3] l1, kl, loss*/, float* a_hat, float* mu, float* logvar, void* stream);
/* loss.backward() (imitate_episodes.py:606) */
int actmi_backward(actmi_handle h, float loss_scale, void* stream);
/* optimizer.zero_grad() / optimizer.step() with the two AdamW groups of detr/main.py:102-110.
 * actmi_adamw_step is gated ON THE DEVICE by the handle's flag word: while ACTMI_FLAG_LOSS or ACTMI_FLAG_WEIGHT is up (a
 * non-finite training loss, a weight beyond its split scale) the update is skipped -- parameters and Adam moments stay
 * untouched until the host has read and cleared the word with actmi_get_flags (and re-finalized for _WEIGHT). */
int actmi_zero_grad(actmi_handle h, void* stream);
int actmi_adamw_step(actmi_handle h, float lr, float lr_backbone, float weight_decay, float beta1, float beta2,
                     float eps, int64_t step, void* stream);
/* Sharded optimizer of data-parallel training (reduce-scatter -> sharded AdamW -> all-gather): the same update
 * on the arena range [offset, offset + count) only (offset a multiple of 64 floats), WITHOUT rebuilding the derived weights;
 * actmi_refresh_weights rebuilds them (conv repack, split images, decoder constants) once the caller has all-gathered the
 * updated parameter arena (actmi_param_arena).  actmi_adamw_step == actmi_adamw_step_range(0, whole arena) + refresh. */
int actmi_adamw_step_range(actmi_handle h, float lr, float lr_backbone, float weight_decay, float beta1, float beta2,
                           float eps, int64_t step, int64_t offset, int64_t count, void* stream);
int actmi_refresh_weights(actmi_handle h, void* stream);
/* the whole fp32 parameter arena (the layout of the gradient arena: actmi_grad_arena) */
int actmi_param_arena(actmi_handle h, void** dev_ptr, int64_t* nfloats);
int actmi_grad_ptr(actmi_handle h, const char* key, void** dev_ptr, int64_t* numel);
/* the whole gradient arena (same layout as the parameter arena) for bucketed data-parallel all-reduce over RCCL */
int actmi_grad_arena(actmi_handle h, void** dev_ptr, int64_t* nfloats);

/* Data-parallel overlap: actmi_backward records an event once the gradients of phase 1 -- every transformer.*
 * parameter, the contiguous head [offset, offset + count) of the gradient arena reported by actmi_grad_phase_range -- are
 * final, before the backbone / CVAE-encoder backward is enqueued.  actmi_wait_grad_phase makes `stream` (the stream the
 * caller issues its RCCL collective from) wait for that event, so the reduction of that range runs under the rest of the
 * backward; phase 2 is the remainder of the arena, final when the stream actmi_backward ran on has drained. */
int actmi_grad_phase_range(actmi_handle h, int phase, int64_t* offset, int64_t* count);
int actmi_wait_grad_phase(actmi_handle h, int phase, void* stream);

/* ---- temporal ensembling over E episodes (imitate_episodes.py:338-339, 402-411) ------------------- */
/* ring [E][Q][Q][A] f32 zero-initialised, tcount [E] i32 zero-initialised, chunk [E][Q][A] f32;
 * out [E][A] f64 (the reference's raw_action is float64), populated [E][Q] u8 or NULL (oldest row first). */
int actmi_ensemble_step(float* ring, int32_t* tcount, const float* chunk, double k, double* out, uint8_t* populated,
                        int E, int Q, int A, void* stream);

/* ---- kernel-level entry points (unit tests, external callers) --------------------------------------- */
int actmi_op_gemm(const actmi_gemm_desc* d, void* stream);
/* weight preparation for actmi_gemm_desc.b_split: dst = fp16-split image of src * scale (device pointers, may not
 * alias).  scale must be a power of two with |src| * scale < 65504; 256 suits network weights: pieces of values
 * around 1e-2 then stay normal fp16 numbers (full 22-bit split) instead of subnormals. */
int actmi_op_split16(const float* src, float* dst, int64_t nfloats, float scale, void* stream);
/* rows [rows][ld] of a convolution weight matrix: the first taps*cin columns re-ordered from (tap, c) to (c / 32, tap, c % 32)
 * (actmi_gemm_desc.k_tap_inner), any further columns (ld > taps*cin: a second source's) copied as they are; cin % 32 == 0 */
int actmi_op_permute_conv_k(const float* src, float* dst, int64_t rows, int taps, int cin, int ld, void* stream);
/* out[0] = the power of two s with max|x| * s in [2^13, 2^14) over the M x N matrix x (row stride ld); 1 if x is all
 * zero or not finite.  out[1] is scratch and must be zero before the first use (the op leaves it zero).  For actmi_gemm_desc.a_scale_dev / b_scale_dev. */
int actmi_op_pow2_scale(const float* x, int64_t ld, int M, int N, float* out, void* stream);
/* second half of a sliced split-K product (actmi_gemm_desc.split_stride): out[m][n] = act((sum_s part[s*split_stride +
 * m*ldp + n]) * scale[n] + bias[n] + res[m][n]); slices are summed in index order, so results are run-to-run identical.
 * relu as in actmi_gemm_desc; scale / bias / res may be NULL. */
int actmi_op_splitk_combine(const float* part, int nsplit, int64_t split_stride, int64_t ldp, int M, int N, const float* scale,
                            const float* bias, const float* res, int64_t ldres, int relu, float* out, int64_t ldc, void* stream);
/* one categorical draw per row: code[i] = one_hot(sample(softmax(logits[i] / temperature))) by inverse CDF on the
 * counter-based generator (replaces torch.multinomial in detr_vae.py:140 and latent_model.py:68-69); probs may be NULL */
int actmi_op_sample_onehot(const float* logits, int n, int V, float temperature, uint64_t seed, float* probs, float* code,
                           void* stream);
int actmi_op_attention(const actmi_attn_desc* d, void* stream);
/* Attention backward without materialised scores (training step, long sequences; csrc/attn_bwd.hip): dq / dk / dv of
 * out = softmax(scale * q k^T [+ key padding mask]) v per (batch, head), from the forward's operands, its output `o`, its
 * log-sum-exp `lse` [B][H][Nq] (actmi_attn_desc.lse) and the output gradient `d_o`.  q / k / v / dq / dk / dv are addressed by
 * (batch stride, row stride) as in actmi_attn_desc, heads contiguous blocks of HD floats in a row; o and d_o are [B][Nq][H*HD].
 * delta_ws: B*H*Nq floats of scratch.  do_scale (optional, device): the power of two d_o is multiplied with on its way into
 * its fp16 pieces (actmi_op_pow2_scale of d_o); amax_out (optional): bits of the largest |gradient| written.
 * Replaces autograd through nn.MultiheadAttention's bmm / softmax / dropout / bmm (transformer.py:217-218). */
typedef struct actmi_attn_bwd_desc {
    const float *q, *k, *v, *o, *d_o, *lse;
    float *dq, *dk, *dv;
    int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs;
    const uint8_t* kpm;        /* optional key padding mask [B][Nk], nonzero = masked */
    int64_t kpm_bs;
    int32_t B, H, Nq, Nk, HD;  /* HD in {16, 32, 64} */
    float drop_p;
    uint64_t drop_seed;
    float* delta_ws;
    const float* do_scale;
    uint32_t* amax_out;
} actmi_attn_bwd_desc;
int actmi_op_attention_bwd(const actmi_attn_bwd_desc* d, void* stream);
int actmi_op_layernorm(const float* x, const float* res, int res_mod, const float* w, const float* b, const float* w2,
                       const float* b2, float* y, int M, int D, float eps, void* stream);
int actmi_op_maxpool3x3s2(const float* in_nhwc, float* out_nhwc, int nimg, int H, int W, int C, void* stream);
/* conv1: w is the torch OIHW [C][Cout][3][7][7] weight, scale/bias the folded FrozenBN; out camera-major NHWC */
int actmi_op_conv1(const void* image, int image_fmt, const float* w_oihw, const float* scale, const float* bias,
                   float* out, float* workspace /* >= C*Cout*148 + 768 floats */, int B, int C, int H, int W, int Cout,
                   int prec /* ACTMI_PREC_*, 0 = environment / native fp32 */, void* stream);
/* The same stem for callers that keep its prepared weights (and may want it without the ACT path's ImageNet normalisation or
 * ReLU: the DiffusionPolicy trunk applies GroupNorm between the convolution and the ReLU and feeds x / 255 un-normalised,
 * policy.py:150-170): actmi_op_conv1_prepare fills `workspace` (>= actmi_op_conv1_workspace_floats(C, Cout) floats) once --
 * repacked weights, the f16x3 LDS weight image, the u8 lookup table (lut_mode 0: ((v / 255) - mean) / std as the ACT path,
 * 1: v / 255), unit scale / zero bias -- and synchronises; actmi_op_conv1_prepared then only launches (graph-capturable): out =
 * act(conv * scale + bias), scale / bias NULL = 1 / 0, relu 0 = no activation (f16x3 only).  u8 NHWC images. */
int64_t actmi_op_conv1_workspace_floats(int C, int Cout);
int actmi_op_conv1_prepare(const float* w_oihw, float* workspace, int C, int Cout, int lut_mode, void* stream);
int actmi_op_conv1_prepared(const void* image_u8, const float* workspace, const float* scale, const float* bias, float* out,
                            int B, int C, int H, int W, int Cout, int relu, void* stream);
/* direct 3x3 / stride 1 / pad 1 convolution for 64 -> 64 channels (ResNet18 layer1), f16x3: x camera-major NHWC
 * [G][B][H][W][64]; w16 = actmi_op_split16 image (built with w_scale) of the weights [G][64][3][3][64] (cout, r, s, cin);
 * out = act(conv * scale + bias (+ res)) */
int actmi_op_conv3x3_c64(const float* x, const float* w16, float w_scale, const float* scale, const float* bias,
                         const float* res, float* out, int G, int B, int H, int W, int relu, void* stream);
/* weight gradient of the same 64 -> 64 channel 3x3 / s1 / p1 convolution (training path; torch.autograd of F.conv2d in
 * torchvision's BasicBlock, reference backbone.py:95-134), f16x3 arithmetic: dw[G][64][(r,s,ci)] = sum over images and pixels of
 * dy[G][B][H][W][64] x x[G][B][H][W][64] shifted by the tap.  ws: workspace of >= G * 64 * 576 * min(256 / G, B * ceil(W/32))
 * floats (per-workgroup partials, summed in a fixed order: bitwise repeatable); dy_scale_dev: optional device power-of-two
 * scale applied to dy before the fp16 split and undone in the result (actmi_op_pow2_scale). */
int actmi_op_wgrad3x3_c64(const float* dy, const float* x, float* dw, float* ws, int64_t ws_floats, const float* dy_scale_dev, int G,
                          int B, int H, int W, void* stream);

/* weight gradient of the ResNet stem (7x7 / s2 / p3 convolution of the 4-channel-padded image to 64 channels; training path):
 * dw[G][64][(r,s,c) = 196] from dy[G][B][Ho][Wo][64] and x4[G][B][H][W][4], Ho = (H-1)/2 + 1, Wo = (W-1)/2 + 1; ws: >= G * 64 *
 * 196 * min(512 / G, B * ceil(Wo/32)) floats of per-workgroup partials (fixed-order sum); dy_scale_dev as above */
int actmi_op_wgrad7x7s2(const float* dy, const float* x4, float* dw, float* ws, int64_t ws_floats, const float* dy_scale_dev, int G,
                        int B, int H, int W, void* stream);

/* ---- DiffusionPolicy inference path (reference policy.py:20-241, imitate_episodes.py:100-118,420-426).
 * The non-GEMM pieces of what the reference delegates to robomimic (ResNet18Conv with BatchNorm -> GroupNorm, SpatialSoftmax,
 * ConditionalUnet1D) and diffusers (DDIMScheduler.step); restated from the published definitions, parity unpinned (neither
 * package is importable offline).  Channel-last tensors; dense contractions go through actmi_op_gemm.
 * groupnorm: out = act(GN_G(x) [+ res when res_mode 1]) [* film_scale[n][c] + film_bias[n][c]] [+ res when res_mode 2];
 *            x [n][P][C], statistics per (sample, group) over P x C/G values (torch.nn.GroupNorm); act 0 none, 1 ReLU, 2 Mish
 * spatial_softmax: logits [n][H*W][K] -> out [n][K][2] = softmax-weighted (x, y) on the [-1, 1] grid
 * unfold1d: x [B][T][C] -> out [B][To][k][C]; transposed = 0: rows of Conv1d(k, stride, pad); 1: gather rows of
 *           ConvTranspose1d(k, stride, pad) (out[b][t][j] = x[b][(t + pad - j) / stride] when divisible and in range)
 * ddim_step: in place x <- sqrt_aprev * clamp((x - sqrt_1m_at * eps) * inv_sqrt_at, -1, 1) + sqrt_1m_aprev * eps
 * u8_to_nhwc4: u8 [B][Cam][H][W][3] -> f32 [Cam][B][H][W][4] = v / 255 (fourth channel 0) */
int actmi_op_groupnorm(const float* x, const float* res, const float* film_scale, const float* film_bias, const float* w,
                       const float* b, float* out, int n, int P, int C, int G, float eps, int act, int res_mode, float* ws,
                       int64_t ws_floats, void* stream);
int actmi_op_spatial_softmax(const float* logits, float* out, int n, int H, int W, int K, float temperature, void* stream);
int actmi_op_unfold1d(const float* x, float* out, int B, int T, int C, int k, int stride, int pad, int To, int transposed,
                      void* stream);
int actmi_op_ddim_step(float* x, const float* eps, int64_t n, float inv_sqrt_at, float sqrt_1m_at, float sqrt_aprev,
                       float sqrt_1m_aprev, int clip, void* stream);
int actmi_op_mish(const float* x, float* y, int64_t n, void* stream);
int actmi_op_u8_to_nhwc4(const uint8_t* image, float* out, int B, int Cam, int H, int W, void* stream);
/* ---- training step of the VQ-ACT latent prior (reference detr/models/latent_model.py:8-56 as driven by
 * train_latent_model.py:323-343: forward_pass, F.cross_entropy, torch.optim.AdamW).  The matrix products of its forward and backward
 * are actmi_op_gemm calls (ta / tb select the transposed operands); these are the remaining pieces.  All fp32, no atomics. */
/* nn.GELU() (exact erf form) and its derivative at the saved pre-activation */
int actmi_op_gelu(const float* x, float* y, int64_t n, void* stream);
int actmi_op_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream);
/* nn.Dropout: y[i] = keep(seed, i) ? x[i] / (1 - p) : 0.  The mask is a pure function of (seed, i): the backward is the same call
 * on the gradient. */
int actmi_op_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream);
/* nn.MultiheadAttention(batch_first) core for short sequences (T <= 64, head_dim <= 64): qkv [n][T][3*H*HD] as in_proj leaves it
 * (q | k | v), out [n][T][H*HD] before out_proj; causal != 0 applies the triu(diagonal=1) mask of latent_model.py:26; drop_p is the
 * attention-weight dropout.  The backward recomputes the weights and writes dqkv [n][T][3*H*HD] = (dq | dk | dv). */
int actmi_op_small_attention(const float* qkv, float* out, int n, int T, int H, int HD, int causal, float drop_p, uint64_t seed,
                             void* stream);
int actmi_op_small_attention_bwd(const float* qkv, const float* dout, float* dqkv, int n, int T, int H, int HD, int causal,
                                 float drop_p, uint64_t seed, void* stream);
/* F.cross_entropy(logits [B][T][V], target [B][T][V]) with probability targets, mean reduction -- the class axis is dim 1 (T), as
 * train_latent_model.py:329 calls it.  loss: one float; dlogits (optional): gradient of the loss; ws: B*V floats of scratch. */
int actmi_op_soft_ce_dim1(const float* logits, const float* target, int B, int T, int V, float* loss, float* dlogits, float* ws,
                          void* stream);
/* mean |one_hot(argmax(logits, -1)) - target| over [rows][V] (train_latent_model.py:331-334); ws: `rows` floats of scratch */
int actmi_op_argmax_l1(const float* logits, const float* target, int rows, int V, float* out, float* ws, void* stream);
/* nn.LayerNorm backward at the saved input x: dx = dx_add (optional) + d/dx, dw += , db += (zero them first); with ws
 * (>= 2*D*min(1024, ceil(M/4)) floats) the parameter gradients are summed in a fixed order */
int actmi_op_layernorm_bwd(const float* x, const float* w, const float* dy, const float* dx_add, float* dx, float* dw, float* db,
                           int M, int D, float eps, float* ws, int64_t ws_floats, void* stream);
/* out[n] += sum_m src[m][n] (bias gradients); with ws (>= ceil(M/256)*N floats) in a fixed order */
int actmi_op_colsum(const float* src, int64_t ld, float* out, int M, int N, float* ws, int64_t ws_floats, void* stream);
/* dst[r][d] (+)= sum_b src[b*batch_stride + r*ld + d] (gradient of a table added to every sample: nn.Embedding positions) */
int actmi_op_sum_batch(const float* src, int64_t batch_stride, int64_t ld, float* dst, int B, int R, int D, int accumulate,
                       void* stream);
/* torch.optim.AdamW update of one flat fp32 tensor (decoupled decay, bias correction with `step` counted from 1) */
int actmi_op_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float weight_decay, float beta1, float beta2,
                   float eps, int64_t step, void* stream);
const char* actmi_op_last_error(void);

/* intermediate activations of the last forward (parity tests): name in {"conv1","maxpool","layer1".."layer4",
 * "src","memory","hs"}; camera-major NHWC for the maps, [B][N][D] for tokens. */
int actmi_debug_tensor(actmi_handle h, const char* name, const float** dev_ptr, int64_t* numel);
/* debug: make the next forwards return right after the named stage ("" = run everything); the maps of the
 * trunk live in rotating buffers, so a stage is only readable when the forward stopped there. */
int actmi_debug_stop_after(actmi_handle h, const char* stage);
/* precision of the handle's forward GEMMs / convolutions: ACTMI_PREC_F32 or ACTMI_PREC_F16X3 (the default, unless the
 * environment says ACTMI_GEMM_PREC=f32).  Call before actmi_finalize (the split weight image is built there). */
int actmi_set_gemm_prec(actmi_handle h, int prec);
/* precision of the GEMMs of the TRAINING step (actmi_forward_train / actmi_backward): 0 = the handle's forward precision (the
 * default: fp32-grade f16x3 products), ACTMI_PREC_BF16 = one bf16 product per fp32 product with fp32 accumulation, fp32 master
 * weights and optimizer state -- BASELINE config 3's "bf16" as written, an opt-in speed mode whose losses / gradients agree
 * with the reference to ~1e-2 relative only (tests/test_gpu_training.py); inference never uses it.  Also ACTMI_TRAIN_PREC=bf16.
 * The direct kernels of the step (stem, layer1, attention forward) keep their f16x3 arithmetic. */
int actmi_set_train_prec(actmi_handle h, int prec);

/* ---- range / finiteness guard (default on) --------------------------------------------------------- */
/* The handle keeps a device flag word raised by its own kernels: ACTMI_FLAG_OUTPUT = an inference output (a_hat) was NaN /
 * infinite -- with the f16x3 products that is how an operand beyond the fp16 range (|x| >= 65504) shows up;
 * ACTMI_FLAG_WEIGHT = a parameter has outgrown the power-of-two scale its split image was calibrated with at finalize
 * (re-run actmi_finalize); ACTMI_FLAG_LOSS = a training loss was not finite.  actmi_get_flags copies the word to the host
 * (it synchronises `stream`: call it at a natural synchronisation point, e.g. right after the actions were copied to the
 * host) and clears it when `clear` is non-zero.  Replaces nothing in the reference (torch propagates NaN silently). */
#define ACTMI_FLAG_OUTPUT 1u
#define ACTMI_FLAG_WEIGHT 2u
#define ACTMI_FLAG_LOSS 4u
int actmi_get_flags(actmi_handle h, uint32_t* host_flags, int clear, void* stream);
/* device address of the flag word (one uint32): data-parallel callers reduce it over the ranks before actmi_adamw_step so
 * that every rank skips the same updates (actmi/engine.py:backward_allreduce) */
int actmi_flags_ptr(actmi_handle h, void** dev_ptr);

/* ---- per-launch HIP-event profiler (bench.py roofline leg) --------------------------------------- */
/* When enabled, every kernel launch of the library is bracketed by two events on its stream.  The report is a
 * JSON array of {"name","count","ms","flops","bytes"} per kernel instantiation (algorithmic flops/bytes). */
int actmi_profile_enable(int on);
int actmi_profile_reset(void);
int actmi_profile_report(char* buf, int buflen);

#ifdef __cplusplus
}
#endif
#endif /* ACTMI_H */
