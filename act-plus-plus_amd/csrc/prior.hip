// Kernels for training the VQ-ACT latent prior (reference detr/models/latent_model.py:8-56, train_latent_model.py:323-343):
// a 3-block causal transformer over vq_class (= 32) positions of width 256.  Every matrix product of its forward and backward
// goes through the MFMA GEMM (actmi_op_gemm, native fp32 products); what is left is small and lives here in fp32 VALU:
// exact GELU and its derivative, causal self-attention over <= 64 positions with the attention-weight dropout of
// nn.MultiheadAttention (one workgroup per (sample, head), forward and a deterministic backward that recomputes the weights),
// the soft-target cross entropy F.cross_entropy applies when it is handed [B, T, V] logits and [B, T, V] probabilities (the class
// axis is dim 1 -- the SEQUENCE axis; the reference's call, kept as written), the argmax / one-hot L1 metric, and torch's AdamW.
// Nothing here uses atomics: results are bitwise repeatable.
#include "common.h"
#include "dropout.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- GELU (nn.GELU default: exact erf form, the same expression as the GEMM epilogue's) -----------------------------------
__global__ void gelu_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float v = x[i]; y[i] = 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
}
// dx = dy * (Phi(x) + x * phi(x))
__global__ void gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float v = x[i];
        const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * expf(-0.5f * v * v);
        dx[i] = dy[i] * (cdf + v * pdf);
    }
}

__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, uint64_t seed, float p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = actmi_keep(seed, (uint64_t)i, p) ? x[i] * (1.f / (1.f - p)) : 0.f;
}

// ---- causal self-attention over T <= 64 positions, head width HD <= 64 -----------------------------------------------------
// qkv [n][T][3D] packed as nn.MultiheadAttention's in_proj leaves it (q | k | v, D = H * HD); one workgroup of four waves per
// (sample, head); a wave owns query rows, its lanes are the keys.  Weights dropped with keep(seed, ((g * T + q) * T + key)).
struct SmallAttn {
    const float* qkv; const float* dout; float* out; float* dqkv;
    int T, H, HD, causal; float scale, drop_p; uint64_t seed;
};

__device__ __forceinline__ void load_tiles(const SmallAttn& a, int g, float* sQ, float* sK, float* sV, float* sdO) {
    const int b = g / a.H, h = g - b * a.H, D = a.H * a.HD, ldt = a.HD + 1;
    const float* base = a.qkv + (int64_t)b * a.T * 3 * D + h * a.HD;
    for (int idx = threadIdx.x; idx < a.T * a.HD; idx += blockDim.x) {
        const int t = idx / a.HD, d = idx - t * a.HD;
        const float* r = base + (int64_t)t * 3 * D + d;
        sQ[t * ldt + d] = r[0];
        sK[t * ldt + d] = r[D];
        sV[t * ldt + d] = r[2 * D];
        if (sdO) sdO[t * ldt + d] = a.dout[((int64_t)b * a.T + t) * D + h * a.HD + d];
    }
}

// softmax row of query q over the lanes (keys); returns the UNdropped weight of this lane's key, 0 outside the mask
__device__ __forceinline__ float prob_row(const SmallAttn& a, const float* sQ, const float* sK, int q, int lane) {
    const int ldt = a.HD + 1;
    const bool live = lane < a.T && (!a.causal || lane <= q);
    float s = -INFINITY;
    if (live) {
        float acc = 0.f;
        for (int d = 0; d < a.HD; ++d) acc = fmaf(sQ[q * ldt + d], sK[lane * ldt + d], acc);
        s = acc * a.scale;
    }
    const float m = wave_max(s);
    const float e = live ? expf(s - m) : 0.f;
    const float sum = wave_sum(e);
    return e / sum;
}

__global__ __launch_bounds__(256) void small_attn_fwd_kernel(SmallAttn a) {
    extern __shared__ float sm[];
    const int ldt = a.HD + 1, g = blockIdx.x;
    float* sQ = sm; float* sK = sQ + a.T * ldt; float* sV = sK + a.T * ldt; float* sP = sV + a.T * ldt;   // sP [4][64]
    load_tiles(a, g, sQ, sK, sV, nullptr);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = g / a.H, h = g - b * a.H, D = a.H * a.HD;
    for (int q0 = 0; q0 < a.T; q0 += 4) {
        const int q = q0 + wave;
        if (q < a.T) {
            float p = prob_row(a, sQ, sK, q, lane);
            if (a.drop_p > 0.f) p = actmi_keep(a.seed, ((uint64_t)g * a.T + q) * a.T + lane, a.drop_p) ? p * (1.f / (1.f - a.drop_p)) : 0.f;
            sP[wave * 64 + lane] = p;
        }
        __syncthreads();
        if (q < a.T && lane < a.HD) {
            float o = 0.f;
            for (int j = 0; j < a.T; ++j) o = fmaf(sP[wave * 64 + j], sV[j * ldt + lane], o);
            a.out[((int64_t)b * a.T + q) * D + h * a.HD + lane] = o;
        }
        __syncthreads();
    }
}

// dqkv [n][T][3D] = (dQ | dK | dV); the weights are recomputed from q and k, the whole T x T matrices of the dropped weights and
// of dS sit in LDS, and every output element is one thread's ordered sum
__global__ __launch_bounds__(256) void small_attn_bwd_kernel(SmallAttn a) {
    extern __shared__ float sm[];
    const int ldt = a.HD + 1, ldp = a.T + 1, g = blockIdx.x;
    float* sQ = sm; float* sK = sQ + a.T * ldt; float* sV = sK + a.T * ldt; float* sdO = sV + a.T * ldt;
    float* sP = sdO + a.T * ldt; float* sdS = sP + a.T * ldp;
    load_tiles(a, g, sQ, sK, sV, sdO);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int q = wave; q < a.T; q += 4) {
        const float p = prob_row(a, sQ, sK, q, lane);
        float dpd = 0.f;                                   // gradient of the DROPPED weight: dO[q] . V[key]
        if (lane < a.T) for (int d = 0; d < a.HD; ++d) dpd = fmaf(sdO[q * ldt + d], sV[lane * ldt + d], dpd);
        float keep_scale = 1.f;
        if (a.drop_p > 0.f) keep_scale = actmi_keep(a.seed, ((uint64_t)g * a.T + q) * a.T + lane, a.drop_p) ? 1.f / (1.f - a.drop_p) : 0.f;
        const float dp = dpd * keep_scale;
        const float delta = wave_sum(p * dp);
        if (lane < a.T) {
            sP[q * ldp + lane] = p * keep_scale;
            sdS[q * ldp + lane] = p * (dp - delta) * a.scale;
        }
    }
    __syncthreads();
    const int b = g / a.H, h = g - b * a.H, D = a.H * a.HD;
    for (int r = wave; r < a.T; r += 4) {
        if (lane < a.HD) {
            float dq = 0.f, dk = 0.f, dv = 0.f;
            for (int j = 0; j < a.T; ++j) {
                dq = fmaf(sdS[r * ldp + j], sK[j * ldt + lane], dq);
                dk = fmaf(sdS[j * ldp + r], sQ[j * ldt + lane], dk);
                dv = fmaf(sP[j * ldp + r], sdO[j * ldt + lane], dv);
            }
            float* o = a.dqkv + ((int64_t)b * a.T + r) * 3 * D + h * a.HD + lane;
            o[0] = dq; o[D] = dk; o[2 * D] = dv;
        }
    }
}

// ---- F.cross_entropy(logits [B,T,V], target [B,T,V]) with probability targets: the class axis is dim 1 ---------------------
// pair (b, v): lse over t; loss_pair = -sum_t target * (x - lse); d logits = (softmax_t(x) * sum_t target - target) / (B * V)
__global__ void soft_ce_dim1_kernel(const float* __restrict__ x, const float* __restrict__ tg, float* __restrict__ dx,
                                    float* __restrict__ pair_loss, int B, int T, int V, float inv_count) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * V) return;
    const int b = idx / V, v = idx - b * V;
    const float* xr = x + (int64_t)b * T * V + v;
    const float* tr = tg + (int64_t)b * T * V + v;
    float m = -INFINITY;
    for (int t = 0; t < T; ++t) m = fmaxf(m, xr[(int64_t)t * V]);
    float s = 0.f, tsum = 0.f;
    for (int t = 0; t < T; ++t) { s += expf(xr[(int64_t)t * V] - m); tsum += tr[(int64_t)t * V]; }
    const float lse = m + logf(s);
    float l = 0.f;
    for (int t = 0; t < T; ++t) l -= tr[(int64_t)t * V] * (xr[(int64_t)t * V] - lse);
    pair_loss[idx] = l;
    if (dx) {
        float* dr = dx + (int64_t)b * T * V + v;
        for (int t = 0; t < T; ++t) dr[(int64_t)t * V] = (expf(xr[(int64_t)t * V] - lse) * tsum - tr[(int64_t)t * V]) * inv_count;
    }
}

// row r: sum_v |onehot(argmax_v x[r][:])[v] - target[r][v]|   (first maximum on ties, as torch.argmax)
__global__ void argmax_l1_kernel(const float* __restrict__ x, const float* __restrict__ tg, float* __restrict__ row_sum, int rows, int V) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* xr = x + (int64_t)r * V;
    const float* tr = tg + (int64_t)r * V;
    int best = 0; float bv = xr[0];
    for (int v = 1; v < V; ++v) if (xr[v] > bv) { bv = xr[v]; best = v; }
    float s = 0.f;
    for (int v = 0; v < V; ++v) s += fabsf((v == best ? 1.f : 0.f) - tr[v]);
    row_sum[r] = s;
}

// out[0] = scale * sum(x[0..n)) in a fixed order: one workgroup, strided per-thread sums, tree over the threads
__global__ __launch_bounds__(256) void sum_ordered_kernel(const float* __restrict__ x, int64_t n, float scale, float* __restrict__ out) {
    __shared__ float s[256];
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) acc += x[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s[0] * scale;
}

// torch.optim.AdamW (decoupled decay first, bias-corrected moments, eps outside the square root of the corrected v)
__global__ void adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                  int64_t n, float lr, float wd, float b1, float b2, float eps, float bc1, float bc2_sqrt) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float pi = p[i] * (1.f - lr * wd);
        const float gi = g[i];
        const float mi = m[i] * b1 + gi * (1.f - b1);
        const float vi = v[i] * b2 + gi * gi * (1.f - b2);
        m[i] = mi; v[i] = vi;
        p[i] = pi - (lr / bc1) * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    }
}

int check_small_attn(int n, int T, int H, int HD) {
    return (n < 1 || T < 1 || T > 64 || H < 1 || HD < 1 || HD > 64) ? -2 : 0;
}
}  // namespace

int launch_gelu(const float* x, float* y, int64_t n, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(gelu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, n);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int launch_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, dy, dx, n);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int launch_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, hipStream_t st) {
    if (n <= 0) return 0;
    if (!(p >= 0.f && p < 1.f)) return -2;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, seed, p, n);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_small_attention(const float* qkv, float* out, int n, int T, int H, int HD, int causal, float drop_p, uint64_t seed,
                           hipStream_t st) {
    if (check_small_attn(n, T, H, HD) || !(drop_p >= 0.f && drop_p < 1.f)) return -2;
    SmallAttn a{qkv, nullptr, out, nullptr, T, H, HD, causal, 1.0f / sqrtf((float)HD), drop_p, seed};
    const size_t lds = ((size_t)3 * T * (HD + 1) + 4 * 64) * sizeof(float);        // <= 51 KB
    hipLaunchKernelGGL(small_attn_fwd_kernel, dim3(n * H), dim3(256), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_small_attention_bwd(const float* qkv, const float* dout, float* dqkv, int n, int T, int H, int HD, int causal, float drop_p,
                               uint64_t seed, hipStream_t st) {
    if (check_small_attn(n, T, H, HD) || !(drop_p >= 0.f && drop_p < 1.f)) return -2;
    SmallAttn a{qkv, dout, nullptr, dqkv, T, H, HD, causal, 1.0f / sqrtf((float)HD), drop_p, seed};
    const size_t lds = ((size_t)4 * T * (HD + 1) + 2 * T * (T + 1)) * sizeof(float);   // <= 100 KB of the CU's 160 KB
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(small_attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                100 * 1024) != hipSuccess) return -3;
        attr_set = true;
    }
    hipLaunchKernelGGL(small_attn_bwd_kernel, dim3(n * H), dim3(256), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_soft_ce_dim1(const float* logits, const float* target, int B, int T, int V, float* loss, float* dlogits, float* ws,
                        hipStream_t st) {
    if (B < 1 || T < 1 || V < 1 || !ws || !loss) return -2;
    const float inv = 1.0f / ((float)B * (float)V);
    hipLaunchKernelGGL(soft_ce_dim1_kernel, dim3((B * V + 255) / 256), dim3(256), 0, st, logits, target, dlogits, ws, B, T, V, inv);
    hipLaunchKernelGGL(sum_ordered_kernel, dim3(1), dim3(256), 0, st, ws, (int64_t)B * V, inv, loss);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_argmax_l1(const float* logits, const float* target, int rows, int V, float* out, float* ws, hipStream_t st) {
    if (rows < 1 || V < 1 || !ws || !out) return -2;
    hipLaunchKernelGGL(argmax_l1_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, logits, target, ws, rows, V);
    hipLaunchKernelGGL(sum_ordered_kernel, dim3(1), dim3(256), 0, st, ws, (int64_t)rows, 1.0f / ((float)rows * (float)V), out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float wd, float b1, float b2, float eps,
                      int64_t step, hipStream_t st) {
    if (n <= 0) return 0;
    if (step < 1) return -2;
    const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adamw_flat_kernel, dim3(blocks), dim3(256), 0, st, p, g, m, v, n, lr, wd, b1, b2, eps, bc1, sqrtf(bc2));
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
