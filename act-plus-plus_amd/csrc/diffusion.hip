// Kernels of the DiffusionPolicy inference path (reference policy.py:20-241) that are not GEMM-shaped.
// The arithmetic the reference delegates to robomimic (ResNet18Conv with BatchNorm replaced by GroupNorm, SpatialSoftmax,
// ConditionalUnet1D) and diffusers (DDIMScheduler) is restated from their published definitions -- neither package is
// importable offline, so parity of this path is UNPINNED (oracle/diffusion_ref.py is the same restatement on torch CPU ops).
// Dense contractions (2-D convolutions, Conv1d / ConvTranspose1d through the unfold kernels below, Linear) run on the MFMA
// GEMM of gemm.hip; everything here is HBM / latency bound.
//
//   groupnorm        y = act(GN(x) [+ res]) [* film_scale + film_bias] [+ res], channel-last [n][P][C], G groups of
//                    C/G consecutive channels, statistics over (P x C/G) per sample as torch.nn.GroupNorm (biased variance);
//                    act 0 none / 1 ReLU / 2 Mish.  Serves the GroupNorm ResNet (P = H*W, res before the ReLU) and the
//                    UNet's Conv1dBlock + FiLM + residual (P = T)
//   spatial_softmax  robomimic SpatialSoftmax: per keypoint softmax over the H*W positions, expected (x, y) on the
//                    [-1, 1] grid
//   unfold1d         im2col of Conv1d (channel-last): out[b][to][j][c] = x[b][to*stride - pad + j][c] or 0;
//                    transposed = 1 gives the gather form of ConvTranspose1d: out[b][t][j][c] = x[b][(t + pad - j)/stride][c]
//                    when divisible and in range
//   ddim_step        diffusers DDIMScheduler.step (eta 0, epsilon prediction, clip_sample): x0 = clamp((x - s1 eps) / sa),
//                    x_prev = sa_prev x0 + s1_prev eps
//   mish / u8 -> NHWC4 helpers
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float mishf(float x) {
    // x * tanh(softplus(x)), softplus with torch's threshold 20
    const float sp = x > 20.f ? x : log1pf(expf(x));
    return x * tanhf(sp);
}

// one workgroup per (sample, group): two passes over P x cg values (cg = C / G channels, contiguous per pixel)
__global__ __launch_bounds__(256) void groupnorm_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                        const float* __restrict__ fs, const float* __restrict__ fb,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        float* __restrict__ out, int P, int C, int G, float eps, int act,
                                                        int res_mode) {
    const int n = blockIdx.x / G, g = blockIdx.x - n * G;
    const int cg = C / G;
    const int64_t base = (int64_t)n * P * C + (int64_t)g * cg;
    const int64_t total = (int64_t)P * cg;
    // pass 1: mean and biased variance in two sweeps (sum, then sum of squared deviations: no cancellation)
    __shared__ float s_red[4];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < total; i += 256) {
        const int64_t px = i / cg;
        s += x[base + px * C + (i - px * cg)];
    }
    s = wave_sum_f(s);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float mean = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / (float)total;
    __syncthreads();
    float q = 0.f;
    for (int64_t i = threadIdx.x; i < total; i += 256) {
        const int64_t px = i / cg;
        const float d = x[base + px * C + (i - px * cg)] - mean;
        q += d * d;
    }
    q = wave_sum_f(q);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = q;
    __syncthreads();
    const float rstd = 1.f / sqrtf(((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / (float)total + eps);
    for (int64_t i = threadIdx.x; i < total; i += 256) {
        const int64_t px = i / cg;
        const int c = g * cg + (int)(i - px * cg);
        const int64_t o = (int64_t)n * P * C + px * C + c;
        float v = (x[o] - mean) * rstd * w[c] + b[c];
        if (res_mode == 1) v += res[o];
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = mishf(v);
        if (fs) v = v * fs[(int64_t)n * C + c] + fb[(int64_t)n * C + c];
        if (res_mode == 2) v += res[o];
        out[o] = v;
    }
}

// ---- the same for LARGE maps (the ResNet trunk: 32 samples x 4 groups would be 128 workgroups striding over a million values
//      each).  Pass 1: workgroup (chunk, sample x group) takes a range of pixels and leaves (count, mean, M2) of its values
//      (mean first, then squared deviations from it: the chunk is read twice, the second time from L2); pass 2: every thread
//      combines its (sample, group)'s chunk statistics in chunk order (Chan's parallel-variance update: no cancellation, the
//      same result in every thread) and normalises 16-byte pieces.  cg % 4 == 0.
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, float* __restrict__ ws, int P, int C, int G,
                                                       int nch) {
    const int ng = blockIdx.y, n = ng / G, g = ng - n * G, ch = blockIdx.x;
    const int cg = C / G, cg4 = cg >> 2;
    const int p0 = (int)((int64_t)P * ch / nch), p1 = (int)((int64_t)P * (ch + 1) / nch);
    const float* xb = x + (int64_t)n * P * C + (int64_t)g * cg;
    const int tot4 = (p1 - p0) * cg4;
    __shared__ float s_red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < tot4; i += 256) {
        const int px = i / cg4, c4 = i - px * cg4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (int64_t)(p0 + px) * C + c4 * 4);
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    s = wave_sum_f(s);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float cnt = (float)(tot4 * 4);
    const float mean = cnt > 0.f ? ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / cnt : 0.f;
    __syncthreads();
    float q = 0.f;
    for (int i = threadIdx.x; i < tot4; i += 256) {
        const int px = i / cg4, c4 = i - px * cg4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (int64_t)(p0 + px) * C + c4 * 4) - mean;
        q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    q = wave_sum_f(q);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = ws + ((int64_t)ng * nch + ch) * 3;
        o[0] = cnt; o[1] = mean; o[2] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                       const float* __restrict__ fs, const float* __restrict__ fb,
                                                       const float* __restrict__ w, const float* __restrict__ b,
                                                       float* __restrict__ out, const float* __restrict__ ws, int P, int C, int G,
                                                       int nch, float eps, int act, int res_mode) {
    // blockIdx.y = sample x group; blockIdx.x strides over the (pixel, 4-channel piece) pairs of that group
    const int ng = blockIdx.y, n = ng / G, g = ng - n * G;
    const int cg = C / G, cg4 = cg >> 2;
    float cnt = 0.f, mean = 0.f, m2 = 0.f;
    for (int ch = 0; ch < nch; ++ch) {
        const float* o = ws + ((int64_t)ng * nch + ch) * 3;
        const float cb = o[0], mb = o[1], qb = o[2];
        if (cb > 0.f) {
            const float ct = cnt + cb, d = mb - mean;
            mean += d * (cb / ct);
            m2 += qb + d * d * (cnt * cb / ct);
            cnt = ct;
        }
    }
    const float rstd = 1.f / sqrtf(m2 / cnt + eps);
    const int64_t base = (int64_t)n * P * C + (int64_t)g * cg;
    const unsigned tot4 = (unsigned)P * cg4;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < tot4; i += gridDim.x * 256u) {
        const unsigned px = i / cg4, c4 = i - px * cg4;
        const int c = g * cg + (int)c4 * 4;
        const int64_t o = base + (int64_t)px * C + c4 * 4;
        f32x4 v = (*reinterpret_cast<const f32x4*>(x + o) - mean) * rstd * *reinterpret_cast<const f32x4*>(w + c) +
                  *reinterpret_cast<const f32x4*>(b + c);
        if (res_mode == 1) v += *reinterpret_cast<const f32x4*>(res + o);
        if (act == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (act == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = mishf(v[e]);
        }
        if (fs) v = v * *reinterpret_cast<const f32x4*>(fs + (int64_t)n * C + c) + *reinterpret_cast<const f32x4*>(fb + (int64_t)n * C + c);
        if (res_mode == 2) v += *reinterpret_cast<const f32x4*>(res + o);
        *reinterpret_cast<f32x4*>(out + o) = v;
    }
}

// logits [n][P][K] (channel-last output of the 1x1 keypoint convolution), one wave per (sample, keypoint)
__global__ __launch_bounds__(64) void spatial_softmax_kernel(const float* __restrict__ logits, float* __restrict__ out, int H, int W,
                                                             int K, float inv_temp) {
    const int n = blockIdx.x / K, k = blockIdx.x - n * K;
    const int P = H * W, lane = threadIdx.x;
    const float* lg = logits + (int64_t)n * P * K + k;
    float m = -INFINITY;
    for (int p = lane; p < P; p += 64) m = fmaxf(m, lg[(int64_t)p * K] * inv_temp);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float se = 0.f, sx = 0.f, sy = 0.f;
    for (int p = lane; p < P; p += 64) {
        const float e = expf(lg[(int64_t)p * K] * inv_temp - m);
        const int h = p / W, ww = p - h * W;
        // np.linspace(-1, 1, W)[ww], np.linspace(-1, 1, H)[h]
        const float px = W > 1 ? -1.f + 2.f * (float)ww / (float)(W - 1) : -1.f;
        const float py = H > 1 ? -1.f + 2.f * (float)h / (float)(H - 1) : -1.f;
        se += e; sx += e * px; sy += e * py;
    }
    se = wave_sum_f(se); sx = wave_sum_f(sx); sy = wave_sum_f(sy);
    if (lane == 0) {
        out[((int64_t)n * K + k) * 2 + 0] = sx / se;
        out[((int64_t)n * K + k) * 2 + 1] = sy / se;
    }
}

__global__ void unfold1d_kernel(const float* __restrict__ x, float* __restrict__ out, int T, int C4, int To, int k, int stride,
                                int pad, int transposed, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // (b, to, j, c4)
    if (idx >= total) return;
    const int c = (int)(idx % C4);
    int64_t r = idx / C4;
    const int j = (int)(r % k); r /= k;
    const int to = (int)(r % To);
    const int64_t b = r / To;
    int ti;
    bool ok;
    if (transposed) {
        const int num = to + pad - j;
        ti = num / stride;
        ok = num >= 0 && ti * stride == num && ti < T;
    } else {
        ti = to * stride - pad + j;
        ok = ti >= 0 && ti < T;
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok) v = reinterpret_cast<const f32x4*>(x)[(b * T + ti) * C4 + c];
    reinterpret_cast<f32x4*>(out)[idx] = v;
}

__global__ void ddim_step_kernel(float* __restrict__ x, const float* __restrict__ eps, int64_t n, float inv_sqrt_at, float sqrt_1m_at,
                                 float sqrt_aprev, float sqrt_1m_aprev, int clip) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float e = eps[i];
    float x0 = (x[i] - sqrt_1m_at * e) * inv_sqrt_at;
    if (clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
    x[i] = sqrt_aprev * x0 + sqrt_1m_aprev * e;
}

__global__ void mish_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = mishf(x[i]);
}

// u8 [B][Cam][H][W][3] -> f32 camera-major NHWC4 [Cam][B][H][W][4] = v / 255 (float(v / 255.0) as imitate_episodes.py:212), pad 0
__global__ void u8_to_nhwc4_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, int B, int Cam, int64_t HW,
                                   int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // (cam, b, pixel)
    if (idx >= total) return;
    const int64_t px = idx % HW;
    const int64_t r = idx / HW;
    const int b = (int)(r % B), cam = (int)(r / B);
    const uint8_t* src = img + (((int64_t)b * Cam + cam) * HW + px) * 3;
    f32x4 v;
    v[0] = (float)((double)src[0] / 255.0); v[1] = (float)((double)src[1] / 255.0); v[2] = (float)((double)src[2] / 255.0); v[3] = 0.f;
    reinterpret_cast<f32x4*>(out)[idx] = v;
}

thread_local std::string g_err;
inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
int done() { return hipGetLastError() == hipSuccess ? 0 : ACTMI_E_LAUNCH; }

}  // namespace

extern "C" {

int actmi_op_groupnorm(const float* x, const float* res, const float* film_scale, const float* film_bias, const float* w,
                       const float* b, float* out, int n, int P, int C, int G, float eps, int act, int res_mode, float* ws,
                       int64_t ws_floats, void* stream) {
    if (!x || !w || !b || !out || n < 1 || P < 1 || C < 1 || G < 1 || C % G || act < 0 || act > 2 || res_mode < 0 || res_mode > 2 ||
        (res_mode && !res) || (!film_scale != !film_bias))
        return ACTMI_E_INVALID;
    prof_begin("groupnorm_kernel", 0.0, 4.0 * n * (double)P * C * (res ? 5.0 : 4.0), S(stream));
    const int cg = C / G;
    const int64_t per = (int64_t)P * cg;
    // large maps: chunked statistics + a vectorised apply pass (needs the workspace: 3 floats per sample, group and chunk)
    if (ws && per >= 16384 && (cg & 3) == 0 && (C & 3) == 0 && per / 4 < ((int64_t)1 << 31) && (int64_t)n * G <= 65535 &&
        !(((uintptr_t)x | (uintptr_t)out | (uintptr_t)w | (uintptr_t)b | (uintptr_t)res | (uintptr_t)film_scale | (uintptr_t)film_bias) & 15)) {
        int nch = (int)(per / 16384);                      // ~16k values per chunk
        const int64_t want = 2048 / ((int64_t)n * G) + 1;  // ... but no more chunks than fill the chip a few times over
        if (nch > want) nch = (int)want;
        if (nch > 256) nch = 256;
        if (nch < 1) nch = 1;
        if ((int64_t)n * G * nch * 3 <= ws_floats) {
            hipLaunchKernelGGL(gn_stats_kernel, dim3(nch, n * G), dim3(256), 0, S(stream), x, ws, P, C, G, nch);
            int bx = (int)((per / 4 + 256 * 8 - 1) / (256 * 8));
            if (bx > 64) bx = 64;
            hipLaunchKernelGGL(gn_apply_kernel, dim3(bx, n * G), dim3(256), 0, S(stream), x, res, film_scale, film_bias, w, b, out, ws, P,
                               C, G, nch, eps, act, res_mode);
            prof_end(S(stream));
            return done();
        }
    }
    hipLaunchKernelGGL(groupnorm_kernel, dim3((unsigned)(n * G)), dim3(256), 0, S(stream), x, res, film_scale, film_bias, w, b, out, P,
                       C, G, eps, act, res_mode);
    prof_end(S(stream));
    return done();
}

int actmi_op_spatial_softmax(const float* logits, float* out, int n, int H, int W, int K, float temperature, void* stream) {
    if (!logits || !out || n < 1 || H < 1 || W < 1 || K < 1 || !(temperature > 0.f)) return ACTMI_E_INVALID;
    hipLaunchKernelGGL(spatial_softmax_kernel, dim3((unsigned)(n * K)), dim3(64), 0, S(stream), logits, out, H, W, K, 1.f / temperature);
    return done();
}

int actmi_op_unfold1d(const float* x, float* out, int B, int T, int C, int k, int stride, int pad, int To, int transposed,
                      void* stream) {
    if (!x || !out || B < 1 || T < 1 || C < 4 || (C & 3) || k < 1 || stride < 1 || pad < 0 || To < 1) return ACTMI_E_INVALID;
    const int64_t total = (int64_t)B * To * k * (C / 4);
    hipLaunchKernelGGL(unfold1d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S(stream), x, out, T, C / 4, To, k, stride,
                       pad, transposed, total);
    return done();
}

int actmi_op_ddim_step(float* x, const float* eps, int64_t n, float inv_sqrt_at, float sqrt_1m_at, float sqrt_aprev,
                       float sqrt_1m_aprev, int clip, void* stream) {
    if (!x || !eps || n < 1) return ACTMI_E_INVALID;
    hipLaunchKernelGGL(ddim_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), x, eps, n, inv_sqrt_at, sqrt_1m_at,
                       sqrt_aprev, sqrt_1m_aprev, clip);
    return done();
}

int actmi_op_mish(const float* x, float* y, int64_t n, void* stream) {
    if (!x || !y || n < 1) return ACTMI_E_INVALID;
    hipLaunchKernelGGL(mish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), x, y, n);
    return done();
}

int actmi_op_u8_to_nhwc4(const uint8_t* image, float* out, int B, int Cam, int H, int W, void* stream) {
    if (!image || !out || B < 1 || Cam < 1 || H < 1 || W < 1) return ACTMI_E_INVALID;
    const int64_t total = (int64_t)Cam * B * H * W;
    hipLaunchKernelGGL(u8_to_nhwc4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S(stream), image, out, B, Cam,
                       (int64_t)H * W, total);
    return done();
}

}  // extern "C"
