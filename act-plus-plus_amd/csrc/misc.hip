// Small kernels of the ACT path: tiny-K linears, token rows, weight repacking, temporal ensembling.
#include "common.h"
#include "split16.h"

namespace {

// y[m][n] = sum_k x[m][k] w[n][k] + b[n]  for the K in {14, 16, 32} projections
// (input_proj_robot_state / encoder_joint_proj / encoder_action_proj / latent_out_proj; detr_vae.py:63,81-82,97)
__global__ void small_linear_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                    const float* __restrict__ b, float* __restrict__ y, int64_t ldy, int M, int N,
                                    int K, float* __restrict__ fill_dst, const float* __restrict__ fill_src, int64_t fill_src_bs) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)M * N) return;
    const int m = (int)(idx / N), n = (int)(idx - (int64_t)m * N);
    const float* xr = x + (int64_t)m * ldx;
    const float* wr = w + (int64_t)n * K;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(xr[k], wr[k], acc);
    y[(int64_t)m * ldy + n] = acc + (b ? b[n] : 0.f);
    // optional companion row (same M x N index space): fill_dst[m][n] = fill_src[m * fill_src_bs + n] -- the latent token
    // next to the proprio token of the encoder input (detr_vae.py:158-159, 213), one launch instead of two
    if (fill_dst) fill_dst[(int64_t)m * ldy + n] = fill_src[(int64_t)m * fill_src_bs + n];
}

__global__ void fill_rows_kernel(float* __restrict__ dst, int64_t batch_stride, const float* __restrict__ src,
                                 int64_t src_bs, int B, int D) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * D) return;
    const int b = (int)(idx / D), d = (int)(idx - (int64_t)b * D);
    dst[(int64_t)b * batch_stride + d] = src[(int64_t)b * src_bs + d];
}

// [G][O][I][KH][KW] -> [G][O][kpad] with k = (r*KW + s)*I + c, zero padded
__global__ void repack_conv_w_kernel(const float* __restrict__ in, float* __restrict__ out, int O, int I, int KH,
                                     int KW, int64_t g_in, int64_t g_out, int kpad, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int k = (int)(idx % kpad);
    int64_t rest = idx / kpad;
    const int o = (int)(rest % O);
    const int g = (int)(rest / O);
    float v = 0.f;
    if (k < KH * KW * I) {
        const int c = k % I, rs = k / I, s = rs % KW, r = rs / KW;
        v = in[g * g_in + (((int64_t)o * I + c) * KH + r) * KW + s];
    }
    out[g * g_out + (int64_t)o * kpad + k] = v;
}

// Temporal ensembling over E episodes (reference imitate_episodes.py:338-339, 402-411), one 256-thread workgroup per episode.
// ring[e][slot = t % Q][i][a] holds the chunk predicted at time t; at step t the rows r in [t-Q+1, t] contribute
// chunk_r[t-r].  A row counts only when ALL its A values are != 0 ("actions_populated"); weights
// exp(-k*i), i = 0 for the OLDEST populated row, normalised; products and sum in float64 as in the reference
// (numpy float64 weights promote the float32 actions).  Every sum has a fixed shape (lane / wave / row-class order), so a
// step is bitwise repeatable.
__global__ __launch_bounds__(256) void ensemble_kernel(float* __restrict__ ring, int* __restrict__ tcount,
                                                       const float* __restrict__ chunk, double k,
                                                       double* __restrict__ out, uint8_t* __restrict__ populated,
                                                       int Q, int A) {
    // LDS: Q doubles (normalised weight of row j, 0 when the row is not populated) + fixed scratch
    extern __shared__ __attribute__((aligned(8))) unsigned char s_raw[];
    double* s_w = reinterpret_cast<double*>(s_raw);
    __shared__ int s_cnt[4];
    __shared__ double s_ws[4];
    __shared__ double s_acc[256];
    const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = tcount[e];
    float* rg = ring + (int64_t)e * Q * Q * A;
    const float* ch = chunk + (int64_t)e * Q * A;
    float* slot = rg + (int64_t)(t % Q) * Q * A;
    for (int i = tid; i < Q * A; i += 256) slot[i] = ch[i];
    __syncthreads();
    // rows oldest first: j -> r = t-(Q-1)+j; populated = all A values non-zero (imitate_episodes.py:405-406);
    // weight exp(-k*i) with i = rank of the row among the populated ones (:407-409)
    int base = 0;
    double wpart = 0.0;
    for (int j0 = 0; j0 < Q; j0 += 256) {
        const int j = j0 + tid;
        const int r = t - (Q - 1) + j;
        bool pop = false;
        if (j < Q && r >= 0) {
            const float* row = rg + ((int64_t)(r % Q) * Q + (t - r)) * A;
            pop = true;
            for (int a = 0; a < A; ++a) pop = pop && (row[a] != 0.f);
        }
        const unsigned long long m = __ballot(pop);
        if (lane == 0) s_cnt[wave] = __popcll(m);
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; ++w) before += s_cnt[w];
        const int idx = base + before + __popcll(m & ((1ull << lane) - 1ull));
        base += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        const double w = pop ? exp(-k * (double)idx) : 0.0;
        if (j < Q) {
            s_w[j] = w;
            if (populated) populated[(int64_t)e * Q + j] = pop ? 1 : 0;
        }
        wpart += w;
        __syncthreads();                                   // s_cnt is rewritten by the next chunk of rows
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wpart += __shfl_xor(wpart, o);
    if (lane == 0) s_ws[wave] = wpart;
    __syncthreads();
    const double wsum = (s_ws[0] + s_ws[1]) + (s_ws[2] + s_ws[3]);
    // threads (a, g): action component a, row class g (rows j = g mod ng); A <= 64
    const int per = A <= 16 ? 16 : (A <= 32 ? 32 : 64);
    const int a = tid % per, g = tid / per, ng = 256 / per;
    double acc = 0.0;
    if (a < A) {
        for (int j = g; j < Q; j += ng) {
            const double w = s_w[j];
            if (w != 0.0) {
                const int r = t - (Q - 1) + j;
                acc += (double)rg[((int64_t)(r % Q) * Q + (t - r)) * A + a] * (w / wsum);
            }
        }
    }
    s_acc[tid] = acc;
    __syncthreads();
    if (tid < A) {
        double tot = 0.0;
        for (int gg = 0; gg < ng; ++gg) tot += s_acc[gg * per + tid];
        out[(int64_t)e * A + tid] = tot;
    }
    if (tid == 0) tcount[e] = t + 1;
}

// FrozenBatchNorm2d folded to a per-channel affine, reference backbone.py:47-57:
// scale = w * rsqrt(rv + 1e-5); bias = b - rm * scale   (IEEE sqrt and divide, as torch's CPU rsqrt)
__global__ void bn_fold_kernel(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ rm,
                               const float* __restrict__ rv, float* __restrict__ scale, float* __restrict__ bias, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = w[i] * (1.0f / sqrtf(rv[i] + 1e-5f));
    scale[i] = s;
    bias[i] = b[i] - rm[i] * s;
}

// feature row m = ((cam*B + b)*fh + h)*fw + w  ->  token row b*N + 2 + h*(fw*C) + cam*fw + w
// (concat along width at detr_vae.py:216, flatten(2).permute at transformer.py:60, 2 extra tokens in front :102)
__global__ void build_rowmap_kernel(int* __restrict__ map, int B, int C, int fh, int fw, int N) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= C * B * fh * fw) return;
    const int w = m % fw;
    int r = m / fw;
    const int h = r % fh; r /= fh;
    const int b = r % B;
    const int cam = r / B;
    map[m] = b * N + 2 + h * (fw * C) + cam * fw + w;
}

// image (u8 NHWC [B][C][H][W][3] or f32 NCHW [B][C][3][H][W]) -> normalised f32 camera-major NHWC4 [C][B][H][W][4]
// (4th channel zero) so that the conv1 weight gradient can use the generic implicit-GEMM gather (Cin % 4 == 0)
__global__ void normalize_pad_kernel(const void* __restrict__ image, int fmt, const float* __restrict__ lut,
                                     float* __restrict__ out, int B, int C, int H, int W, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // over [C][B][H][W]
    if (idx >= total) return;
    const int w = (int)(idx % W);
    int64_t r = idx / W;
    const int h = (int)(r % H); r /= H;
    const int b = (int)(r % B);
    const int cam = (int)(r / B);
    const int64_t img = (int64_t)b * C + cam;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (fmt == 0) {
        const uint8_t* src = reinterpret_cast<const uint8_t*>(image) + (img * H * W + (int64_t)h * W + w) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = lut[c * 256 + src[c]];
    } else {
        const float* src = reinterpret_cast<const float*>(image) + img * 3 * H * W + (int64_t)h * W + w;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (src[(int64_t)c * H * W] - mean[c]) / stdv[c];
    }
    reinterpret_cast<f32x4*>(out)[idx] = v;
}

// dst[m][:] = src[map[m]][:]
__global__ void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ map, float* __restrict__ dst,
                                   int D4, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int64_t m = idx / D4;
    const int d = (int)(idx - m * D4);
    reinterpret_cast<f32x4*>(dst)[idx] = reinterpret_cast<const f32x4*>(src)[(int64_t)map[m] * D4 + d];
}

// dW[n][k] += sum_m dy[m*lddy + n] * x[m*ldx + k]   (tiny K such as the 14-wide qpos projections)
__global__ void small_linear_wgrad_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x,
                                          int64_t ldx, float* __restrict__ dW, int M, int N, int K) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * K) return;
    const int n = (int)(idx / K), k = (int)(idx - (int64_t)n * K);
    float acc = 0.f;
    for (int m = 0; m < M; ++m) acc = fmaf(dy[(int64_t)m * lddy + n], x[(int64_t)m * ldx + k], acc);
    dW[idx] += acc;
}

// CVAE token rows: map[m = b*Q + t] = b*(Q+2) + 2 + t ; key padding mask [B][Q+2] = [0, 0, is_pad]
__global__ void cvae_maps_kernel(int* __restrict__ map, uint8_t* __restrict__ kpm, const uint8_t* __restrict__ is_pad, int B,
                                 int Q) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < B * Q) map[idx] = (idx / Q) * (Q + 2) + 2 + idx % Q;
    if (idx < B * (Q + 2)) {
        const int b = idx / (Q + 2), j = idx % (Q + 2);
        kpm[idx] = j < 2 ? 0 : is_pad[b * Q + j - 2];
    }
}

__global__ void bcast_add_rows_kernel(float* __restrict__ dst, const float* __restrict__ vec, int D, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) dst[i] += vec[i % D];
}

__global__ void axpy_rows_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

}  // namespace

int launch_normalize_pad(const void* image, int fmt, const float* lut, float* out, int B, int C, int H, int W,
                         hipStream_t st) {
    const int64_t total = (int64_t)C * B * H * W;
    hipLaunchKernelGGL(normalize_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, image, fmt, lut, out, B, C,
                       H, W, total);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_gather_rows(const float* src, const int* map, float* dst, int M, int D, hipStream_t st) {
    const int64_t total = (int64_t)M * (D / 4);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, map, dst, D / 4, total);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_small_linear_wgrad(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dW, int M, int N, int K,
                              hipStream_t st) {
    const int64_t total = (int64_t)N * K;
    hipLaunchKernelGGL(small_linear_wgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dy, lddy, x, ldx, dW, M,
                       N, K);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_cvae_maps(int* map, uint8_t* kpm, const uint8_t* is_pad, int B, int Q, hipStream_t st) {
    const int total = B * (Q + 2);
    hipLaunchKernelGGL(cvae_maps_kernel, dim3((total + 255) / 256), dim3(256), 0, st, map, kpm, is_pad, B, Q);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_bcast_add_rows(float* dst, const float* vec, int R, int D, hipStream_t st) {
    const int64_t total = (int64_t)R * D;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(bcast_add_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dst, vec, D, total);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// weights for the f16x3 GEMM: every aligned group of 4 floats -> {4 hi halfs, 4 lo halfs} in the same 16 bytes.
// flag (optional): bit 1 is raised when a scaled value leaves the fp16 range or is not finite (the image would hold inf)
__global__ void split16_kernel(const actmi_f32x4* __restrict__ src, uint4* __restrict__ dst, int64_t n4, float scale,
                               uint32_t* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const actmi_f32x4 v = src[i] * scale;
    uint2 hi, lo;
    split16(v, hi, lo);
    dst[i] = uint4{hi.x, hi.y, lo.x, lo.y};
    if (flag && !(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))) < 65504.f)) atomicOr(flag, 2u);
}

int launch_split16(const float* src, float* dst, int64_t nfloats, float scale, hipStream_t st, uint32_t* flag) {
    if (nfloats <= 0) return 0;
    if ((nfloats & 3) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return -2;
    const int64_t n4 = nfloats / 4;
    hipLaunchKernelGGL(split16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const actmi_f32x4*>(src), reinterpret_cast<uint4*>(dst), n4, scale, flag);
    return (int)hipGetLastError();
}

// the parameter arena: 64-float slots, slot g belongs to segment (parameter) seg_of_group64[g] whose scale is seg_scale[.]
__global__ void split16_map_kernel(const actmi_f32x4* __restrict__ src, uint4* __restrict__ dst, int64_t n4,
                                   const int* __restrict__ seg_of_group64, const float* __restrict__ seg_scale,
                                   uint32_t* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float scale = seg_scale[seg_of_group64[i >> 4]];
    const actmi_f32x4 v = src[i] * scale;
    uint2 hi, lo;
    split16(v, hi, lo);
    dst[i] = uint4{hi.x, hi.y, lo.x, lo.y};
    if (flag && !(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))) < 65504.f)) atomicOr(flag, 2u);
}

int launch_split16_map(const float* src, float* dst, int64_t nfloats, const int* seg_of_group64, const float* seg_scale,
                       uint32_t* flag, hipStream_t st) {
    if (nfloats <= 0) return 0;
    if ((nfloats & 63) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return -2;
    const int64_t n4 = nfloats / 4;
    hipLaunchKernelGGL(split16_map_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const actmi_f32x4*>(src), reinterpret_cast<uint4*>(dst), n4, seg_of_group64, seg_scale, flag);
    return (int)hipGetLastError();
}

// out_bits[s] = bits of max |x| over segment s (non-negative floats order like unsigned); NaN / inf bits order above
// every finite value, so a non-finite parameter shows up as a huge "amax"
__global__ void seg_amax_kernel(const float* __restrict__ base, const int64_t* __restrict__ off, const int64_t* __restrict__ numel,
                                unsigned* __restrict__ out_bits) {
    const int sgm = blockIdx.x;
    const float* x = base + off[sgm];
    const int64_t n = numel[sgm];
    unsigned m = 0;
    for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.y * blockDim.x)
        m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out_bits + sgm, m);
}

int launch_seg_amax(const float* base, const int64_t* off, const int64_t* numel, int nseg, unsigned* out_bits, hipStream_t st) {
    if (nseg <= 0) return 0;
    if (hipMemsetAsync(out_bits, 0, (size_t)nseg * sizeof(unsigned), st) != hipSuccess) return -3;
    hipLaunchKernelGGL(seg_amax_kernel, dim3(nseg, 16), dim3(256), 0, st, base, off, numel, out_bits);
    return (int)hipGetLastError();
}

// raises `bit` in *flag when any of x[0..n) is NaN or infinite (the default-on output check of the forward passes)
__global__ void check_finite_kernel(const float* __restrict__ x, int64_t n, uint32_t* __restrict__ flag, uint32_t bit) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        bad = bad || !(fabsf(x[i]) <= 3.402823466e38f);
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, bit);
}

int launch_check_finite(const float* x, int64_t n, uint32_t* flag, uint32_t bit, hipStream_t st) {
    if (n <= 0 || !flag) return 0;
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(check_finite_kernel, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(256), 0, st, x, n, flag, bit);
    return (int)hipGetLastError();
}


__global__ void scale_kernel(float* __restrict__ x, int64_t n4, int64_t n, float s) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) {
        actmi_f32x4* p = reinterpret_cast<actmi_f32x4*>(x) + i;
        *p = *p * s;
    } else if (i == n4) {
        for (int64_t j = n4 * 4; j < n; ++j) x[j] *= s;
    }
}

// max |x| over an M x N matrix (row stride ld) as the bits of a non-negative float (monotone as unsigned), then the
// power-of-two scale derived from it
__global__ void amax_kernel(const float* __restrict__ x, int64_t ld, int M, int N, unsigned* __restrict__ bits) {
    const int64_t total = (int64_t)M * N;
    unsigned m = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / N, c = i - r * N;
        m = max(m, __float_as_uint(x[r * ld + c]) & 0x7fffffffu);
    }
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) amax_commit(bits, m);
}

// the same over 16-byte pieces (N, ld multiples of 4, 16-byte aligned base): four independent loads in flight per thread,
// 32-bit index arithmetic (host: M * N / 4 < 2^31); FLAT: rows are contiguous (ld == N), no row / column split at all
template <int FLAT>
__global__ __launch_bounds__(256) void amax_vec_kernel(const actmi_f32x4* __restrict__ x, unsigned ld4, unsigned nv, unsigned total,
                                                       unsigned* __restrict__ bits) {
    const unsigned step = gridDim.x * 256u;
    unsigned m = 0;
    auto at = [&](unsigned i) -> actmi_f32x4 {
        if (FLAT) return x[i];
        const unsigned r = i / nv, c = i - r * nv;
        return x[(uint64_t)r * ld4 + c];
    };
    auto acc = [&](const actmi_f32x4& v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) m = max(m, __float_as_uint(v[e]) & 0x7fffffffu);
    };
    unsigned i = blockIdx.x * 256u + threadIdx.x;
    for (; i + 3u * step < total && i + 3u * step >= i; i += 4u * step) {
        const actmi_f32x4 v0 = at(i), v1 = at(i + step), v2 = at(i + 2u * step), v3 = at(i + 3u * step);
        acc(v0); acc(v1); acc(v2); acc(v3);
    }
    for (; i < total; i += step) {
        acc(at(i));
        if (i + step < i) break;                           // 32-bit wrap (total close to 2^32)
    }
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) amax_commit(bits, m);
}

__global__ void pow2_scale_kernel(unsigned* __restrict__ bits, float* __restrict__ out) {
    const unsigned b = *bits;
    const int e = (int)(b >> 23);                      // biased exponent of max|x|
    float s = 1.f;
    if (b != 0 && e != 255) s = ldexpf(1.f, 13 - (e - 127));     // 2^e' <= max < 2^(e'+1)  ->  max * s in [2^13, 2^14)
    *out = s;
    *bits = 0;                                         // re-arm the slot
}

int launch_amax_bits(const float* x, int64_t ld, int M, int N, unsigned* bits, hipStream_t st) {
    const int64_t total = (int64_t)M * N;
    if (total <= 0) return 0;
    const bool vec = (N & 3) == 0 && (ld & 3) == 0 && ((uintptr_t)x & 15) == 0 && total / 4 < ((int64_t)1 << 31) &&
                     (ld == N || ld / 4 < ((int64_t)1 << 31));
    if (vec) {
        const unsigned tv = (unsigned)(total / 4);
        unsigned blocks = (tv + 256u * 4u - 1) / (256u * 4u);
        if (blocks < 1) blocks = 1;
        if (blocks > 4096) blocks = 4096;
        const actmi_f32x4* xv = reinterpret_cast<const actmi_f32x4*>(x);
        if (ld == N) hipLaunchKernelGGL(amax_vec_kernel<1>, dim3(blocks), dim3(256), 0, st, xv, 0u, (unsigned)(N / 4), tv, bits);
        else hipLaunchKernelGGL(amax_vec_kernel<0>, dim3(blocks), dim3(256), 0, st, xv, (unsigned)(ld / 4), (unsigned)(N / 4), tv, bits);
    } else {
        int blocks = (int)((total + 256 * 16 - 1) / (256 * 16));
        if (blocks < 1) blocks = 1;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(amax_kernel, dim3(blocks), dim3(256), 0, st, x, ld, M, N, bits);
    }
    return (int)hipGetLastError();
}

// scale from bits that a producing kernel already accumulated (its own atomicMax of |value| bits into out + 1)
int launch_pow2_from_bits(float* out, hipStream_t st) {
    hipLaunchKernelGGL(pow2_scale_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<unsigned*>(out + 1), out);
    return (int)hipGetLastError();
}

int launch_pow2_scale(const float* x, int64_t ld, int M, int N, float* out, hipStream_t st) {
    // the word after the scale holds the running max bits (zero between uses)
    unsigned* bits = reinterpret_cast<unsigned*>(out + 1);
    const int rc = launch_amax_bits(x, ld, M, N, bits, st);
    if (rc != 0) return rc;
    return launch_pow2_from_bits(out, st);
}

namespace {
template <int V>
__global__ __launch_bounds__(256) void splitk_combine_kernel(SplitCombineArgs a) {
    const int nv = a.N / V;
    const int64_t per_group = (int64_t)a.M * nv;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_group * a.groups) return;
    const int g = (int)(i / per_group);
    const int64_t r = i - (int64_t)g * per_group;
    const int m = (int)(r / nv), n = (int)(r - (int64_t)m * nv) * V;
    const float* src = a.part + g * a.gP + (int64_t)m * a.ldp + n;
    float v[V];
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = 0.f;
    // slices in groups of 8: all loads of a group in flight together, summed in index order (run-to-run identical);
    // adding the zero of an absent slice changes nothing
    for (int s0 = 0; s0 < a.nsplit; s0 += 8) {
        float tv[8][V];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool on = s0 + u < a.nsplit;
            const float* q = src + (int64_t)(on ? s0 + u : s0) * a.split_stride;
            if (V == 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(q);
#pragma unroll
                for (int e = 0; e < V; ++e) tv[u][e] = on ? t[e] : 0.f;
            } else tv[u][0] = on ? q[0] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < V; ++e) v[e] += tv[u][e];
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const float sc = a.scale ? a.scale[g * a.gSB + n + e] : 1.f;
        const float bi = a.bias ? a.bias[g * a.gSB + n + e] : 0.f;
        float x = v[e] * sc + bi;
        if (a.res) x += a.res[g * a.gRes + (int64_t)m * a.ldres + n + e];
        if (a.relu == 1) x = fmaxf(x, 0.f);
        else if (a.relu == 2) x = 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
        v[e] = x;
    }
    float* dst = a.C + g * a.gC + (int64_t)m * a.ldc + n;
    if (V == 4) *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
    else dst[0] = v[0];
}
}  // namespace

int launch_splitk_combine(const SplitCombineArgs& a, hipStream_t st) {
    if (a.M <= 0 || a.N <= 0 || a.groups <= 0) return 0;
    const bool v4 = (a.N & 3) == 0 && (a.ldp & 3) == 0 && (a.ldc & 3) == 0 && (a.split_stride & 3) == 0 && (a.gP & 3) == 0 &&
                    (a.gC & 3) == 0 && ((uintptr_t)a.part & 15) == 0 && ((uintptr_t)a.C & 15) == 0;
    const int64_t total = (int64_t)a.groups * a.M * (v4 ? a.N / 4 : a.N);
    if (prof_enabled())
        prof_begin("splitk_combine_kernel", 0.0, 4.0 * a.groups * (double)a.M * a.N * (a.nsplit + 1 + (a.res ? 1 : 0)), st);
    if (v4) hipLaunchKernelGGL(splitk_combine_kernel<4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(splitk_combine_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
    prof_end(st);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_scale(float* x, int64_t n, float s, hipStream_t st) {
    if (n <= 0) return 0;
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n4 + 1 + 255) / 256)), dim3(256), 0, st, x, n4, n, s);
    return (int)hipGetLastError();
}

int launch_axpy(float* dst, const float* src, int64_t n, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(axpy_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dst, src, n);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_bn_fold(const float* w, const float* b, const float* rm, const float* rv, float* scale, float* bias, int n,
                   hipStream_t st) {
    hipLaunchKernelGGL(bn_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w, b, rm, rv, scale, bias, n);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// Weight matrix of a ResNet block's second convolution with the block's downsample branch riding in the same contraction
// (gemm.hip, second source): out[g][n][:] = [ s2[g][n] * w2[g][n][0..K2) | sd[g][n] * wd[g][n][0..Kd) ], bias[g][n] = b2 + bd.
// The FrozenBN scales are folded into the weights because the two convolutions share one accumulator (reference:
// out = relu(bn2(conv2(y1)) + bn_ds(conv_ds(x))), torchvision BasicBlock as used at backbone.py:66-71).
__global__ void fold_cat_w_kernel(const float* __restrict__ w2, const float* __restrict__ s2, const float* __restrict__ b2,
                                  const float* __restrict__ wd, const float* __restrict__ sd, const float* __restrict__ bd,
                                  float* __restrict__ out, float* __restrict__ bias, int N, int K2, int Kd, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int Kf = K2 + Kd;
    const int k = (int)(idx % Kf);
    const int64_t gn = idx / Kf;                  // g * N + n
    out[idx] = k < K2 ? w2[gn * K2 + k] * s2[gn] : wd[gn * Kd + (k - K2)] * sd[gn];
    if (k == 0) bias[gn] = b2[gn] + bd[gn];
}

int launch_fold_cat_w(const float* w2, const float* s2, const float* b2, const float* wd, const float* sd, const float* bd,
                      float* out, float* bias, int G, int N, int K2, int Kd, hipStream_t st) {
    const int64_t total = (int64_t)G * N * (K2 + Kd);
    hipLaunchKernelGGL(fold_cat_w_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w2, s2, b2, wd, sd, bd, out, bias,
                       N, K2, Kd, total);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// K order of a convolution weight matrix for L2 reuse of the input patch (gemm.hip, k_tap_inner): (tap, c) -> (c / 32, tap, c % 32)
__global__ void permute_conv_k_kernel(const float* __restrict__ src, float* __restrict__ dst, int taps, int cin, int ld, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int k = (int)(idx % ld);
    const int64_t row = idx / ld;
    int ks = k;                                   // source column of destination column k
    if (k < taps * cin) {
        const int cb = k / (taps * 32), rem = k - cb * taps * 32;
        const int tap = rem / 32, ci = rem - tap * 32;
        ks = tap * cin + cb * 32 + ci;
    }
    dst[idx] = src[row * ld + ks];
}

int launch_permute_conv_k(const float* src, float* dst, int64_t rows, int taps, int cin, int ld, hipStream_t st) {
    if (rows <= 0) return 0;
    if ((cin & 31) || taps < 1 || ld < taps * cin || src == dst) return -2;
    const int64_t total = rows * ld;
    hipLaunchKernelGGL(permute_conv_k_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, dst, taps, cin, ld, total);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_build_rowmap(int* map, int B, int C, int fh, int fw, int N, hipStream_t st) {
    const int total = C * B * fh * fw;
    hipLaunchKernelGGL(build_rowmap_kernel, dim3((total + 255) / 256), dim3(256), 0, st, map, B, C, fh, fw, N);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_small_linear(const float* x, int64_t ldx, const float* w, const float* b, float* y, int64_t ldy, int M,
                        int N, int K, hipStream_t st, float* fill_dst, const float* fill_src, int64_t fill_src_bs) {
    const int64_t total = (int64_t)M * N;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(small_linear_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, ldx, w, b, y,
                       ldy, M, N, K, fill_dst, fill_src, fill_src_bs);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_fill_rows(float* dst, int64_t ld, int64_t batch_stride, const float* src, int64_t src_bs, int B, int D,
                     hipStream_t st) {
    (void)ld;
    const int64_t total = (int64_t)B * D;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dst, batch_stride,
                       src, src_bs, B, D);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_repack_conv_w(const float* w_oihw, float* w_ohwi, int G, int O, int I, int KH, int KW, int64_t g_in,
                         int64_t g_out, int kpad, hipStream_t st) {
    const int64_t total = (int64_t)G * O * kpad;
    hipLaunchKernelGGL(repack_conv_w_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w_oihw, w_ohwi,
                       O, I, KH, KW, g_in, g_out, kpad, total);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_ensemble(float* ring, int* tcount, const float* chunk, double k, double* out, uint8_t* populated, int E,
                    int Q, int A, hipStream_t st) {
    if (E <= 0) return 0;
    if (A > 64) return -2;
    hipLaunchKernelGGL(ensemble_kernel, dim3(E), dim3(256), Q * sizeof(double), st, ring, tcount, chunk, k, out, populated, Q, A);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
