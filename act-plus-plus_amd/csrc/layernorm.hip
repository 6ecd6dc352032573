// LayerNorm over the last dimension with fused residual add, one 64-lane wave per row, wave-shuffle
// reductions (reference: nn.LayerNorm eps 1e-5 in transformer.py:200-201,261-263,35; the residual adds of
// forward_post transformer.py:219-223, 284-294).  Optional second LayerNorm on top (decoder.norm applied to the
// layer output, transformer.py:175).  Rows are held in registers (D <= 2048), two-pass mean / variance.
#include "common.h"

namespace {
constexpr int MAXV = 8;   // float4 per lane -> D <= 64*4*8 = 2048

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// EXTRA: the consumers' work that only needs the finished row (LnExtra in common.h):
//   y2[row] = y[row] + add2[row % add2_mod]   -- the q = k = x + pos operand of the NEXT attention block (transformer.py:216),
//                                                so that its packed QKV product is a plain GEMM (no addend in the operand loader)
//   head_out[row][n] = y[row] . head_w[n] + head_b[n], n < head_n -- the action head (detr_vae.py:252) on the row that is still in
//                                                registers (a 13-workgroup GEMM launch of 21.7 us otherwise), with the
//                                                output finiteness check of the range guard
template <int NV, bool EXTRA>      // float4 per lane actually needed: ceil(D / 256); the loops carry no dead iterations
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                        int res_mod, const float* __restrict__ w,
                                                        const float* __restrict__ b, const float* __restrict__ w2,
                                                        const float* __restrict__ b2, float* __restrict__ y, int M,
                                                        int D, float eps, int nsplit, int64_t split_stride,
                                                        const float* __restrict__ bias, LnExtra ex) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int D4 = D >> 2;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (int64_t)row * D);
    const f32x4* rr = res ? reinterpret_cast<const f32x4*>(res + (int64_t)(res_mod ? row % res_mod : row) * D) : nullptr;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < D4) {
            f32x4 t = xr[c];
            // x given as nsplit split-K slices (ctx_gemm): summed in slice order, then + bias, then + res -- the sequence of
            // splitk_combine_kernel, so the fused form gives the same bits as combine + layernorm
            for (int sidx = 1; sidx < nsplit; ++sidx) t += xr[(int64_t)sidx * (split_stride >> 2) + c];
            if (bias) t += reinterpret_cast<const f32x4*>(bias)[c];
            if (rr) t += rr[c];
            v[i] = t;
            s += (t[0] + t[1]) + (t[2] + t[3]);
        }
    }
    const float invD = 1.f / (float)D;
    float mean = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (lane + 64 * i < D4) {
            const f32x4 d = v[i] - mean;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
    float rstd = 1.f / sqrtf(wave_sum(q) * invD + eps);
    const f32x4* w4 = reinterpret_cast<const f32x4*>(w);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < D4) v[i] = (v[i] - mean) * rstd * w4[c] + b4[c];
    }
    if (w2) {
        s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < D4) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        mean = wave_sum(s) * invD;
        q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (lane + 64 * i < D4) {
                const f32x4 d = v[i] - mean;
                q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
        }
        rstd = 1.f / sqrtf(wave_sum(q) * invD + eps);
        const f32x4* w24 = reinterpret_cast<const f32x4*>(w2);
        const f32x4* b24 = reinterpret_cast<const f32x4*>(b2);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < D4) v[i] = (v[i] - mean) * rstd * w24[c] + b24[c];
        }
    }
    f32x4* yr = reinterpret_cast<f32x4*>(y + (int64_t)row * D);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < D4) yr[c] = v[i];
    }
    if (EXTRA) {
        if (ex.y2) {
            const f32x4* ar = reinterpret_cast<const f32x4*>(ex.add2 + (int64_t)(ex.add2_mod ? row % ex.add2_mod : row) * D);
            f32x4* y2r = reinterpret_cast<f32x4*>(ex.y2 + (int64_t)row * D);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + 64 * i;
                if (c < D4) y2r[c] = v[i] + ar[c];
            }
        }
        if (ex.head_out) {
            bool bad = false;
            for (int n = 0; n < ex.head_n; ++n) {
                const f32x4* hw = reinterpret_cast<const f32x4*>(ex.head_w + (int64_t)n * D);
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int c = lane + 64 * i;
                    if (c < D4) {
                        const f32x4 t = hw[c];
                        acc = fmaf(v[i][0], t[0], acc); acc = fmaf(v[i][1], t[1], acc);
                        acc = fmaf(v[i][2], t[2], acc); acc = fmaf(v[i][3], t[3], acc);
                    }
                }
                acc = wave_sum(acc) + (ex.head_b ? ex.head_b[n] : 0.f);
                if (lane == 0) ex.head_out[(int64_t)row * ex.head_n + n] = acc;
                bad = bad || !(fabsf(acc) <= 3.402823466e38f);
            }
            if (ex.flag && bad && lane == 0) atomicOr(ex.flag, ex.flag_bit);
        }
    }
}
}  // namespace

int launch_layernorm(const float* x, const float* res, int res_mod, const float* w, const float* b, const float* w2,
                     const float* b2, float* y, int M, int D, float eps, hipStream_t st, std::string* err, int nsplit,
                     int64_t split_stride, const float* bias, const LnExtra* extra) {
    if ((D & 3) || D > 64 * 4 * MAXV) { if (err) *err = "layernorm: D must be a multiple of 4 and <= 2048"; return -2; }
    if (nsplit > 1 && (split_stride & 3)) { if (err) *err = "layernorm: slice stride must be a multiple of 4"; return -2; }
    if (M <= 0) return 0;
    const int nv = (D / 4 + 63) / 64;
    const bool has_extra = extra && (extra->y2 || extra->head_out);
    if (has_extra && extra->y2 && (!extra->add2 || ((uintptr_t)extra->y2 & 15) || ((uintptr_t)extra->add2 & 15))) { if (err) *err = "layernorm: bad second output"; return -2; }
    if (has_extra && extra->head_out && (!extra->head_w || extra->head_n < 1 || ((uintptr_t)extra->head_w & 15))) { if (err) *err = "layernorm: bad head"; return -2; }
    const LnExtra ex = has_extra ? *extra : LnExtra{};
    prof_begin("layernorm_kernel", 0.0, 4.0 * M * D * (res && !res_mod ? 3.0 : 2.0), st);
#define ACTMI_LN(NV) do { if (has_extra) hipLaunchKernelGGL((layernorm_kernel<NV, true>), dim3((M + 3) / 4), dim3(256), 0, st, x, res, res_mod, w, b, w2, b2, y, M, D, eps, nsplit, split_stride, bias, ex); \
                          else hipLaunchKernelGGL((layernorm_kernel<NV, false>), dim3((M + 3) / 4), dim3(256), 0, st, x, res, res_mod, w, b, w2, b2, y, M, D, eps, nsplit, split_stride, bias, ex); } while (0)
    switch (nv) {
        case 1: ACTMI_LN(1); break;
        case 2: ACTMI_LN(2); break;
        case 3: ACTMI_LN(3); break;
        case 4: ACTMI_LN(4); break;
        default: ACTMI_LN(MAXV); break;
    }
#undef ACTMI_LN
    prof_end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { if (err) *err = std::string("layernorm launch: ") + hipGetErrorString(e); return -3; }
    return 0;
}
