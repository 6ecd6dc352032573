// 3x3 / stride 2 / pad 1 max pooling on NHWC maps (torchvision resnet ``maxpool``).  HBM-bound:
// each thread produces 4 channels (one float4) of one output pixel; consecutive lanes walk the channel
// dimension first so that every load/store instruction covers whole 16-byte-per-lane contiguous runs.
#include "common.h"

namespace {
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                      int H, int W, int C4, int Ho, int Wo, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        int64_t pix = idx / C4;
        const int wo = (int)(pix % Wo); pix /= Wo;
        const int ho = (int)(pix % Ho);
        const int64_t img = pix / Ho;
        const f32x4* src = reinterpret_cast<const f32x4*>(in) + img * H * W * C4 + c4;
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = 2 * ho - 1 + r;
            if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int wi = 2 * wo - 1 + s;
                if ((unsigned)wi >= (unsigned)W) continue;
                const f32x4 v = src[((int64_t)hi * W + wi) * C4];
                m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
            }
        }
        reinterpret_cast<f32x4*>(out)[idx] = m;
    }
}
// horizontal half of the pool: rows are independent; each thread produces 4 channels of one output pixel
__global__ __launch_bounds__(256) void hpool_kernel(const float* __restrict__ in, float* __restrict__ out, int W, int C4,
                                                    int Wo, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        const int64_t pix = idx / C4;
        const int wo = (int)(pix % Wo);
        const int64_t row = pix / Wo;
        const f32x4* src = reinterpret_cast<const f32x4*>(in) + row * W * C4 + c4;
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int wi = 2 * wo - 1 + s;
            if ((unsigned)wi >= (unsigned)W) continue;
            const f32x4 v = src[(int64_t)wi * C4];
            m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
        }
        reinterpret_cast<f32x4*>(out)[idx] = m;
    }
}
// training variant: also records, per output element, WHICH of the 9 window positions (r*3+s) held the maximum (the
// first one in scan order, ATen's tie rule) so that the backward pass is a gather of at most 4 (byte, float) pairs per
// input element instead of recomputing four 9-element maxima
__global__ __launch_bounds__(256) void maxpool_idx_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          uint8_t* __restrict__ arg, int H, int W, int C4, int Ho, int Wo,
                                                          unsigned total) {
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const unsigned pix0 = idx / (unsigned)C4;
        const int c4 = (int)(idx - pix0 * (unsigned)C4);
        const unsigned row = pix0 / (unsigned)Wo;
        const int wo = (int)(pix0 - row * (unsigned)Wo);
        const unsigned img = row / (unsigned)Ho;
        const int ho = (int)(row - img * (unsigned)Ho);
        const f32x4* src = reinterpret_cast<const f32x4*>(in) + (int64_t)img * H * W * C4 + c4;
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int code[4] = {255, 255, 255, 255};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = 2 * ho - 1 + r;
            if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int wi = 2 * wo - 1 + s;
                if ((unsigned)wi >= (unsigned)W) continue;
                const f32x4 v = src[((int64_t)hi * W + wi) * C4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[e] > m[e] || code[e] == 255) { m[e] = v[e]; code[e] = r * 3 + s; }
            }
        }
        reinterpret_cast<f32x4*>(out)[idx] = m;
        reinterpret_cast<uchar4*>(arg)[idx] = make_uchar4((uint8_t)code[0], (uint8_t)code[1], (uint8_t)code[2], (uint8_t)code[3]);
    }
}
}  // namespace

int launch_maxpool_idx(const float* in, float* out, uint8_t* arg, int nimg, int H, int W, int C, int Ho, int Wo, hipStream_t st) {
    if (C & 3) return -2;
    const int64_t total = (int64_t)nimg * Ho * Wo * (C / 4);
    if (total >= ((int64_t)1 << 31)) return -2;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    prof_begin("maxpool_idx_kernel", 0.0, 4.0 * nimg * C * ((double)H * W + 1.25 * Ho * Wo), st);
    hipLaunchKernelGGL(maxpool_idx_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, out, arg, H, W, C / 4, Ho, Wo, (unsigned)total);
    prof_end(st);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_maxpool(const float* in, float* out, int nimg, int H, int W, int C, int Ho, int Wo, hipStream_t st) {
    if (C & 3) return -2;
    const int64_t total = (int64_t)nimg * Ho * Wo * (C / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    prof_begin("maxpool_kernel", 0.0, 4.0 * nimg * C * ((double)H * W + (double)Ho * Wo), st);
    hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, out, H, W, C / 4, Ho, Wo, total);
    prof_end(st);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_hpool(const float* in, float* out, int nrows, int W, int C, int Wo, hipStream_t st) {
    if (C & 3) return -2;
    const int64_t total = (int64_t)nrows * Wo * (C / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) return 0;
    prof_begin("hpool_kernel", 0.0, 4.0 * nrows * C * ((double)W + (double)Wo), st);
    hipLaunchKernelGGL(hpool_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, out, W, C / 4, Wo, total);
    prof_end(st);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
