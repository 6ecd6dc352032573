// ResNet stem: conv 7x7 / stride 2 / pad 3 (3 -> Cout<=64) + FrozenBatchNorm2d + ReLU, NHWC output.
// Fuses the image contract of the reference into the loader: u8/255 (imitate_episodes.py:212, utils.py:152)
// and the ImageNet normalisation (policy.py:268-272) come from a 3x256 lookup table built on the host
// with exactly the reference's float arithmetic; the f32 NCHW input form (the ACTPolicy.__call__ signature)
// is normalised in the loader with the same (x - mean) / std.  FrozenBN = per-channel scale/bias epilogue
// (backbone.py:47-57).  Zero padding applies to the NORMALISED image, as in the reference.
//
// Work decomposition: one block per (camera, run of tiles); a tile is 64 consecutive output pixels of one
// output row x all Cout channels.  The 7 x 133 x 3 input patch of a tile is staged once in LDS as f32; each of
// the 4 waves owns one 32x32 MFMA tile (v_mfma_f32_32x32x2_f32) and walks K = 147 (+1 zero pad) in 74 steps,
// reading its A operand straight from the patch (im2col on the fly) and B from an LDS copy of the
// camera's weights (row stride 149 floats: conflict-free).  Blocks are persistent over tiles so the 37 KB
// weight image is staged once per block.
#include "common.h"

namespace {

constexpr int KREAL = 147, KPAD = 148, WSTRIDE = 149;
constexpr int TILE_P = 64;                 // output pixels per tile
constexpr int PCOLS = 2 * TILE_P + 5;      // 133 input columns
constexpr int PSTRIDE = 400;               // floats per patch row (133*3 = 399, padded)
constexpr int PROWS = 8;                   // 7 real rows + 1 zero row for the K pad

__host__ __device__ constexpr int koff(int k) { return (k / 21) * PSTRIDE + (k % 21); }

template <int FMT>
__global__ __launch_bounds__(256) void conv1_kernel(Conv1Args p, int tiles_per_row, int tiles_per_cam) {
    __shared__ __attribute__((aligned(16))) float s_w[64 * WSTRIDE];
    __shared__ __attribute__((aligned(16))) float s_patch[PROWS * PSTRIDE];
    __shared__ float s_lut[3 * 256];
    const int cam = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int mtile = wave & 1, ntile = wave >> 1;

    // weights of this camera: [Cout][KPAD] -> LDS [64][149]; rows >= Cout are zero
    const float* wg = p.w + (int64_t)cam * p.Cout * KPAD;
    for (int e = t; e < 64 * KPAD; e += 256) {
        const int n = e / KPAD, k = e - n * KPAD;
        s_w[n * WSTRIDE + k] = (n < p.Cout) ? wg[n * KPAD + k] : 0.f;
    }
    for (int e = t; e < 3 * 256; e += 256) s_lut[e] = (FMT == 0) ? p.lut[e] : 0.f;
    for (int e = t; e < PROWS * PSTRIDE; e += 256) s_patch[e] = 0.f;
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float stdv[3] = {0.229f, 0.224f, 0.225f};
    __syncthreads();

    const bool active = ntile * 32 < p.Cout;
    const int n = ntile * 32 + li;
    const float sc = (n < p.Cout) ? p.scale[cam * p.Cout + n] : 0.f;
    const float bi = (n < p.Cout) ? p.bias[cam * p.Cout + n] : 0.f;
    const float* a_base = s_patch + (mtile * 32 + li) * 6;
    const float* b_base = s_w + n * WSTRIDE + lh;

    for (int tile = blockIdx.x; tile < tiles_per_cam; tile += gridDim.x) {
        const int b = tile / (p.Ho * tiles_per_row);
        const int rem = tile - b * (p.Ho * tiles_per_row);
        const int ho = rem / tiles_per_row;
        const int wo0 = (rem - ho * tiles_per_row) * TILE_P;
        const int64_t img = (int64_t)b * p.C + cam;      // input image index ([B][C] layout of the caller)
        const int64_t oimg = (int64_t)cam * p.B + b;     // output is camera-major: each camera is one GEMM group
        // ---- stage the patch (7 rows x 133 cols x 3 ch), zero outside the image
        const int hi0 = 2 * ho - 3, wi0 = 2 * wo0 - 3;
        if (FMT == 0) {
            const uint8_t* src = reinterpret_cast<const uint8_t*>(p.image) + img * p.H * p.W * 3;
            for (int e = t; e < 7 * PCOLS * 3; e += 256) {
                const int r = e / (PCOLS * 3), x = e - r * (PCOLS * 3);
                const int pc = x / 3, c = x - pc * 3;
                const int hi = hi0 + r, wi = wi0 + pc;
                float v = 0.f;
                if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                    v = s_lut[c * 256 + src[((int64_t)hi * p.W + wi) * 3 + c]];
                s_patch[r * PSTRIDE + x] = v;
            }
        } else {
            const float* src = reinterpret_cast<const float*>(p.image) + img * 3 * p.H * p.W;
            for (int e = t; e < 7 * 3 * PCOLS; e += 256) {
                const int rc = e / PCOLS, pc = e - rc * PCOLS;
                const int r = rc / 3, c = rc - r * 3;
                const int hi = hi0 + r, wi = wi0 + pc;
                float v = 0.f;
                if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                    v = (src[((int64_t)c * p.H + hi) * p.W + wi] - mean[c]) / stdv[c];
                s_patch[r * PSTRIDE + pc * 3 + c] = v;
            }
        }
        __syncthreads();
        if (active) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int s = 0; s < KPAD / 2; ++s) {
                const int o0 = koff(2 * s), o1 = koff(2 * s + 1);
                const float av = a_base[lh ? o1 : o0];
                const float bv = b_base[2 * s];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            }
            if (n < p.Cout) {
                float* orow = p.out + ((oimg * p.Ho + ho) * (int64_t)p.Wo) * p.Cout + n;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int wo = wo0 + mtile * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (wo < p.Wo) orow[(int64_t)wo * p.Cout] = fmaxf(acc[e] * sc + bi, 0.f);
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace

int launch_conv1(const Conv1Args& a, hipStream_t st, std::string* err) {
    if (a.Cout > 64 || a.Cout < 1) { if (err) *err = "conv1: Cout must be in 1..64"; return -2; }
    if (a.Ho != (a.H + 6 - 7) / 2 + 1 || a.Wo != (a.W + 6 - 7) / 2 + 1) { if (err) *err = "conv1: bad output size"; return -2; }
    const int tiles_per_row = (a.Wo + TILE_P - 1) / TILE_P;
    const int tiles_per_cam = a.B * a.Ho * tiles_per_row;
    int gx = tiles_per_cam < 512 ? tiles_per_cam : 512;
    // keep tiles-per-block balanced
    const int per = (tiles_per_cam + gx - 1) / gx;
    gx = (tiles_per_cam + per - 1) / per;
    dim3 grid(gx, a.C);
    prof_begin(a.fmt == 0 ? "conv1_kernel<0>" : "conv1_kernel<1>", 2.0 * a.B * a.C * a.Ho * a.Wo * a.Cout * 147.0,
               (double)a.B * a.C * ((double)a.H * a.W * 3 * (a.fmt == 0 ? 1 : 4) + 4.0 * a.Ho * a.Wo * a.Cout), st);
    if (a.fmt == 0) hipLaunchKernelGGL(conv1_kernel<0>, grid, dim3(256), 0, st, a, tiles_per_row, tiles_per_cam);
    else hipLaunchKernelGGL(conv1_kernel<1>, grid, dim3(256), 0, st, a, tiles_per_row, tiles_per_cam);
    prof_end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { if (err) *err = std::string("conv1 launch: ") + hipGetErrorString(e); return -3; }
    return 0;
}
