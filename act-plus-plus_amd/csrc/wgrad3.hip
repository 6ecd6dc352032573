// wgrad3: weight gradient of the 3x3 / stride 1 / pad 1 convolutions with 64 input and 64 output channels (ResNet18 layer1,
// torchvision BasicBlock; backward of backbone.py:95-134's body for the layer1 blocks) as a DIRECT kernel, f16x3 arithmetic.
//
//   dW[co][(r,s,ci)] = sum over pixels p of  dY[p][co] * X[p + (r-1, s-1)][ci]
//
// As a GEMM (gemm.hip, AMODE 3 / BMODE 2) this is M = 64, N = 576, K = pixels: nine 64x64 tiles that each re-read dY and
// gather X once per tap, 6 MFMAs per wave between two loader passes -- 60 TFLOP/s.  Here a workgroup walks DOWN a strip of
// 32 pixel columns of one image and keeps all nine taps' 64x64 results in registers (wave = a 32x32 (co, ci) quadrant x 9
// taps = 144 accumulator registers): per row of the strip it stages dY^T [co][32 px] and the new input row X^T [ci][32 px]
// (three copies, shifted by s = 0, 1, 2 pixels, so that every tap reads aligned 16-byte fragments) ONCE and issues
// 54 MFMAs per wave on them; the two older input rows stay in an LDS ring of four row slots.  The contraction index of the
// matrix instruction is the pixel, so both operands are staged TRANSPOSED (pixels contiguous): each thread loads a block of
// 4 pixels x 4 channels as four 16-byte loads and writes, per channel, the 4 pixels' hi / lo halfs as 8-byte LDS stores.
// Workgroups are persistent (grid = workgroups per camera x cameras), accumulate over all their (image, strip) units and
// write ONE partial each; launch_splitk_combine adds the partials to dW in a fixed order (bitwise repeatable).
#include "common.h"
#include "split16.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int CH = 64;                 // channels in and out
constexpr int SW = 32;                 // pixels per strip row = contraction depth of one step (2 MFMA k-steps)
constexpr int TILE_B = CH * 128;       // one transposed tile [64 rows][32 px hi | 32 px lo]: 8 KB
constexpr int SLOT_B = 3 * TILE_B;     // one input row: three shifted copies
constexpr int NSLOT = 4;
constexpr int DY_OFF = NSLOT * SLOT_B;             // two dY^T buffers behind the ring
constexpr int SMEM_B = DY_OFF + 2 * TILE_B;        // 112 KB

struct Wgrad3Args {
    const float* dy;          // [G][B][H][W][64]
    const float* x;           // [G][B][H][W][64]
    float* part;              // [G][nwg][64][576]
    const float* dy_scale;    // device: power-of-two scale of dY (or NULL = 1)
    int B, H, W, nwg;
};

// byte offset inside a tile of (row, logical 16-byte chunk): chunks 0-3 = hi halfs of pixels 8c..8c+7, 4-7 = lo halfs
__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__global__ __launch_bounds__(256) void wgrad3x3_c64_kernel(Wgrad3Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, kg = lane >> 5;
    const int co0 = (wave >> 1) * 32, ci0 = (wave & 1) * 32;
    const int g = blockIdx.y, wg = blockIdx.x;
    const int strips = (p.W + SW - 1) / SW;
    const int units = p.B * strips;
    const int64_t img = (int64_t)p.H * p.W * CH;
    const float* dy_g = p.dy + (int64_t)g * p.B * img;
    const float* x_g = p.x + (int64_t)g * p.B * img;
    const float sc = p.dy_scale ? *p.dy_scale : 1.f;

    // staging role of this thread: two 4 px x 4 ch blocks per step; block b = t + 256 k: tile = b >> 7 (0-2: input row
    // shifted by s = tile, 3: dY), pixel group pg = (b >> 4) & 7, channel group cg = b & 15
    int st_tile[2], st_pg[2], st_cg[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int b = t + 256 * k;
        st_tile[k] = b >> 7; st_pg[k] = (b >> 4) & 7; st_cg[k] = b & 15;
    }
    f32x4 ld[2][4];
    unsigned ld_ok = 0;                // bit 4k+i: block k, pixel i lies inside the image (else the staged value is zero)
    // loads of (input row h_in | dY row h_dy) of one image, strip origin w0, into registers.  Always from a valid (clamped)
    // address, zeroed at the store: a load under a branch is waited for on the spot (vmcnt(0) inside the branch), which
    // serialised the eight loads of a step.
    auto issue_loads = [&](const float* xim, const float* dyim, int h_in, int h_dy, int w0) {
        ld_ok = 0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool isdy = st_tile[k] == 3;
            const int h = isdy ? h_dy : h_in;
            const float* base = isdy ? dyim : xim;
            const int wofs = isdy ? 0 : st_tile[k] - 1;             // shift s -> pixel w0 - 1 + s + px
            const bool hok = h >= 0 && h < p.H;
            const int hc = hok ? h : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int w = w0 + wofs + st_pg[k] * 4 + i;
                const bool ok = hok && w >= 0 && w < p.W;
                const int wc = w < 0 ? 0 : (w < p.W ? w : p.W - 1);
                ld[k][i] = *reinterpret_cast<const f32x4*>(base + ((int64_t)hc * p.W + wc) * CH + st_cg[k] * 4);
                ld_ok |= (ok ? 1u : 0u) << (4 * k + i);
            }
        }
    };
    // registers -> LDS: input copies into ring slot `slot`, dY^T into dY buffer `dbuf`
    auto store_lds = [&](int slot, int dbuf) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool isdy = st_tile[k] == 3;
            unsigned char* tb = smem + (isdy ? DY_OFF + dbuf * TILE_B : slot * SLOT_B + st_tile[k] * TILE_B);
            const float mul = isdy ? sc : 1.f;
            const int chunk = st_pg[k] >> 1, half = (st_pg[k] & 1) * 8;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = st_cg[k] * 4 + e;
                uint2 hi, lo;
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = ((ld_ok >> (4 * k + i)) & 1u) ? ld[k][i][e] * mul : 0.f;
                split16(v, hi, lo);
                *reinterpret_cast<uint2*>(tb + tile_off(row, chunk) + half) = hi;
                *reinterpret_cast<uint2*>(tb + tile_off(row, 4 + chunk) + half) = lo;
            }
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;

    const int arow = co0 + li, brow = ci0 + li;
    for (int u = wg; u < units; u += p.nwg) {
        const int im = u / strips, w0 = (u - im * strips) * SW;
        const float* xim = x_g + (int64_t)im * img;
        const float* dyim = dy_g + (int64_t)im * img;
        __syncthreads();                                   // the previous unit's last step has been read
        // prologue: input rows -1, 0, 1 into slots 3, 0, 1 (row h lives in slot h & 3); dY row 0 into buffer 0
        issue_loads(xim, dyim, -1, -1, w0); store_lds(3, 1);               // (dY part: zeros into the idle buffer)
        issue_loads(xim, dyim, 0, 0, w0);   store_lds(0, 0);
        issue_loads(xim, dyim, 1, -1, w0);  store_lds(1, 1);               // (zeros again: buffer 1 is written in step 0)
        __syncthreads();
        for (int h = 0; h < p.H; ++h) {
            // next step's operands: input row h + 2, dY row h + 1 (zeros past the image)
            issue_loads(xim, dyim, h + 2, h + 1 < p.H ? h + 1 : -1, w0);
            __builtin_amdgcn_sched_barrier(0);             // the loads stay HERE: the scheduler otherwise sinks them to their
                                                           // first use behind the MFMAs and their latency is exposed
            const unsigned char* dyt = smem + DY_OFF + (h & 1) * TILE_B;
            u32x4 ah[2], al[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                ah[ks] = *reinterpret_cast<const u32x4*>(dyt + tile_off(arow, ks * 2 + kg));
                al[ks] = *reinterpret_cast<const u32x4*>(dyt + tile_off(arow, 4 + ks * 2 + kg));
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const unsigned char* rowb = smem + ((h - 1 + r) & 3) * SLOT_B;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const unsigned char* tb = rowb + s * TILE_B;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const u32x4 bh = *reinterpret_cast<const u32x4*>(tb + tile_off(brow, ks * 2 + kg));
                        const u32x4 bl = *reinterpret_cast<const u32x4*>(tb + tile_off(brow, 4 + ks * 2 + kg));
                        const h16x8 xh = __builtin_bit_cast(h16x8, ah[ks]), xl = __builtin_bit_cast(h16x8, al[ks]);
                        const h16x8 yh = __builtin_bit_cast(h16x8, bh), yl = __builtin_bit_cast(h16x8, bl);
                        acc[r * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, yh, acc[r * 3 + s], 0, 0, 0);
                        acc[r * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yl, acc[r * 3 + s], 0, 0, 0);
                        acc[r * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yh, acc[r * 3 + s], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // row h + 2 -> slot (h + 2) & 3 (row h - 2's, last read in step h - 1); dY row h + 1 -> the other buffer
            store_lds((h + 2) & 3, (h + 1) & 1);
            __syncthreads();
        }
    }
    // partial of this workgroup: part[g][wg][co][tap * 64 + ci], true scale
    const float inv = 1.f / sc;
    float* out = p.part + ((int64_t)g * p.nwg + wg) * (CH * 576);
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = co0 + (e & 3) + 8 * (e >> 2) + 4 * kg;
            out[co * 576 + q * 64 + ci0 + li] = acc[q][e] * inv;
        }
}

}  // namespace

// dW[g][64][576] += sum over the images of group g; ws: >= groups * nwg * 64 * 576 floats.  Returns the number of workgroups
// per group it used through *nwg_out (the caller combines that many partials).
int launch_wgrad3x3_c64(const float* dy, const float* x, float* ws, int64_t ws_floats, const float* dy_scale_dev, int groups, int B,
                        int H, int W, int* nwg_out, hipStream_t st) {
    if (groups <= 0 || B <= 0 || H <= 0 || W <= 0) return -2;
    if (((uintptr_t)dy & 15) || ((uintptr_t)x & 15) || ((uintptr_t)ws & 15)) return -2;
    const int strips = (W + SW - 1) / SW, units = B * strips;
    int nwg = 256 / groups;                                   // one workgroup per CU (112 KB of LDS each)
    if (nwg < 1) nwg = 1;
    if (nwg > units) nwg = units;
    // equal shares: the largest count not above nwg that divides the units evenly, if that costs at most a quarter
    for (int d = nwg; d >= 1 && d * 4 >= nwg * 3; --d)
        if (units % d == 0) { nwg = d; break; }
    while (nwg > 1 && (int64_t)groups * nwg * CH * 576 > ws_floats) --nwg;
    if ((int64_t)groups * nwg * CH * 576 > ws_floats) return -2;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3x3_c64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B) !=
            hipSuccess)
            return -3;
        attr_set = true;
    }
    Wgrad3Args a{dy, x, ws, dy_scale_dev, B, H, W, nwg};
    prof_begin("wgrad3x3_c64_kernel", 2.0 * CH * 576 * (double)groups * B * H * W, 8.0 * CH * (double)groups * B * H * W, st);
    hipLaunchKernelGGL(wgrad3x3_c64_kernel, dim3(nwg, groups), dim3(256), SMEM_B, st, a);
    prof_end(st);
    if (nwg_out) *nwg_out = nwg;
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
