// wgrad7: weight gradient of the ResNet stem (7x7 / stride 2 / pad 3 convolution of the 4-channel-padded normalised image to 64
// channels; torchvision resnet18.conv1 inside reference backbone.py:95-134) as a DIRECT kernel, f16x3 arithmetic.
//
//   dW[co][(r, s, c)] = sum over output pixels (ho, wo) of  dY[ho][wo][co] * X[2 ho + r - 3][2 wo + s - 3][c]
//
// As a GEMM (gemm.hip, AMODE 3 / BMODE 2: M = 64, N = 196, K = pixels) every 64x64 tile re-reads dY and gathers 16-byte
// pieces of the image per (pixel, tap): 53 TFLOP/s, 9.2 ms per training step.  Here a workgroup walks DOWN a strip of 32
// output columns of one image.  Per output row it stages dY^T [co][32 px] and the TWO new image rows the next output row
// brings in; an image row lives in LDS as the transposed im2col slice of its filter row: 28 rows (s, c) x 32 output pixels,
// entry = X[row][2 wo + s - 3][c] -- the stride-2 column walk and the seven horizontal taps are resolved when the row is
// staged (each thread loads 4 stride-2 pixels of one tap as four 16-byte loads and writes, per channel, their hi / lo halfs as
// 8-byte LDS stores), so every MFMA fragment is an aligned 16-byte read.  Seven filter rows = seven 32-column tiles of the
// result (28 real columns each) against the two 32-row tiles of dY^T: 84 MFMAs per 32 pixels.  Nine row slots (seven live,
// two being staged).  Persistent workgroups, one partial each, summed in a fixed order by launch_splitk_combine.
#include "common.h"
#include "split16.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int CO = 64;                 // output channels
constexpr int SW = 32;                 // output pixels per step
constexpr int ROW_B = 32 * 128;        // one image row's tile: 32 rows (s * 4 + c; 28 real) x [32 px hi | 32 px lo]
constexpr int NSLOT = 9;
constexpr int DY_B = CO * 128;         // dY^T tile
constexpr int DY_OFF = NSLOT * ROW_B;
constexpr int SMEM_B = DY_OFF + 2 * DY_B;          // 52 KB: three workgroups per CU

struct Wgrad7Args {
    const float* dy;          // [G][B][Ho][Wo][64]
    const float* x;           // [G][B][H][W][4]
    float* part;              // [G][nwg][64][196]
    const float* dy_scale;    // device: power-of-two scale of dY (or NULL = 1)
    int B, H, W, Ho, Wo, nwg;
};

__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__global__ __launch_bounds__(256, 2) void wgrad7x7s2_kernel(Wgrad7Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, kg = lane >> 5;
    const int itile = wave & 1;                    // dY^T rows [32 itile, 32 itile + 32)
    const int j0 = wave >> 1;                      // filter rows j0, j0 + 2, j0 + 4 (, 6 for j0 = 0)
    const int g = blockIdx.y, wg = blockIdx.x;
    const int strips = (p.Wo + SW - 1) / SW;
    const int units = p.B * strips;
    const int64_t ximg = (int64_t)p.H * p.W * 4, yimg = (int64_t)p.Ho * p.Wo * CO;
    const float* dy_g = p.dy + (int64_t)g * p.B * yimg;
    const float* x_g = p.x + (int64_t)g * p.B * ximg;
    const float sc = p.dy_scale ? *p.dy_scale : 1.f;

    // rows 28..31 of every image-row tile are the zero padding of the 32-column MFMA tile: written once
    for (int i = t; i < NSLOT * 4 * 8; i += 256) {
        const int slot = i / 32, rem = i - slot * 32, row = 28 + (rem >> 3), chunk = rem & 7;
        *reinterpret_cast<uint4*>(smem + slot * ROW_B + tile_off(row, chunk)) = uint4{0u, 0u, 0u, 0u};
    }

    // staging roles.  Threads 0 .. 111: image block (which of the two new rows, tap s, pixel group pg of 4 output pixels);
    // threads 128 .. 255: dY block (pixel group pg, channel group cg of 4); the rest idle during staging.
    const bool is_x = t < 112, is_dy = t >= 128;
    const int xr = t / 56, xs = (t % 56) >> 3, xpg = t & 7;
    const int ypg = ((t - 128) >> 4) & 7, ycg = (t - 128) & 15;
    f32x4 ld[4];
    unsigned ld_ok = 0;
    // image rows y_a, y_a + 1 (block xr picks one) and dY row h_dy, strip origin wo0; clamped addresses, zeroed at the store
    auto issue_loads = [&](const float* xim, const float* dyim, int y_a, int h_dy, int wo0) {
        ld_ok = 0;
        if (is_x) {
            const int y = y_a + xr;
            const bool yok = y >= 0 && y < p.H;
            const int yc = yok ? y : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = 2 * (wo0 + xpg * 4 + i) + xs - 3;
                const bool ok = yok && col >= 0 && col < p.W;
                const int cc = col < 0 ? 0 : (col < p.W ? col : p.W - 1);
                ld[i] = *reinterpret_cast<const f32x4*>(xim + ((int64_t)yc * p.W + cc) * 4);
                ld_ok |= (ok ? 1u : 0u) << i;
            }
        } else if (is_dy) {
            const bool hok = h_dy >= 0 && h_dy < p.Ho;
            const int hc = hok ? h_dy : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int w = wo0 + ypg * 4 + i;
                const bool ok = hok && w < p.Wo;
                const int wc = w < p.Wo ? w : p.Wo - 1;
                ld[i] = *reinterpret_cast<const f32x4*>(dyim + ((int64_t)hc * p.Wo + wc) * CO + ycg * 4);
                ld_ok |= (ok ? 1u : 0u) << i;
            }
        }
    };
    // registers -> LDS: image rows y_a, y_a + 1 into their slots (row y lives in slot (y + 9) % 9), dY^T into buffer dbuf
    auto store_lds = [&](int y_a, int dbuf) {
        if (!is_x && !is_dy) return;
        unsigned char* tb;
        int row0, pg;
        float mul;
        if (is_x) {
            const int y = y_a + xr;
            tb = smem + ((y + 9 * 1024) % NSLOT) * ROW_B; row0 = xs * 4; pg = xpg; mul = 1.f;
        } else {
            tb = smem + DY_OFF + dbuf * DY_B; row0 = ycg * 4; pg = ypg; mul = sc;
        }
        const int chunk = pg >> 1, half = (pg & 1) * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ((ld_ok >> i) & 1u) ? ld[i][e] * mul : 0.f;
            uint2 hi, lo;
            split16(v, hi, lo);
            *reinterpret_cast<uint2*>(tb + tile_off(row0 + e, chunk) + half) = hi;
            *reinterpret_cast<uint2*>(tb + tile_off(row0 + e, 4 + chunk) + half) = lo;
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;

    const int arow = itile * 32 + li;
    for (int u = wg; u < units; u += p.nwg) {
        const int im = u / strips, wo0 = (u - im * strips) * SW;
        const float* xim = x_g + (int64_t)im * ximg;
        const float* dyim = dy_g + (int64_t)im * yimg;
        __syncthreads();                                   // the previous unit's last step has been read
        // prologue: image rows -3 .. 3 (output row 0's seven filter rows) and dY row 0; row pairs (-3,-2), (-1,0), (1,2), then
        // (3,4) together with dY row 0 -- row 4 belongs to output row 1 and is restaged with row 5 in step 0 (same bytes)
        issue_loads(xim, dyim, -3, -1, wo0); store_lds(-3, 1);
        issue_loads(xim, dyim, -1, -1, wo0); store_lds(-1, 1);
        issue_loads(xim, dyim, 1, -1, wo0);  store_lds(1, 1);
        issue_loads(xim, dyim, 3, 0, wo0);   store_lds(3, 0);
        __syncthreads();
        for (int ho = 0; ho < p.Ho; ++ho) {
            // next output row's operands: image rows 2 ho + 4, 2 ho + 5; dY row ho + 1
            issue_loads(xim, dyim, 2 * ho + 4, ho + 1 < p.Ho ? ho + 1 : -1, wo0);
            __builtin_amdgcn_sched_barrier(0);             // the loads stay at the top of the step (wgrad3.hip)
            const unsigned char* dyt = smem + DY_OFF + (ho & 1) * DY_B;
            u32x4 ah[2], al[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                ah[ks] = *reinterpret_cast<const u32x4*>(dyt + tile_off(arow, ks * 2 + kg));
                al[ks] = *reinterpret_cast<const u32x4*>(dyt + tile_off(arow, 4 + ks * 2 + kg));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = j0 + 2 * q;                  // filter row
                if (r < 7) {
                    const unsigned char* tb = smem + ((2 * ho + r - 3 + 9 * 1024) % NSLOT) * ROW_B;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const u32x4 bh = *reinterpret_cast<const u32x4*>(tb + tile_off(li, ks * 2 + kg));
                        const u32x4 bl = *reinterpret_cast<const u32x4*>(tb + tile_off(li, 4 + ks * 2 + kg));
                        const h16x8 xh = __builtin_bit_cast(h16x8, ah[ks]), xl = __builtin_bit_cast(h16x8, al[ks]);
                        const h16x8 yh = __builtin_bit_cast(h16x8, bh), yl = __builtin_bit_cast(h16x8, bl);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, yh, acc[q], 0, 0, 0);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yl, acc[q], 0, 0, 0);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yh, acc[q], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // rows 2 ho + 4, 2 ho + 5 -> the slots of rows 2 ho - 5, 2 ho - 4 (last read in step ho - 1); dY row ho + 1
            store_lds(2 * ho + 4, (ho + 1) & 1);
            __syncthreads();
        }
    }
    // partial of this workgroup: part[g][wg][co][r * 28 + s * 4 + c], true scale
    const float inv = 1.f / sc;
    float* out = p.part + ((int64_t)g * p.nwg + wg) * (CO * 196);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = j0 + 2 * q;
        if (r < 7 && li < 28) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = itile * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
                out[co * 196 + r * 28 + li] = acc[q][e] * inv;
            }
        }
    }
}

}  // namespace

// dW[g][64][196] partials over the images of group g: ws >= groups * nwg * 64 * 196 floats; *nwg_out = partials per group
int launch_wgrad7x7s2(const float* dy, const float* x4, float* ws, int64_t ws_floats, const float* dy_scale_dev, int groups, int B,
                      int H, int W, int Ho, int Wo, int* nwg_out, hipStream_t st) {
    if (groups <= 0 || B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return -2;
    if (Ho != (H + 6 - 7) / 2 + 1 || Wo != (W + 6 - 7) / 2 + 1) return -2;
    if (((uintptr_t)dy & 15) || ((uintptr_t)x4 & 15) || ((uintptr_t)ws & 15)) return -2;
    const int strips = (Wo + SW - 1) / SW, units = B * strips;
    int nwg = 512 / groups;                                   // two workgroups per CU (launch bounds), 52 KB of LDS each
    if (nwg < 1) nwg = 1;
    if (nwg > units) nwg = units;
    for (int d = nwg; d >= 1 && d * 4 >= nwg * 3; --d)
        if (units % d == 0) { nwg = d; break; }
    while (nwg > 1 && (int64_t)groups * nwg * CO * 196 > ws_floats) --nwg;
    if ((int64_t)groups * nwg * CO * 196 > ws_floats) return -2;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad7x7s2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B) !=
            hipSuccess)
            return -3;
        attr_set = true;
    }
    Wgrad7Args a{dy, x4, ws, dy_scale_dev, B, H, W, Ho, Wo, nwg};
    prof_begin("wgrad7x7s2_kernel", 2.0 * CO * 196 * (double)groups * B * Ho * Wo,
               4.0 * (double)groups * B * ((double)Ho * Wo * CO + (double)H * W * 4), st);
    hipLaunchKernelGGL(wgrad7x7s2_kernel, dim3(nwg, groups), dim3(256), SMEM_B, st, a);
    prof_end(st);
    if (nwg_out) *nwg_out = nwg;
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
