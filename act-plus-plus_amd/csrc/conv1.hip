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
#include <cstdlib>

namespace {

constexpr int KREAL = 147, KPAD = 148, WSTRIDE = 149;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
constexpr int TILE_P = 64;                 // output pixels per tile
constexpr int PCOLS = 2 * TILE_P + 5;      // 133 input columns
constexpr int PSTRIDE = 400;               // floats per patch row (133*3 = 399, padded)
constexpr int PROWS = 8;                   // 7 real rows + 1 zero row for the K pad

__host__ __device__ constexpr int koff(int k) { return (k / 21) * PSTRIDE + (k % 21); }

// number of per-thread staging registers: u8 rows are fetched as 4-byte words (7 rows x 101 words), f32 as scalars
template <int FMT> struct StageN { static constexpr int value = (FMT == 0) ? 3 : 11; };

template <int FMT>
__global__ __launch_bounds__(256) void conv1_kernel(Conv1Args p, int tiles_per_row, int tiles_per_cam) {
    // one dynamic LDS block (66.8 KB > the 64 KB static limit): weights | two patch buffers | LUT, 16-byte aligned carves
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* s_w = reinterpret_cast<float*>(smem_raw);
    float (*s_patch)[PROWS * PSTRIDE] = reinterpret_cast<float (*)[PROWS * PSTRIDE]>(s_w + 64 * WSTRIDE);
    float* s_lut = &s_patch[0][0] + 2 * PROWS * PSTRIDE;
    const int cam = blockIdx.y + p.cam0;
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
    for (int e = t; e < 2 * PROWS * PSTRIDE; e += 256) (&s_patch[0][0])[e] = 0.f;
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float stdv[3] = {0.229f, 0.224f, 0.225f};
    __syncthreads();

    const bool active = ntile * 32 < p.Cout;
    const int n = ntile * 32 + li;
    const float sc = (n < p.Cout) ? p.scale[cam * p.Cout + n] : 0.f;
    const float bi = (n < p.Cout) ? p.bias[cam * p.Cout + n] : 0.f;
    const float* b_base = s_w + n * WSTRIDE + lh;

    // ---- staging of one tile's 7 x 133 x 3 patch, split in two phases so that the global loads of tile i+1 fly while the
    //      MFMAs of tile i run: fetch() issues the loads into registers, commit() normalises and writes LDS.
    constexpr int NS = StageN<FMT>::value;
    uint32_t sreg[NS];
    auto tile_coords = [&](int tile, int& b, int& ho, int& wo0) {
        b = tile / (p.Ho * tiles_per_row);
        const int rem = tile - b * (p.Ho * tiles_per_row);
        ho = rem / tiles_per_row;
        wo0 = (rem - ho * tiles_per_row) * TILE_P;
    };
    auto fetch = [&](int tile) {
        int b, ho, wo0;
        tile_coords(tile, b, ho, wo0);
        const int64_t img = (int64_t)b * p.C + cam;       // input image index ([B][C] layout of the caller)
        const int hi0 = 2 * ho - 3, wi0 = 2 * wo0 - 3;
        if (FMT == 0) {
            // row r of the patch = 399 consecutive bytes starting at byte (hi*W + wi0)*3 of the image; fetched as
            // 101 aligned 4-byte words (clamped into the image; out-of-image bytes are discarded in commit())
            const uint8_t* src = reinterpret_cast<const uint8_t*>(p.image) + img * (int64_t)p.H * p.W * 3;
            const int64_t img_bytes = (int64_t)p.H * p.W * 3;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int e = t + 256 * i;
                const int r = e / 101, j = e - r * 101;
                const int hi = hi0 + r;
                uint32_t v = 0;
                if (r < 7 && (unsigned)hi < (unsigned)p.H) {
                    const int64_t a0 = ((int64_t)hi * p.W + wi0) * 3;          // may be negative / past the row: clamp below
                    int64_t wa = ((a0 >> 2) + j) << 2;                          // aligned word address (floor for negatives)
                    if (wa < 0) wa = 0;
                    if (wa > img_bytes - 4) wa = (img_bytes - 4) & ~int64_t(3);
                    v = *reinterpret_cast<const uint32_t*>(src + wa);
                }
                sreg[i] = v;
            }
        } else {
            const float* src = reinterpret_cast<const float*>(p.image) + img * 3 * (int64_t)p.H * p.W;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int e = t + 256 * i;
                const int rc = e / PCOLS, pc = e - rc * PCOLS;
                const int r = rc / 3, c = rc - r * 3;
                const int hi = hi0 + r, wi = wi0 + pc;
                float v = 0.f;
                if (e < 7 * 3 * PCOLS && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                    v = src[((int64_t)c * p.H + hi) * p.W + wi];
                sreg[i] = __float_as_uint(v);
            }
        }
    };
    auto commit = [&](int tile, float* patch) {
        int b, ho, wo0;
        tile_coords(tile, b, ho, wo0);
        const int hi0 = 2 * ho - 3, wi0 = 2 * wo0 - 3;
        if (FMT == 0) {
            const int64_t img_bytes = (int64_t)p.H * p.W * 3;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int e = t + 256 * i;
                const int r = e / 101, j = e - r * 101;
                const int hi = hi0 + r;
                if (r >= 7) continue;
                const bool row_ok = (unsigned)hi < (unsigned)p.H;
                const int64_t a0 = ((int64_t)hi * p.W + wi0) * 3;
                int64_t wa = ((a0 >> 2) + j) << 2;
                const int64_t wa_req = wa;
                if (wa < 0) wa = 0;
                if (wa > img_bytes - 4) wa = (img_bytes - 4) & ~int64_t(3);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int64_t ba = wa_req + k;                 // byte address this slot stands for
                    const int x = (int)(ba - a0);                  // position inside the 399-byte patch row
                    if (x < 0 || x >= PCOLS * 3) continue;
                    const int pc = x / 3, c = x - pc * 3;
                    const int wi = wi0 + pc;
                    float v = 0.f;
                    // the word was clamped only when it lies (partly) outside the image: such bytes are padding anyway
                    if (row_ok && (unsigned)wi < (unsigned)p.W && wa == wa_req)
                        v = s_lut[c * 256 + ((sreg[i] >> (8 * k)) & 0xFF)];
                    else if (row_ok && (unsigned)wi < (unsigned)p.W) {
                        const int64_t sh = ba - wa;                // clamped word still contains this byte if 0 <= sh < 4
                        if (sh >= 0 && sh < 4) v = s_lut[c * 256 + ((sreg[i] >> (8 * sh)) & 0xFF)];
                    }
                    patch[r * PSTRIDE + x] = v;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int e = t + 256 * i;
                if (e >= 7 * 3 * PCOLS) continue;
                const int rc = e / PCOLS, pc = e - rc * PCOLS;
                const int r = rc / 3, c = rc - r * 3;
                const int hi = hi0 + r, wi = wi0 + pc;
                float v = 0.f;
                if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                    v = (__uint_as_float(sreg[i]) - mean[c]) / stdv[c];
                patch[r * PSTRIDE + pc * 3 + c] = v;
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < tiles_per_cam) {
        fetch(tile);
        commit(tile, s_patch[0]);
    }
    __syncthreads();
    int cur = 0;
    for (; tile < tiles_per_cam; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        const bool has_next = next < tiles_per_cam;
        if (has_next) fetch(next);
        int b, ho, wo0;
        tile_coords(tile, b, ho, wo0);
        const int64_t oimg = (int64_t)cam * p.B + b;     // output is camera-major: each camera is one GEMM group
        if (active) {
            const float* a_base = s_patch[cur] + (mtile * 32 + li) * 6;
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
        if (has_next) commit(next, s_patch[cur ^ 1]);     // the other buffer was last read one iteration ago
        __syncthreads();
        cur ^= 1;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// fp16-split stem (PREC f16x3, see gemm.hip): same contract and outputs, products on v_mfma_f32_32x32x16_f16 with both
// operands split exactly into (hi, lo) fp16 pieces.
//   tile  = 2 output rows x 64 output pixels x all Cout channels; wave w owns row (w>>1), pixels 32*(w&1).., and BOTH
//           32-channel MFMA tiles, so one A fragment feeds 6 MFMAs.
//   patch = 9 input rows x 399 values staged as halfs, hi row | lo row (816 B each); the u8 path needs no conversion at
//           all: the lookup table holds the split of every normalised byte value ((hi | lo << 16) per entry).
//   K     = 7 filter rows x 24 (21 real taps x channels + 3 zero-weight pad) = 21 groups of 8, two groups per MFMA step
//           (one per lane half), 11 steps; the A group of pixel p, filter row r, group g is the 16 bytes at
//           patch[2*row + r][6p + 8g]: 4-byte aligned only, hence 4 ds_read_b32.
//   B     = this camera's weights x 2^8 (keeps the lo pieces normal fp16 numbers; undone in the epilogue scale) as
//           [n][22 groups][8 halfs] hi | lo, row stride 368 B (conflict-free ds_read_b128).
// 66 MFMAs of 32 cycles per wave and 128 output pixels, against 74 of 64 cycles per 64 pixels for the fp32 kernel.
constexpr int F_TP = 64, F_ROWS = 2, F_PROWS = 2 * F_ROWS + 5;      // 9 input rows
constexpr int F_RB = 816;                                          // bytes per half-row (408 halfs)
constexpr int F_PATCH = F_PROWS * 2 * F_RB;                        // bytes per patch buffer
constexpr int F_NG = 22;                                           // contraction groups incl. the zero one
constexpr int F_WROW = 368;                                        // bytes per channel row of one weight piece
constexpr int F_WBYTES = 64 * F_WROW;
constexpr int F_SMEM = 2 * F_WBYTES + 2 * F_PATCH + 3 * 256 * 4;
constexpr float F_WSCALE = 256.f;      // default of Conv1Args::wscale

// 4x4 transpose across a quad of lanes: before, lane j of the quad holds (r0..r3) = row j; after, lane k holds column k
// as (r0..r3) = (row0[k], row1[k], row2[k], row3[k]).  Two exchange stages on DPP quad permutes (no LDS traffic).
__device__ __forceinline__ float dpp_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }
__device__ __forceinline__ void quad_transpose(float& r0, float& r1, float& r2, float& r3, int k) {
    const bool o1 = k & 1, o2 = k & 2;
    // stage 1: lanes differing in bit 0 swap (r1 of the even lane) <-> (r0 of the odd lane), and r3 <-> r2
    float t = dpp_xor1(o1 ? r0 : r1);
    if (o1) r0 = t; else r1 = t;
    t = dpp_xor1(o1 ? r2 : r3);
    if (o1) r2 = t; else r3 = t;
    // stage 2: lanes differing in bit 1 swap (r2, r3 of the low pair) <-> (r0, r1 of the high pair)
    t = dpp_xor2(o2 ? r0 : r2);
    if (o2) r0 = t; else r2 = t;
    t = dpp_xor2(o2 ? r1 : r3);
    if (o2) r1 = t; else r3 = t;
}

__device__ __forceinline__ uint32_t split1(float v) {              // (hi | lo << 16) of one value
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    return (uint32_t)__builtin_bit_cast(uint16_t, h) | ((uint32_t)__builtin_bit_cast(uint16_t, l) << 16);
}

// one entry of the weight image: element e = (n, group gi = r*3+g, j) -> (hi, lo) halfs of w[n][r*21 + 8g + j] * 2^8
__device__ __forceinline__ void conv1_wimg_entry(const float* wg, int Cout, int e, float wscale, uint16_t& h, uint16_t& l) {
    const int n = e / (F_NG * 8), rem = e - n * (F_NG * 8);
    const int gi = rem >> 3, j = rem & 7;
    const int r = gi / 3, g = gi - r * 3, x = 8 * g + j;
    float v = 0.f;
    if (n < Cout && gi < 21 && x < 21) v = wg[n * KPAD + r * 21 + x] * wscale;
    const uint32_t hl = split1(v);
    h = (uint16_t)(hl & 0xffffu);
    l = (uint16_t)(hl >> 16);
}

// the LDS weight image of conv1_f16x3_kernel for every camera: [cam][hi piece | lo piece], F_WBYTES each
__global__ __launch_bounds__(256) void conv1_wimg_kernel(const float* __restrict__ w, unsigned char* __restrict__ img, int Cout, float wscale) {
    const int cam = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 64 * F_NG * 8) return;
    uint16_t h, l;
    conv1_wimg_entry(w + (int64_t)cam * Cout * KPAD, Cout, e, wscale, h, l);
    const int n = e / (F_NG * 8), rem = e - n * (F_NG * 8);
    unsigned char* dst = img + (int64_t)cam * 2 * F_WBYTES;
    *reinterpret_cast<uint16_t*>(dst + n * F_WROW + rem * 2) = h;
    *reinterpret_cast<uint16_t*>(dst + F_WBYTES + n * F_WROW + rem * 2) = l;
}

template <int FMT>
__global__ __launch_bounds__(256) void conv1_f16x3_kernel(Conv1Args p, int tiles_per_row, int tiles_per_cam) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned char* s_wh = reinterpret_cast<unsigned char*>(smem_raw);
    unsigned char* s_wl = s_wh + F_WBYTES;
    unsigned char* s_patch = s_wl + F_WBYTES;                      // two buffers of F_PATCH bytes
    uint32_t* s_lut = reinterpret_cast<uint32_t*>(s_patch + 2 * F_PATCH);
    const int cam = blockIdx.y + p.cam0;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int prow = wave >> 1, phalf = wave & 1;

    // weights: [Cout][148] (k = r*21 + x) -> [n][group r*3+g][8], split, scaled; everything else zero.  The engine hands
    // over the finished image (built once per weight update, conv1_wimg_kernel): a straight 46 KB copy instead of 44
    // gather / split / 2-byte-store rounds per workgroup, which dominated the launch at small batches.
    if (p.wimg) {
        const uint4* src = reinterpret_cast<const uint4*>(p.wimg + (int64_t)cam * 2 * F_WBYTES);
        for (int e = t; e < 2 * F_WBYTES / 16; e += 256) reinterpret_cast<uint4*>(s_wh)[e] = src[e];
    } else {
        const float* wg = p.w + (int64_t)cam * p.Cout * KPAD;
        for (int e = t; e < 64 * F_NG * 8; e += 256) {
            uint16_t h, l;
            conv1_wimg_entry(wg, p.Cout, e, p.wscale, h, l);
            const int n = e / (F_NG * 8), rem = e - n * (F_NG * 8);
            *reinterpret_cast<uint16_t*>(s_wh + n * F_WROW + rem * 2) = h;
            *reinterpret_cast<uint16_t*>(s_wl + n * F_WROW + rem * 2) = l;
        }
    }
    for (int e = t; e < 3 * 256; e += 256) s_lut[e] = (FMT == 0) ? split1(p.lut[e]) : 0u;
    for (int e = t; e < 2 * F_PATCH / 4; e += 256) reinterpret_cast<uint32_t*>(s_patch)[e] = 0u;
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float stdv[3] = {0.229f, 0.224f, 0.225f};
    __syncthreads();

    float sc[2], bi[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = nt * 32 + li;
        sc[nt] = (n < p.Cout) ? p.scale[cam * p.Cout + n] * (1.f / p.wscale) : 0.f;
        bi[nt] = (n < p.Cout) ? p.bias[cam * p.Cout + n] : 0.f;
    }

    constexpr int NS = (FMT == 0) ? (F_PROWS * 101 + 255) / 256 : (F_PROWS * 3 * PCOLS + 255) / 256;
    uint32_t sreg[NS];
    const int hpairs = (p.Ho + F_ROWS - 1) / F_ROWS;
    auto tile_coords = [&](int tile, int& b, int& ho0, int& wo0) {
        b = tile / (hpairs * tiles_per_row);
        const int rem = tile - b * (hpairs * tiles_per_row);
        const int hp = rem / tiles_per_row;
        ho0 = hp * F_ROWS;
        wo0 = (rem - hp * tiles_per_row) * F_TP;
    };
    auto fetch = [&](int tile) {
        int b, ho0, wo0;
        tile_coords(tile, b, ho0, wo0);
        const int64_t img = (int64_t)b * p.C + cam;
        const int hi0 = 2 * ho0 - 3, wi0 = 2 * wo0 - 3;
        if (FMT == 0) {
            const uint8_t* src = reinterpret_cast<const uint8_t*>(p.image) + img * (int64_t)p.H * p.W * 3;
            const int64_t img_bytes = (int64_t)p.H * p.W * 3;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int e = t + 256 * i;
                const int r = e / 101, j = e - r * 101;
                const int hi = hi0 + r;
                uint32_t v = 0;
                if (r < F_PROWS && (unsigned)hi < (unsigned)p.H) {
                    const int64_t a0 = ((int64_t)hi * p.W + wi0) * 3;
                    int64_t wa = ((a0 >> 2) + j) << 2;
                    if (wa < 0) wa = 0;
                    if (wa > img_bytes - 4) wa = (img_bytes - 4) & ~int64_t(3);
                    v = *reinterpret_cast<const uint32_t*>(src + wa);
                }
                sreg[i] = v;
            }
        } else {
            const float* src = reinterpret_cast<const float*>(p.image) + img * 3 * (int64_t)p.H * p.W;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int e = t + 256 * i;
                const int rc = e / PCOLS, pc = e - rc * PCOLS;
                const int r = rc / 3, c = rc - r * 3;
                const int hi = hi0 + r, wi = wi0 + pc;
                float v = 0.f;
                if (e < F_PROWS * 3 * PCOLS && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                    v = src[((int64_t)c * p.H + hi) * p.W + wi];
                sreg[i] = __float_as_uint(v);
            }
        }
    };
    auto put = [&](unsigned char* patch, int r, int x, uint32_t hl) {
        *reinterpret_cast<uint16_t*>(patch + r * 2 * F_RB + x * 2) = (uint16_t)(hl & 0xffffu);
        *reinterpret_cast<uint16_t*>(patch + r * 2 * F_RB + F_RB + x * 2) = (uint16_t)(hl >> 16);
    };
    auto commit = [&](int tile, unsigned char* patch) {
        int b, ho0, wo0;
        tile_coords(tile, b, ho0, wo0);
        const int hi0 = 2 * ho0 - 3, wi0 = 2 * wo0 - 3;
        if (FMT == 0) {
            const int64_t img_bytes = (int64_t)p.H * p.W * 3;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int e = t + 256 * i;
                const int r = e / 101, j = e - r * 101;
                const int hi = hi0 + r;
                if (r >= F_PROWS) continue;
                const bool row_ok = (unsigned)hi < (unsigned)p.H;
                const int64_t a0 = ((int64_t)hi * p.W + wi0) * 3;
                int64_t wa = ((a0 >> 2) + j) << 2;
                const int64_t wa_req = wa;
                if (wa < 0) wa = 0;
                if (wa > img_bytes - 4) wa = (img_bytes - 4) & ~int64_t(3);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int64_t ba = wa_req + k;
                    const int x = (int)(ba - a0);
                    if (x < 0 || x >= PCOLS * 3) continue;
                    const int pc = x / 3, c = x - pc * 3;
                    const int wi = wi0 + pc;
                    uint32_t hl = 0u;
                    if (row_ok && (unsigned)wi < (unsigned)p.W) {
                        const int64_t sh = ba - wa;            // the (possibly clamped) word holds this byte if 0 <= sh < 4
                        if (sh >= 0 && sh < 4) hl = s_lut[c * 256 + ((sreg[i] >> (8 * sh)) & 0xFF)];
                    }
                    put(patch, r, x, hl);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int e = t + 256 * i;
                if (e >= F_PROWS * 3 * PCOLS) continue;
                const int rc = e / PCOLS, pc = e - rc * PCOLS;
                const int r = rc / 3, c = rc - r * 3;
                const int hi = hi0 + r, wi = wi0 + pc;
                float v = 0.f;
                if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                    v = (__uint_as_float(sreg[i]) - mean[c]) / stdv[c];
                put(patch, r, pc * 3 + c, split1(v));
            }
        }
    };

    // Tile walk.  Plain mode: tiles blockIdx.x, +gridDim.x, ... in any order.  vpool mode (inference): a workgroup owns a
    // chain = (image, 64-column strip, segment of row pairs) and walks DOWN it, one row pair per step, so that the lower
    // conv row of the previous step stays in registers: the vertical 3-max of the 3x3/s2 pool (rows 2a-1, 2a, 2a+1) is
    // formed in the epilogue and only the vertically pooled map [Ho/2][Wo][Cout] is written -- half the bytes, and the
    // pool pass that follows reads half as much.  A chain that does not start at the top first computes the row pair above
    // it without storing (halo).  Values are post-ReLU (>= 0), so 0 stands in for the padded row above the image.
    int tile = blockIdx.x, tile_stride = gridDim.x, tile_end = tiles_per_cam, store_from = 0;
    if (p.vpool) {
        const int chains_per_img = tiles_per_row * p.vpool_nseg;
        const int b = blockIdx.x / chains_per_img, rem = blockIdx.x - b * chains_per_img;
        const int strip = rem / p.vpool_nseg, seg = rem - strip * p.vpool_nseg;
        const int len = (hpairs + p.vpool_nseg - 1) / p.vpool_nseg;
        const int hp0 = seg * len, hp1 = (hp0 + len < hpairs) ? hp0 + len : hpairs;
        const int first = hp0 > 0 ? hp0 - 1 : 0;
        tile = (b * hpairs + first) * tiles_per_row + strip;
        tile_stride = tiles_per_row;
        tile_end = hp0 < hp1 ? (b * hpairs + hp1 - 1) * tiles_per_row + strip + 1 : 0;      // empty segment: no steps
        store_from = (b * hpairs + hp0) * tiles_per_row + strip;
    }
    if (tile < tile_end) {
        fetch(tile);
        commit(tile, s_patch);
    }
    __syncthreads();
    int cur = 0;
    const unsigned char* bbase = s_wh + li * F_WROW + lh * 16;
    float prev[2][2][4];                  // vpool: lower conv row of the previous step (this lane's channel, 8 columns)
#pragma unroll
    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
            for (int c_ = 0; c_ < 4; ++c_) prev[a_][b_][c_] = 0.f;
    for (; tile < tile_end; tile += tile_stride) {
        const int next = tile + tile_stride;
        const bool has_next = next < tile_end;
        if (has_next) fetch(next);
        int b, ho0, wo0;
        tile_coords(tile, b, ho0, wo0);
        const int64_t oimg = (int64_t)cam * p.B + b;
        {
            // plain: wave = (row, 32-column half), lane pixel = column.  vpool: wave = 16-column quarter, lane pixel =
            // (row = li >> 4, column = li & 15), so that both rows of a column sit in the same lane's accumulators
            const unsigned char* abase = p.vpool
                ? s_patch + cur * F_PATCH + (2 * (li >> 4)) * 2 * F_RB + 12 * (wave * 16 + (li & 15))
                : s_patch + cur * F_PATCH + (2 * prow) * 2 * F_RB + 12 * (phalf * 32 + li);
            f32x16 acc[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
#pragma unroll
            for (int s = 0; s < 11; ++s) {
                // this lane half's contraction group 2s+lh -> (filter row, 8-wide window); the 22nd group has zero weights
                constexpr int NOFF = 0;
                const int g0 = 2 * s, g1 = (2 * s + 1 < 21) ? 2 * s + 1 : 20;
                const int off0 = (g0 / 3) * 2 * F_RB + (g0 % 3) * 16, off1 = (g1 / 3) * 2 * F_RB + (g1 % 3) * 16;
                const unsigned char* ap = abase + (lh ? off1 : off0) + NOFF;
                uint4 ah, al;
                ah.x = *reinterpret_cast<const uint32_t*>(ap + 0);  ah.y = *reinterpret_cast<const uint32_t*>(ap + 4);
                ah.z = *reinterpret_cast<const uint32_t*>(ap + 8);  ah.w = *reinterpret_cast<const uint32_t*>(ap + 12);
                al.x = *reinterpret_cast<const uint32_t*>(ap + F_RB + 0);  al.y = *reinterpret_cast<const uint32_t*>(ap + F_RB + 4);
                al.z = *reinterpret_cast<const uint32_t*>(ap + F_RB + 8);  al.w = *reinterpret_cast<const uint32_t*>(ap + F_RB + 12);
                const h16x8 xh = __builtin_bit_cast(h16x8, ah), xl = __builtin_bit_cast(h16x8, al);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const h16x8 yh = __builtin_bit_cast(h16x8, *reinterpret_cast<const uint4*>(bbase + nt * 32 * F_WROW + s * 32));
                    const h16x8 yl = __builtin_bit_cast(h16x8, *reinterpret_cast<const uint4*>(bbase + F_WBYTES + nt * 32 * F_WROW + s * 32));
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, yh, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yl, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yh, acc[nt], 0, 0, 0);
                }
            }
            if (p.vpool) {
                // accumulator register e = 4*gq + i holds pixel 8*gq + 4*lh + i: gq 0,1 = upper row, gq 2,3 = lower row
                const int q4 = (li >> 2) * 4, k = li & 3;
                const bool do_store = tile >= store_from;            // block-uniform
                const int hp = ho0 >> 1;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                    for (int gq = 0; gq < 2; ++gq) {
                        float v[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float up = fmaxf(acc[nt][4 * gq + i] * sc[nt] + bi[nt], 0.f);
                            const float lo = fmaxf(acc[nt][4 * (gq + 2) + i] * sc[nt] + bi[nt], 0.f);
                            v[i] = fmaxf(prev[nt][gq][i], fmaxf(up, lo));
                            prev[nt][gq][i] = lo;
                        }
                        quad_transpose(v[0], v[1], v[2], v[3], k);
                        const int wo = wo0 + wave * 16 + 8 * gq + 4 * lh + k;
                        const int nb = nt * 32 + q4;
                        if (do_store && wo < p.Wo && nb < p.Cout) {
                            float* dst = p.out + ((oimg * (p.Ho >> 1) + hp) * (int64_t)p.Wo + wo) * p.Cout + nb;
                            *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                        }
                    }
                }
            }
            const int ho = ho0 + prow;
            if (!p.vpool && ho < p.Ho) {
                // C layout: lane = channel, registers = pixels.  A 4x4 transpose inside each quad of lanes (4 channels x
                // 4 consecutive pixels, two DPP quad-permute stages) gives every lane 4 consecutive channels of ONE
                // pixel: 16-byte stores instead of 4-byte ones (the scalar form ran this epilogue at ~1.5 TB/s).
                const int q4 = (li >> 2) * 4, k = li & 3;
                const bool vec_ok = (p.Cout & 3) == 0;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int n = nt * 32 + li;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        float r0 = fmaxf(acc[nt][4 * gq + 0] * sc[nt] + bi[nt], p.relu_floor);
                        float r1 = fmaxf(acc[nt][4 * gq + 1] * sc[nt] + bi[nt], p.relu_floor);
                        float r2 = fmaxf(acc[nt][4 * gq + 2] * sc[nt] + bi[nt], p.relu_floor);
                        float r3 = fmaxf(acc[nt][4 * gq + 3] * sc[nt] + bi[nt], p.relu_floor);
                        if (vec_ok) {
                            quad_transpose(r0, r1, r2, r3, k);
                            const int wo = wo0 + phalf * 32 + 8 * gq + 4 * lh + k;
                            const int nb = nt * 32 + q4;
                            if (wo < p.Wo && nb < p.Cout) {
                                float* dst = p.out + ((oimg * p.Ho + ho) * (int64_t)p.Wo + wo) * p.Cout + nb;
                                *reinterpret_cast<f32x4*>(dst) = f32x4{r0, r1, r2, r3};
                            }
                        } else if (n < p.Cout) {
                            float* orow = p.out + ((oimg * p.Ho + ho) * (int64_t)p.Wo) * p.Cout + n;
                            const float rr[4] = {r0, r1, r2, r3};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int wo = wo0 + phalf * 32 + e + 8 * gq + 4 * lh;
                                if (wo < p.Wo) orow[(int64_t)wo * p.Cout] = rr[e];
                            }
                        }
                    }
                }
            }
        }
        if (has_next) commit(next, s_patch + (cur ^ 1) * F_PATCH);
        __syncthreads();
        cur ^= 1;
    }
}

}  // namespace

int64_t conv1_wimg_bytes() { return 2 * F_WBYTES; }

int launch_conv1_wimg(const float* w, void* img, int C, int Cout, hipStream_t st, float wscale) {
    // the pad bytes of each row (22 groups x 16 B = 352 of 368) are never read; zeroed so that the image is deterministic
    if (hipMemsetAsync(img, 0, (size_t)C * 2 * F_WBYTES, st) != hipSuccess) return -3;
    hipLaunchKernelGGL(conv1_wimg_kernel, dim3((64 * F_NG * 8 + 255) / 256, C), dim3(256), 0, st, w,
                       reinterpret_cast<unsigned char*>(img), Cout, wscale);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_conv1(const Conv1Args& a, hipStream_t st, std::string* err) {
    if (a.Cout > 64 || a.Cout < 1) { if (err) *err = "conv1: Cout must be in 1..64"; return -2; }
    if (a.Ho != (a.H + 6 - 7) / 2 + 1 || a.Wo != (a.W + 6 - 7) / 2 + 1) { if (err) *err = "conv1: bad output size"; return -2; }
    static const int env_prec = [] {
        const char* e = getenv("ACTMI_GEMM_PREC");
        if (!e) return ACTMI_PREC_F32;
        return (e[0] == 'f' && e[1] == '3') ? ACTMI_PREC_F32 : ACTMI_PREC_F16X3;
    }();
    const int prec = a.prec ? a.prec : env_prec;
    if (prec == ACTMI_PREC_F16X3) {
        const int tiles_per_row = (a.Wo + F_TP - 1) / F_TP;
        const int tiles_per_cam = a.B * ((a.Ho + F_ROWS - 1) / F_ROWS) * tiles_per_row;
        static const int blocks_target = getenv("ACTMI_CONV1_BLOCKS") ? atoi(getenv("ACTMI_CONV1_BLOCKS")) : 512;   // tuning aid; 512 measured best at B = 1, 2, 8 (tools/conv1_blocks_sweep.sh)
        int cap = blocks_target / (a.C > 0 ? a.C : 1);
        if (cap < 1) cap = 1;
        int gx = tiles_per_cam < cap ? tiles_per_cam : cap;
        const int per = (tiles_per_cam + gx - 1) / gx;
        gx = (tiles_per_cam + per - 1) / per;
        Conv1Args av = a;
        if (a.vpool) {
            if ((a.Ho & 1) || (a.Cout & 3)) { if (err) *err = "conv1: vpool needs an even output height and Cout % 4 == 0"; return -2; }
            const int hpairs = a.Ho / 2;
            const int nc_l = a.ncam > 0 ? a.ncam : a.C;
            int nseg = blocks_target / (nc_l * a.B * tiles_per_row > 0 ? nc_l * a.B * tiles_per_row : 1);
            if (nseg > hpairs / 4) nseg = hpairs / 4;            // chains of >= 4 row pairs (+1 halo step); only B = 1 gets there
            if (nseg < 1) nseg = 1;
            av.vpool_nseg = nseg;
            gx = a.B * tiles_per_row * nseg;
        }
        dim3 grid(gx, a.ncam > 0 ? a.ncam : a.C);
        prof_begin(a.fmt == 0 ? "conv1_f16x3_kernel<0>" : "conv1_f16x3_kernel<1>", 2.0 * a.B * a.C * a.Ho * a.Wo * a.Cout * 147.0,
                   (double)a.B * a.C * ((double)a.H * a.W * 3 * (a.fmt == 0 ? 1 : 4) + 4.0 * a.Ho * a.Wo * a.Cout), st);
        static bool attr16 = false;
        if (!attr16) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_f16x3_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, F_SMEM) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_f16x3_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, F_SMEM) != hipSuccess) {
                if (err) *err = "conv1: cannot raise the dynamic LDS limit";
                return -3;
            }
            attr16 = true;
        }
        if (a.fmt == 0) hipLaunchKernelGGL(conv1_f16x3_kernel<0>, grid, dim3(256), F_SMEM, st, av, tiles_per_row, tiles_per_cam);
        else hipLaunchKernelGGL(conv1_f16x3_kernel<1>, grid, dim3(256), F_SMEM, st, av, tiles_per_row, tiles_per_cam);
    } else {
    if (a.vpool) { if (err) *err = "conv1: vpool needs prec f16x3"; return -2; }
    const int tiles_per_row = (a.Wo + TILE_P - 1) / TILE_P;
    const int tiles_per_cam = a.B * a.Ho * tiles_per_row;
    int gx = tiles_per_cam < 512 ? tiles_per_cam : 512;
    // keep tiles-per-block balanced
    const int per = (tiles_per_cam + gx - 1) / gx;
    gx = (tiles_per_cam + per - 1) / per;
    dim3 grid(gx, a.ncam > 0 ? a.ncam : a.C);
    prof_begin(a.fmt == 0 ? "conv1_kernel<0>" : "conv1_kernel<1>", 2.0 * a.B * a.C * a.Ho * a.Wo * a.Cout * 147.0,
               (double)a.B * a.C * ((double)a.H * a.W * 3 * (a.fmt == 0 ? 1 : 4) + 4.0 * a.Ho * a.Wo * a.Cout), st);
    constexpr int smem = (64 * WSTRIDE + 2 * PROWS * PSTRIDE + 3 * 256) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
            if (err) *err = "conv1: cannot raise the dynamic LDS limit";
            return -3;
        }
        attr_set = true;
    }
    if (a.fmt == 0) hipLaunchKernelGGL(conv1_kernel<0>, grid, dim3(256), smem, st, a, tiles_per_row, tiles_per_cam);
    else hipLaunchKernelGGL(conv1_kernel<1>, grid, dim3(256), smem, st, a, tiles_per_row, tiles_per_cam);
    }
    prof_end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { if (err) *err = std::string("conv1 launch: ") + hipGetErrorString(e); return -3; }
    return 0;
}
