// fp32 MFMA GEMM / implicit-GEMM convolution for gfx950 (v_mfma_f32_32x32x2_f32, exact f32).
//
// One kernel serves every dense contraction of the ACT path: nn.Linear, the packed MHA in/out projections,
// the FFN, the 1x1 input_proj and the 3x3 / 1x1-stride-2 ResNet convolutions as NHWC implicit im2col
// (reference: detr/models/transformer.py:196-224, detr_vae.py:57-61,184; torchvision BasicBlock).
//
// Tiling (CDNA4, 64-wide waves): 256 threads = 4 waves, one per SIMD.  Block tile BMxBN, wave tile WMxWN made
// of 32x32 MFMA tiles.  K is consumed 32 at a time through a double-buffered LDS stage.  Both operands are
// "row-major with K contiguous" (A[m][k], W[n][k]); a 16-byte global chunk (4 consecutive k) is stored as
// one float4 in LDS plane p = chunk index, row r:  lds[p][r].  The f32 MFMA takes ONE k per lane-half, and
// the sum over k is order-free, so lane (i = lane&31, h = lane>>5) reads the float4 at plane 2*kb+h, row i
// and feeds its 4 components to 4 consecutive MFMAs: A and B use the same (h, j) -> k assignment, so the
// contraction is complete and no LDS transpose or shuffle is needed.  One ds_read_b128 per operand tile
// feeds 4 MFMAs (256 SIMD cycles): LDS bandwidth is irrelevant, the kernel is MFMA-issue bound.
// Planes are padded by one float4 so that the 8 lanes that write one row's 8 chunks hit 8 different
// 16-byte bank groups (ds_write_b128 is serviced 8 lanes at a time).
#include "common.h"
#include <cstdio>

namespace {

constexpr int BK = 32;
constexpr int NPL = BK / 4;

template <int BM, int BN>
constexpr int stage_f4() { return NPL * ((BM + 1) + (BN + 1)); }

template <int BM, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p, int tiles_m, int tiles_n) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int PSA = BM + 1, PSB = BN + 1;
    constexpr int STAGE = stage_f4<BM, BN>();
    constexpr int NLA = BM / 32, NLB = BN / 32;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f32x4* smem = reinterpret_cast<f32x4*>(smem_raw);

    // XCD-aware bijective remap: hardware deals consecutive block ids round-robin over the 8 XCDs; give each
    // XCD a contiguous run of tile ids so that neighbouring tiles (same A rows) share one L2.
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        bid = base + (bid >> 3);
    }
    const int m0 = (bid / tiles_n) * BM;
    const int n0 = (bid % tiles_n) * BN;
    const int g = blockIdx.z;

    const float* __restrict__ A = p.A + (int64_t)g * p.gA;
    const float* __restrict__ Bw = p.Bw + (int64_t)g * p.gB;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wrow0 = (wave / WAVES_N) * WM;
    const int wcol0 = (wave % WAVES_N) * WN;
    const int cidx = t & 7;          // which 16-byte chunk of the 32-wide k tile this thread stages
    const int srow = t >> 3;         // staging row (plus 32*i)

    // ---- per-thread row descriptors (constant over the K loop)
    const float* a_ptr[NLA];
    const float* add_ptr[NLA];
    int a_hi0[NLA], a_wi0[NLA];
    bool a_ok[NLA];
    const bool use_add = (MODE == 0) && p.A_add != nullptr && n0 < p.add_ncols;
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
        const int m = m0 + srow + 32 * i;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        if (MODE == 0) {
            a_ptr[i] = A + (int64_t)mm * p.lda;
            add_ptr[i] = use_add ? p.A_add + (int64_t)(mm % p.add_mod) * p.ld_add : nullptr;
            a_hi0[i] = a_wi0[i] = 0;
        } else {
            const int hw = p.Ho * p.Wo;
            const int b = mm / hw, rem = mm - b * hw;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            a_ptr[i] = A + (int64_t)b * p.img_stride;
            add_ptr[i] = nullptr;
            a_hi0[i] = ho * p.stride - p.pad;
            a_wi0[i] = wo * p.stride - p.pad;
        }
    }
    const float* b_ptr[NLB];
    bool b_ok[NLB];
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
        const int n = n0 + srow + 32 * i;
        b_ok[i] = n < p.N;
        b_ptr[i] = Bw + (int64_t)(b_ok[i] ? n : 0) * p.ldb;
    }

    f32x4 ra[NLA], rb[NLB];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto load_tile = [&](int kt) {
        const int k = kt * BK + cidx * 4;
        const bool kok = k < p.K;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < NLA; ++i) {
                f32x4 v = zero4;
                if (a_ok[i] && kok) {
                    v = *reinterpret_cast<const f32x4*>(a_ptr[i] + k);
                    if (use_add) v += *reinterpret_cast<const f32x4*>(add_ptr[i] + k);
                }
                ra[i] = v;
            }
        } else {
            const int rs = k / p.Cin;
            const int c = k - rs * p.Cin;
            const int r = rs / p.KW;
            const int s = rs - r * p.KW;
#pragma unroll
            for (int i = 0; i < NLA; ++i) {
                const int hi = a_hi0[i] + r, wi = a_wi0[i] + s;
                f32x4 v = zero4;
                if (a_ok[i] && kok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                    v = *reinterpret_cast<const f32x4*>(a_ptr[i] + ((int64_t)hi * p.W + wi) * p.Cin + c);
                ra[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            f32x4 v = zero4;
            if (b_ok[i] && kok) v = *reinterpret_cast<const f32x4*>(b_ptr[i] + k);
            rb[i] = v;
        }
    };
    auto store_tile = [&](int stage) {
        f32x4* sa = smem + stage * STAGE;
        f32x4* sb = sa + NPL * PSA;
#pragma unroll
        for (int i = 0; i < NLA; ++i) sa[cidx * PSA + srow + 32 * i] = ra[i];
#pragma unroll
        for (int i = 0; i < NLB; ++i) sb[cidx * PSB + srow + 32 * i] = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) load_tile(kt + 1);
        const f32x4* sa = smem + cur * STAGE;
        const f32x4* sb = sa + NPL * PSA;
#pragma unroll
        for (int kb = 0; kb < NPL / 2; ++kb) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = sa[(2 * kb + lh) * PSA + wrow0 + i * 32 + li];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = sb[(2 * kb + lh) * PSB + wcol0 + j * 32 + li];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const float* scale = p.scale ? p.scale + (int64_t)g * p.gSB : nullptr;
    const float* bias = p.bias ? p.bias + (int64_t)g * p.gSB : nullptr;
    const float* res = p.res ? p.res + (int64_t)g * p.gRes : nullptr;
    float* C = p.C + (int64_t)g * p.gC;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wcol0 + j * 32 + li;
        const bool nok = n < p.N;
        const float sc = (scale && nok) ? scale[n] : 1.f;
        const float bi = (bias && nok) ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wrow0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (nok && m < p.M) {
                    float v = acc[i][j][e];
                    v = scale ? v * sc + bi : v + bi;
                    if (res) {
                        const int rm = p.res_mod ? (m % p.res_mod) : m;
                        v += res[(int64_t)rm * p.ldres + n];
                    }
                    if (p.relu) v = fmaxf(v, 0.f);
                    const int64_t orow = p.rowmap ? p.rowmap[m] : m;
                    C[orow * p.ldc + n] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int MODE>
int launch_cfg(const GemmArgs& a, hipStream_t st) {
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    constexpr int smem = 2 * stage_f4<BM, BN>() * 16;
    auto kern = gemm_f32_kernel<BM, BN, WM, WN, MODE>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid(tiles_m * tiles_n, 1, a.groups > 0 ? a.groups : 1);
    if (prof_enabled()) {
        char nm[64];
        snprintf(nm, sizeof(nm), "gemm_f32_kernel<%d,%d,%d,%d,%d>", BM, BN, WM, WN, MODE);
        const double g = a.groups;
        // algorithmic work: 2*M*N*K flops; operands read once + result written once
        prof_begin(nm, 2.0 * a.M * a.N * a.K * g, 4.0 * g * ((double)a.M * a.N + (double)a.N * a.K +
                   (MODE == 0 ? (double)a.M * a.K : (double)(a.M / (a.Ho * a.Wo)) * a.H * a.W * a.Cin)), st);
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, a, tiles_m, tiles_n);
    prof_end(st);
    return (int)hipGetLastError();
}

double tile_eff(int M, int N, int groups, int BM, int BN, double factor) {
    const long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN) * groups;
    const long rounds = (tiles + 255) / 256;
    return factor * ((double)M * N * groups) / ((double)rounds * 256 * BM * BN);
}

}  // namespace

int launch_gemm(const GemmArgs& a_in, hipStream_t st, std::string* err) {
    GemmArgs a = a_in;
    if (a.groups <= 0) a.groups = 1;
    if (a.M <= 0 || a.N <= 0) return 0;
    auto fail = [&](const char* m) { if (err) *err = std::string("gemm: ") + m; return -2; };
    if (a.K <= 0 || (a.K & 3)) return fail("K must be a positive multiple of 4");
    if (a.ldb & 3) return fail("ldb must be a multiple of 4");
    if (((uintptr_t)a.A & 15) || ((uintptr_t)a.Bw & 15)) return fail("A/B must be 16-byte aligned");
    if (a.mode == 0) {
        if (a.lda & 3) return fail("lda must be a multiple of 4");
        if (a.A_add && ((a.ld_add & 3) || a.add_mod <= 0 || ((uintptr_t)a.A_add & 15))) return fail("bad addend");
    } else {
        if (a.Cin & 3) return fail("Cin must be a multiple of 4");
        if (a.K != a.KH * a.KW * a.Cin) return fail("K != KH*KW*Cin");
        if (a.A_add) return fail("addend not supported in conv mode");
        if (a.M % (a.Ho * a.Wo)) return fail("M must be images*Ho*Wo");
    }
    const double eL = tile_eff(a.M, a.N, a.groups, 128, 128, 1.00);
    const double eM = tile_eff(a.M, a.N, a.groups, 128, 64, 0.95);
    const double eS = tile_eff(a.M, a.N, a.groups, 64, 64, 0.88);
    int rc;
    if (a.mode == 0) {
        if (eL >= eM && eL >= eS) rc = launch_cfg<128, 128, 64, 64, 0>(a, st);
        else if (eM >= eS) rc = launch_cfg<128, 64, 64, 32, 0>(a, st);
        else rc = launch_cfg<64, 64, 32, 32, 0>(a, st);
    } else {
        if (eL >= eM && eL >= eS) rc = launch_cfg<128, 128, 64, 64, 1>(a, st);
        else if (eM >= eS) rc = launch_cfg<128, 64, 64, 32, 1>(a, st);
        else rc = launch_cfg<64, 64, 32, 32, 1>(a, st);
    }
    if (rc != 0 && err) *err = std::string("gemm launch: ") + hipGetErrorString((hipError_t)rc);
    return rc == 0 ? 0 : -3;
}
