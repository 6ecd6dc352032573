// gemm16: the forward GEMM / implicit-GEMM convolution of the inference path on PRE-SPLIT operands (gfx950).
//
// Same arithmetic as gemm.hip's PREC_F16X3 (every fp32 product = hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with fp32
// accumulation, operands split exactly into two fp16 pieces), but BOTH operands already live in memory in the split form
// ("s16": every aligned group of 8 consecutive k of a row is 32 bytes, [8 hi halfs][8 lo halfs] -- the bytes of the fp32
// row, so strides and addressing are those of the fp32 tensor).  Weights are split once at finalize, activations are
// written in this form by the epilogue that produces them (this kernel's own, the direct layer1 convolution, the pool, the
// LayerNorm, the attention output).  Nothing is converted in the main loop and nothing is staged through registers:
//
//   * both operand tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4): one wave instruction moves 8 rows x 128
//     bytes (a row's whole K tile of 32 fp32-k: full 128-byte lines), destination linear, XOR swizzle applied on the SOURCE
//     chunk index and again on the fragment reads (16 lanes of a ds_read_b128 group then hit 16 distinct 16-byte slots);
//   * a ring of NS LDS stages (4 x 32 KB for the 128-row tile, 3 x 48 KB for the 256-row tile), one workgroup of 8 waves
//     per CU; loads run NS-1 tiles ahead across raw s_barriers with counted vmcnt (never drained inside the loop);
//   * the two waves that share a SIMD (wave w and w+4) run half a K tile apart ("ping-pong"): while one issues its MFMAs
//     (and its share of the DMA for a later tile) the other reads its fragments, two barriers per K tile.  The matrix pipe
//     of a SIMD always has one wave feeding it.
//
// Tile: BM x 128 outputs, BM = 128 or 256; wave (g, n) = (w >> 2, w & 3) owns rows [g*BM/2, (g+1)*BM/2) x columns
// [32n, 32n+32): BM/64 MFMA tiles of 32x32, 3 * 2 * BM/64 MFMAs per K tile of 32.
// A forms: plain rows (nn.Linear, MHA projections, FFN, 1x1 input_proj: transformer.py:196-224, detr_vae.py:184) and the NHWC
// implicit im2col of the 3x3 / 1x1 ResNet convolutions with Cin % 32 == 0 (a K tile lies inside one filter tap; padding
// taps read a zero line).  Epilogue: acc * alpha * scale[n] + bias[n] (+ residual: s16 tensor or an f32 table indexed by
// row % res_mod) -> ReLU -> s16 (times the activation scale) or f32 rows, optional row scatter (token layout of input_proj).
#include "common.h"
#include "split16.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

constexpr int BN = 128;
constexpr int ROWB = 128;                    // bytes of one operand row per K tile (32 fp32-k)
constexpr int NTHR = 512;
constexpr int RS = BN + 4;                   // epilogue scratch row stride (floats)

template <int BM> struct Cfg {
    static constexpr int NS = (BM == 256) ? 3 : 4;
    static constexpr int STAGE = (BM + BN) * ROWB;
    static constexpr int GA = BM / 64;           // A-side DMA instructions per wave and K tile
    static constexpr int GB = BN / 64;           // B-side
    static constexpr int G = GA + GB;
    static constexpr int TM = BM / 64;           // 32x32 MFMA tiles per wave (rows)
    static constexpr int SMEM = (NS * STAGE > BM * RS * 4) ? NS * STAGE : BM * RS * 4;
};

template <int N> __device__ __forceinline__ void vmcnt_wait() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// wait until all but the `pending` most recently issued K tiles (G DMA instructions each) of this wave have landed
template <int G> __device__ __forceinline__ void wait_pending(int pending) {
    if (pending >= 3) vmcnt_wait<3 * G>();
    else if (pending == 2) vmcnt_wait<2 * G>();
    else if (pending == 1) vmcnt_wait<G>();
    else vmcnt_wait<0>();
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int OFF> __device__ __forceinline__ u32x4 lds_read16(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

__device__ __forceinline__ void dma16(const void* src, unsigned char* lds_dst) {
    // 64 lanes x 16 bytes: LDS destination = wave-uniform base + lane * 16
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)lds_dst, 16, 0, 0);
}

template <int BM, int CONV>
__global__ __launch_bounds__(NTHR) void gemm16_kernel(Gemm16Args p, int tiles_m, int tiles_n) {
    using C = Cfg<BM>;
    constexpr int NS = C::NS, G = C::G, TM = C::TM;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    // ---- tile id: XCD-aware bijective remap over the flattened (group, tile) space, n fastest (tiles that share A rows
    //      run side by side on one XCD and find them in its L2)
    const int nwg = tiles_m * tiles_n;
    int zz, bid;
    {
        const int total = nwg * (int)gridDim.y;
        const int lin = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = lin & 7, q = total >> 3, r = total & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int flat = base + (lin >> 3);
        zz = flat / nwg;
        bid = flat - zz * nwg;
    }
    const int splitk = p.splitk > 1 ? p.splitk : 1;
    const int g = zz / splitk, split = zz - g * splitk;
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
    const unsigned char* __restrict__ Ab = reinterpret_cast<const unsigned char*>(p.A) + (int64_t)g * p.gA * 4;
    const unsigned char* __restrict__ Bb = reinterpret_cast<const unsigned char*>(p.Bw) + (int64_t)g * p.gB * 4;

    const int nk_total = p.K / 32;
    const int tps = (nk_total + splitk - 1) / splitk;
    const int kt0 = split * tps;
    const int nk = (kt0 + tps < nk_total ? kt0 + tps : nk_total) - kt0;       // >= 1 (host checks)

    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = wv >> 2, wn = wv & 3;
    const int li = lane & 31, lh = lane >> 5;

    // ---- DMA descriptors: instruction q = wv + 8 j of a K tile covers tile rows 8q .. 8q+7 (A rows first, then B rows);
    //      lane -> row 8q + lane/8, physical 16-byte chunk lane%8, which holds LOGICAL chunk (lane%8) ^ ((row/2)%8)
    const unsigned char* a_src[C::GA];
    int a_off[CONV ? C::GA : 1];
    unsigned a_mask[CONV ? C::GA : 1];
    const unsigned char* b_src[C::GB];
    const unsigned char* zero_line = reinterpret_cast<const unsigned char*>(p.zero_page) + (lane & 7) * 16;
#pragma unroll
    for (int j = 0; j < C::GA; ++j) {
        const int row = (wv + 8 * j) * 8 + (lane >> 3);
        const int lch = (lane & 7) ^ ((row >> 1) & 7);
        int m = m0 + row;
        const bool mok = m < p.M;
        m = mok ? m : p.M - 1;
        if (CONV) {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, rem = m - b * hw;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
            // bit (r*KW+s): tap (r,s) of this output pixel lies inside the image
            unsigned mk = 0;
            for (int r = 0; r < p.KH; ++r)
                for (int s = 0; s < p.KW; ++s)
                    if ((unsigned)(hi0 + r) < (unsigned)p.H && (unsigned)(wi0 + s) < (unsigned)p.W) mk |= 1u << (r * p.KW + s);
            a_mask[j] = mk;                    // rows past M replay row M-1: valid memory, never stored
            a_src[j] = Ab + (int64_t)b * p.img_stride * 4;
            a_off[j] = ((hi0 * p.W + wi0) * p.Cin) * 4 + lch * 16;
        } else {
            a_src[j] = Ab + (int64_t)m * p.lda * 4 + lch * 16;
        }
    }
#pragma unroll
    for (int j = 0; j < C::GB; ++j) {
        const int row = (wv + 8 * j) * 8 + (lane >> 3);           // B tile row: q' = wv + 8j of the B region
        const int lch = (lane & 7) ^ ((row >> 1) & 7);
        int n = n0 + row;
        n = n < p.N ? n : p.N - 1;
        b_src[j] = Bb + (int64_t)n * p.ldb * 4 + lch * 16;
    }
    const int tpr = CONV ? p.Cin / 32 : 1;                        // K tiles per filter tap
    auto issue = [&](int kt, int stage) {
        unsigned char* st = smem + stage * C::STAGE;
        const int ktg = kt0 + kt;
        if (CONV) {
            const int rs = ktg / tpr, cb = ktg - rs * tpr;
            const int r = rs / p.KW, s = rs - r * p.KW;
            const int delta = ((r * p.W + s) * p.Cin + cb * 32) * 4;
#pragma unroll
            for (int j = 0; j < C::GA; ++j) {
                const bool inb = (a_mask[j] >> rs) & 1u;
                const unsigned char* src = inb ? a_src[j] + (a_off[j] + delta) : zero_line;
                dma16(src, st + (wv + 8 * j) * 1024);
            }
        } else {
#pragma unroll
            for (int j = 0; j < C::GA; ++j) dma16(a_src[j] + (int64_t)ktg * ROWB, st + (wv + 8 * j) * 1024);
        }
#pragma unroll
        for (int j = 0; j < C::GB; ++j) dma16(b_src[j] + (int64_t)ktg * ROWB, st + BM * ROWB + (wv + 8 * j) * 1024);
    };

    // ---- fragment addresses: lane (i, h) of k step s reads logical chunks 2(2s+h) (hi) and 2(2s+h)+1 (lo) of its row
    int co[2][2];
    {
        const int sw = (li >> 1) & 7;              // tile rows of one lane differ by multiples of 32: same swizzle term
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q = 0; q < 2; ++q) co[s][q] = ((2 * (2 * s + lh) + q) ^ sw) << 4;
    }
    const int a_row_off = (grp * (BM / 2) + li) * ROWB;
    const int b_row_off = BM * ROWB + (wn * 32 + li) * ROWB;

    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // ---- prologue: group 0 has tiles 0 .. NS-2 in flight, group 1 one more (it runs one barrier behind)
    const int npre = NS - 1 + grp;
    int issued = 0;                                   // tiles this wave has issued
    for (int d = 0; d < npre && d < nk; ++d) { issue(d, d % NS); ++issued; }
    wait_pending<G>(issued - 1);                      // tile 0 landed
    __builtin_amdgcn_s_barrier();                     // tile 0 complete in LDS
    if (grp == 1) __builtin_amdgcn_s_barrier();       // stagger: group 1's phases lag group 0's by one barrier

    // The fragment reads are inline asm: the compiler orders every LDS read it can see behind ALL outstanding LDS-DMA
    // (s_waitcnt vmcnt(0) in front of the first ds_read of every K tile), which would drain the ring each step.  The DMA that
    // these reads depend on has been retired by the counted vmcnt + barrier protocol.
    u32x4 fa[2][TM][2], fb[2][2];
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
    // One K tile.  STEADY: the ring is full -- this step issues tile u + NS - 1 + grp, and exactly NS - 2 tiles are in
    // flight behind tile u + 1 at the wait points, so the counts are literals and the body has no branch.
    auto step = [&](int u, auto steady) {
        constexpr bool STEADY = decltype(steady)::value;
        // ---------------- L(u): fragments of tile u
        {
            const unsigned st = lds0 + (unsigned)(u % NS) * C::STAGE;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const unsigned ab = st + a_row_off + co[s][q], bb = st + b_row_off + co[s][q];
                    fb[s][q] = lds_read16<0>(bb);
                    fa[s][0][q] = lds_read16<0>(ab);
                    if (TM > 1) fa[s][1 % TM][q] = lds_read16<32 * ROWB>(ab);
                    if (TM > 2) fa[s][2 % TM][q] = lds_read16<64 * ROWB>(ab);
                    if (TM > 3) fa[s][3 % TM][q] = lds_read16<96 * ROWB>(ab);
                }
            }
        }
        if (grp == 1) {                                      // own share of tile u+1 landed (group 1 waits a phase early)
            if (STEADY) vmcnt_wait<(NS - 2) * G>();
            else if (u + 1 < nk) wait_pending<G>(nk - 2 - u);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---------------- C(u): MFMAs of tile u, with this wave's share of the DMA for tile u + NS - 1 + grp woven between
        //                  them (one DMA instruction per few MFMAs: its issue slot hides under the matrix pipe's busy time)
        __builtin_amdgcn_s_setprio(1);
        if (STEADY) {
            const int x = u + NS - 1 + grp;
            issue(x, x % NS);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const h16x8 yh = __builtin_bit_cast(h16x8, fb[s][0]), yl = __builtin_bit_cast(h16x8, fb[s][1]);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const h16x8 xh = __builtin_bit_cast(h16x8, fa[s][i][0]), xl = __builtin_bit_cast(h16x8, fa[s][i][1]);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, yh, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yl, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yh, acc[i], 0, 0, 0);
            }
        }
        if (STEADY) {
            constexpr int NMF = 6 * TM, PER = NMF / G;
#pragma unroll
            for (int j = 0; j < G; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);       // MFMA
                __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);         // address arithmetic (VALU / SALU)
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);         // one LDS-DMA (VMEM read)
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (grp == 0) {
            if (STEADY) vmcnt_wait<(NS - 2) * G>();
            else if (u + 1 < nk) wait_pending<G>(nk - 2 - u);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    {
        int u = 0;
        const int u_steady = nk - (NS - 1 + grp);              // u < u_steady  <=>  tile u + NS - 1 + grp exists
        for (; u < u_steady; ++u) step(u, std::true_type{});
        for (; u < nk; ++u) step(u, std::false_type{});
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();       // balances group 1's extra barrier
    __builtin_amdgcn_s_barrier();                     // every wave is past its last fragment read: the stages are dead

    // ---- epilogue.  C layout of a 32x32 tile: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5).
    //      The whole BM x 128 tile goes through LDS and leaves as full rows: 16 lanes x 32 bytes = one 512-byte row segment.
    float* scr = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e)
            scr[(grp * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * RS + wn * 32 + li] = acc[i][e];
    __syncthreads();
    const int c8 = t & 15, r0 = t >> 4;
    const int n = n0 + c8 * 8;
    if (n >= p.N) return;
    float alpha = p.alpha != 0.f ? p.alpha : 1.f;
    float sc[8], bi[8];
    {
        const float* scale = p.scale ? p.scale + (int64_t)g * p.gSB : nullptr;
        const float* bias = p.bias ? p.bias + (int64_t)g * p.gSB : nullptr;
        f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0, b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
        if (scale) { s0 = *reinterpret_cast<const f32x4*>(scale + n); s1 = *reinterpret_cast<const f32x4*>(scale + n + 4); }
        if (bias) { b0 = *reinterpret_cast<const f32x4*>(bias + n); b1 = *reinterpret_cast<const f32x4*>(bias + n + 4); }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sc[e] = alpha * s0[e]; sc[4 + e] = alpha * s1[e];
            bi[e] = b0[e]; bi[4 + e] = b1[e];
        }
    }
    const unsigned char* res = p.res ? reinterpret_cast<const unsigned char*>(p.res) + (int64_t)g * p.gRes * 4 : nullptr;
    unsigned char* Cb = reinterpret_cast<unsigned char*>(p.C) + ((int64_t)g * p.gC + (int64_t)split * p.split_stride) * 4;
    const float res_mul = p.res_scale != 0.f ? p.res_scale : 1.f;
    const float out_mul = p.c_scale != 0.f ? p.c_scale : 1.f;
    float vmax = 0.f;
#pragma unroll 2
    for (int pass = 0; pass < BM / 32; ++pass) {
        const int row = pass * 32 + r0;
        const int m = m0 + row;
        if (m >= p.M) break;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(scr + row * RS + c8 * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(scr + row * RS + c8 * 8 + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + bi[e];
        if (res) {
            const int mr = p.res_mod ? m % p.res_mod : m;
            const unsigned char* rp = res + ((int64_t)mr * p.ldres + n) * 4;
            if (p.res_fmt) {
                const h16x8 rh = *reinterpret_cast<const h16x8*>(rp), rl = *reinterpret_cast<const h16x8*>(rp + 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += ((float)rh[e] + (float)rl[e]) * res_mul;
            } else {
                const f32x4 ra = *reinterpret_cast<const f32x4*>(rp), rb = *reinterpret_cast<const f32x4*>(rp + 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += ra[e]; v[4 + e] += rb[e]; }
            }
        }
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        const int64_t orow = p.rowmap ? p.rowmap[m] : m;
        unsigned char* cp = Cb + (orow * p.ldc + n) * 4;
        if (p.c_fmt) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e] *= out_mul; vmax = fmaxf(vmax, fabsf(v[e])); }
            uint2 h0, l0, h1, l1;
            split16(f32x4{v[0], v[1], v[2], v[3]}, h0, l0);
            split16(f32x4{v[4], v[5], v[6], v[7]}, h1, l1);
            *reinterpret_cast<uint4*>(cp) = uint4{h0.x, h0.y, h1.x, h1.y};
            *reinterpret_cast<uint4*>(cp + 16) = uint4{l0.x, l0.y, l1.x, l1.y};
        } else {
            *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(cp + 16) = f32x4{v[4], v[5], v[6], v[7]};
        }
    }
    // range guard of the split form: a stored value beyond the fp16 range (or not finite) raises the handle's flag
    if (p.c_fmt && p.flag && !(vmax < 65504.f)) atomicOr(p.flag, 1u);
}

template <int BM, int CONV>
int launch_t(const Gemm16Args& a, hipStream_t st) {
    using C = Cfg<BM>;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    auto kern = gemm16_kernel<BM, CONV>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int splitk = a.splitk > 1 ? a.splitk : 1;
    const int groups = a.groups > 0 ? a.groups : 1;
    if (prof_enabled()) {
        char nm[128];
        static const bool by_shape = getenv("ACTMI_PROF_SHAPES") && getenv("ACTMI_PROF_SHAPES")[0] == '1';
        if (by_shape)
            snprintf(nm, sizeof(nm), "gemm16_kernel<%d,%d>[M=%d,N=%d,K=%d,g=%d,sk=%d,wgs=%d]", BM, CONV, a.M, a.N, a.K, groups, splitk,
                     tiles_m * tiles_n * groups * splitk);
        else snprintf(nm, sizeof(nm), "gemm16_kernel<%d,%d>", BM, CONV);
        const double abytes = CONV ? (double)(a.M / (a.Ho * a.Wo)) * a.H * a.W * a.Cin : (double)a.M * a.K;
        prof_begin(nm, 2.0 * a.M * a.N * a.K * groups, 4.0 * groups * ((double)a.M * a.N + abytes + (double)a.N * a.K), st);
    }
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n, groups * splitk), dim3(NTHR), C::SMEM, st, a, tiles_m, tiles_n);
    prof_end(st);
    return (int)hipGetLastError();
}

}  // namespace

int gemm16_pick_bm(int M, int N, int groups, int splitk) {
    static const int force = getenv("ACTMI_G16_BM") ? atoi(getenv("ACTMI_G16_BM")) : 0;      // tuning aid
    if (force == 128 || force == 256) return force;
    // one workgroup per CU: a launch takes ceil(tiles / 256) rounds of (rows per tile) work; the 256-row tile moves a
    // third fewer operand bytes per MFMA, so it wins ties
    const long z = (long)(groups > 0 ? groups : 1) * (splitk > 1 ? splitk : 1);
    const long tn = (N + BN - 1) / BN;
    const long t128 = (long)((M + 127) / 128) * tn * z, t256 = (long)((M + 255) / 256) * tn * z;
    const double c128 = (double)((t128 + 255) / 256) * 128 * 1.08, c256 = (double)((t256 + 255) / 256) * 256;
    return c256 <= c128 ? 256 : 128;
}

int launch_gemm16(const Gemm16Args& a_in, hipStream_t st, std::string* err) {
    Gemm16Args a = a_in;
    if (a.groups <= 0) a.groups = 1;
    if (a.M <= 0 || a.N <= 0) return 0;
    auto fail = [&](const char* m) { if (err) *err = std::string("gemm16: ") + m; return -2; };
    if (a.K <= 0 || (a.K & 31)) return fail("K must be a positive multiple of 32");
    if (a.N & 7) return fail("N must be a multiple of 8");
    if (((uintptr_t)a.A & 15) || ((uintptr_t)a.Bw & 15) || ((uintptr_t)a.C & 15)) return fail("A / B / C must be 16-byte aligned");
    if ((a.ldb & 7) || (a.ldc & 7) || (a.gA & 3) || (a.gB & 3) || (a.gC & 3)) return fail("leading dimensions must be multiples of 8 elements");
    if (a.res && ((a.ldres & 7) || ((uintptr_t)a.res & 15) || (a.gRes & 3))) return fail("bad residual");
    if (a.res_mod < 0) return fail("bad res_mod");
    if (a.mode == 0) {
        if (a.lda & 7) return fail("lda must be a multiple of 8");
    } else if (a.mode == 1) {
        if ((a.Cin & 31) || a.K != a.KH * a.KW * a.Cin) return fail("convolution needs Cin % 32 == 0 and K == KH*KW*Cin");
        if (a.KH * a.KW > 32) return fail("at most 32 filter taps");
        if (a.M % (a.Ho * a.Wo)) return fail("M must be images*Ho*Wo");
        if (!a.zero_page) return fail("convolution needs the zero line");
        if ((int64_t)a.H * a.W * a.Cin * 4 >= ((int64_t)1 << 31)) return fail("image too large for 32-bit offsets");
    } else return fail("bad mode");
    const int nk = a.K / 32;
    if (a.splitk > 1) {
        if (a.split_stride == 0 || a.c_fmt != 0 || a.scale || a.bias || a.res || a.relu || a.rowmap)
            return fail("a split contraction writes plain f32 slices (split_stride) and takes no epilogue");
        const int tps = (nk + a.splitk - 1) / a.splitk;
        if ((a.splitk - 1) * tps >= nk) return fail("every split must own at least one K tile");
    }
    const int bm = a.bm ? a.bm : gemm16_pick_bm(a.M, a.N, a.groups, a.splitk);
    int rc;
    if (bm == 256) rc = a.mode ? launch_t<256, 1>(a, st) : launch_t<256, 0>(a, st);
    else if (bm == 128) rc = a.mode ? launch_t<128, 1>(a, st) : launch_t<128, 0>(a, st);
    else return fail("bm must be 0, 128 or 256");
    if (rc != 0 && err) *err = std::string("gemm16 launch: ") + hipGetErrorString((hipError_t)rc);
    return rc == 0 ? 0 : -3;
}
