// fp32 MFMA GEMM / implicit-GEMM convolution for gfx950 (v_mfma_f32_32x32x2_f32, exact f32), forward and backward.
//
// One kernel family serves every dense contraction of the ACT path: nn.Linear, the packed MHA in/out projections,
// the FFN, the 1x1 input_proj and the 3x3 / 1x1-stride-2 ResNet convolutions as NHWC implicit im2col
// (reference: detr/models/transformer.py:196-224, detr_vae.py:57-61,184; torchvision BasicBlock), plus their
// gradients: dX = dY W (B stored [contraction][out]), dW = dY^T X (both operands stored [contraction][out]), the
// convolution data gradient (gather over (r,s,n)) and weight gradient (gather of X, split-K with float atomics),
// and the batched products of the attention backward.
//
// Tiling (CDNA4, 64-wide waves): 256 threads = 4 waves, one per SIMD.  Block tile BMxBN, wave tile WMxWN made
// of 32x32 MFMA tiles.  The contraction is consumed 32 at a time through a double-buffered LDS stage.  In LDS both
// operands live as float4 "planes": plane p holds, for every out-row r, the 4 contraction values 4p..4p+3.
// The f32 MFMA takes ONE k per lane-half and the sum over k is order-free, so lane (i = lane&31, h = lane>>5) reads
// the float4 at plane 2*kb+h, row i and feeds its 4 components to 4 consecutive MFMAs: A and B use the same
// (h, j) -> k assignment, so the contraction is complete and no LDS transpose or shuffle is needed.  One
// ds_read_b128 per operand tile feeds 4 MFMAs (256 SIMD cycles): LDS bandwidth is irrelevant, the kernel is
// MFMA-issue bound.  Operands stored contraction-contiguous are staged with one 16-byte global load per float4;
// operands stored out-contiguous ([contraction][out]) are staged as 4x4 micro-tiles transposed in registers.
// Planes are padded by one float4 so that the lanes writing one row's chunks hit different bank groups.
#include "common.h"
#include "dropout.h"
#include "split16.h"
#include <cstdio>
#include <string>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int BK = 32;
constexpr int NPL = BK / 4;

enum { A_N = 0, A_CONV = 1, A_DGRAD = 2, A_T = 3, A_NADD = 4 };   // A_NADD: A_N with the broadcast addend
enum { B_N = 0, B_T = 1, B_WGRAD = 2 };

// PREC: how an fp32 product is formed.
//   PREC_F32   v_mfma_f32_32x32x2_f32, the native fp32 matrix instruction (157 TFLOP/s dense on MI355X)
//   PREC_F16X3 each fp32 operand x is split EXACTLY into two fp16 pieces, hi = rn16(x), lo = rn16(x - hi) (together 22-23
//              significand bits; fp16 subnormals are kept by gfx950's conversions and MFMA), and a*b is taken as
//              hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation.  The dropped lo*lo term is
//              2^-22 relative -- below the fp32 accumulation noise of a K >= 64 dot product -- so results are fp32-grade
//              (measured: same |a_hat - reference| as PREC_F32), while the fp16 pipe is 16x wider: 3 products still
//              leave 5.3x the native fp32 MFMA rate.  Operands must be finite and |x| < 65504.
//   PREC_BF16  ONE bf16 product per fp32 product on v_mfma_f32_32x32x16_bf16 (operands rounded to 8 significand bits, fp32
//              accumulation): the opt-in "speed mode" of the TRAINING step (BASELINE config 3 names bf16; fp32 master weights and
//              optimizer state).  A third of the MFMAs and half the LDS traffic of PREC_F16X3, results at ~1e-2 relative: never the
//              default, never the inference path (the 1e-4 parity bar rules it out -- DESIGN.md section 4).
enum { PREC_F32 = 0, PREC_F16X3 = 1, PREC_BF16 = 2 };
constexpr int ACTMI_PREC_DEFAULT_IS = ACTMI_PREC_F32;      // library default when neither descriptor nor environment says

#ifndef ACTMI_LDS_PAD
#define ACTMI_LDS_PAD 1
#endif
constexpr int LDS_PAD = ACTMI_LDS_PAD;      // plane stride = rows + LDS_PAD 16-byte units

template <int BM, int BN, int PREC>
constexpr int stage_f4() { return NPL * ((BM + LDS_PAD) + (BN + LDS_PAD)); }

typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 4 floats -> 4 bf16 (round to nearest even; v_cvt_pk_bf16_f32 keeps a NaN a NaN), packed in 8 bytes
__device__ __forceinline__ uint2 pack_bf16x4(const f32x4 v) {
    const bf16x2 a = {(__bf16)v[0], (__bf16)v[1]}, b = {(__bf16)v[2], (__bf16)v[3]};
    return uint2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
}

// Row swizzle of the transposed-staging forms.  A thread of those loaders holds 4 CONSECUTIVE out-rows of one k group, so
// the 8/16 lanes of an LDS store group would hit rows 64 bytes apart -- 2 distinct bank slots, a 4- to 8-way conflict.
// XOR-ing the two low row bits with bits 3-4 spreads a group over all 8 slots; the fragment reads apply the same
// permutation (a read group covers 32 consecutive rows either way, so reads stay conflict-free).
__device__ __forceinline__ int swz_row(int r) { return r ^ ((r >> 3) & 3); }

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// UNMASKED (f16x3, forward operand forms only): the loop flavour without zero-fill selects and operand pre-scales is
// its own instantiation -- as a block-uniform branch inside one kernel both flavours shared one register allocation and the
// hot one spilled (A_N: 184 bytes of scratch per lane, A_NADD: 104)
// XS (A_CONV only): the contraction continues past the filter taps into a SECOND source tensor -- a 1x1 / stride_x convolution
// of Ax (NHWC, Cx channels) whose weights are the columns k >= kx_begin of the same B rows.  This is how the downsample branch
// of a ResNet block (1x1 / s2 convolution + FrozenBN of the block input, torchvision BasicBlock as used at backbone.py:66-71)
// rides inside the block's second 3x3 convolution: out = relu(W2' * y1 + Wd' * x_strided + (b2 + bd)) with the FrozenBN scales
// folded into W2' / Wd' -- one launch and no residual round trip instead of three launches.
template <int BM, int BN, int WM, int WN, int AMODE, int BMODE, int PREC, int BSPLIT, int UNMASKED, int XS = 0>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, (PREC && BM * BN <= 128 * 64 && (AMODE == 0 /*A_N*/ || AMODE == 3 /*A_T*/) && !(AMODE == 0 && BMODE == 1)) ? 3 : 2)
    void gemm_f32_kernel(GemmArgs p, int tiles_m, int tiles_n) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;      // 256 threads (4 waves) or 512 (8 waves, 2 per SIMD) for the 256x128 tile
    constexpr int RPP = NT / 8;                          // rows staged per pass of the N-form loaders
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    // LDS plane stride in 16-byte units.  PREC_F16X3 re-uses the 8-plane stage: plane 2*kg holds the hi halves of
    // contraction group kg (8 consecutive k), plane 2*kg+1 the lo halves.  LDS stores bank on (addr/4) % 32 in groups of
    // 16 consecutive lanes: with a stride of 1 mod 4 units the 8-byte staging stores of a group (2 rows x 4 groups x 2
    // halves) fill the 128-byte window exactly once (a stride of 2 mod 8 measured 4 % slower: 2-way conflicts).
    constexpr int PSA = BM + LDS_PAD, PSB = BN + LDS_PAD;
    constexpr int STAGE = stage_f4<BM, BN, PREC>();
    constexpr int NLA = (AMODE == A_T) ? 4 : BM / RPP;
    constexpr int NLB = (BMODE != B_N) ? 4 : BN / RPP;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the rows staged per pass");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f32x4* smem = reinterpret_cast<f32x4*>(smem_raw);

    // XCD-aware bijective remap: hardware deals consecutive block ids round-robin over the 8 XCDs; give each
    // XCD a contiguous run of tile ids so that neighbouring tiles (same A rows) share one L2.
    const int nwg = tiles_m * tiles_n;
    const int splitk = p.splitk > 1 ? p.splitk : 1;
    int bid, zz;
    {
        // the remap runs over the FLATTENED (group, split, tile) space: an XCD then works through one group's (and one
        // K slice's) tiles at a time (for the per-camera convolutions: one camera's weights per XCD instead of a slice of
        // every camera at once -- layer3/4 fetched 10-12x their operand bytes through the fabric before this; a split-K
        // launch whose planes were dealt over all 8 XCDs fetched every K slice's operands into every L2: 670 MB
        // instead of 210 MB for the layer4 convolutions at splitk = 4, PMC round 2)
        const int total = nwg * (int)gridDim.z;
        const int lin = blockIdx.z * gridDim.x + blockIdx.x;
        const int xcd = lin & 7, q = total >> 3, r = total & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int flat = base + (lin >> 3);
        zz = flat / nwg;
        bid = flat - zz * nwg;
    }
    // grouped rasterisation: walk the tile grid in bands of GROUP_M tile rows, column by column, so that the ~64 tiles
    // an XCD runs concurrently form a compact patch (8 A panels x 8 B panels) and re-use each other's operand panels
    // in that XCD's 4 MB L2.  fp32 operands make the kernel L2/fabric-bandwidth sensitive (32 FLOP per loaded byte at
    // a 128x128 tile), so the hit rate matters as much as the MFMA schedule.
    constexpr int GROUP_M = 8;
    const int width = GROUP_M * tiles_n;
    const int group_id = bid / width;
    const int first_m = group_id * GROUP_M;
    const int gsz = (tiles_m - first_m < GROUP_M) ? tiles_m - first_m : GROUP_M;
    const int in_group = bid - group_id * width;
    const int m0 = (first_m + in_group % gsz) * BM;
    const int n0 = (in_group / gsz) * BN;
    const int split = zz % splitk;
    const int g = zz / splitk;
    int64_t offA, offB, offC, offRes;
    if (p.groups_inner > 0) {
        const int g1 = g / p.groups_inner, g2 = g % p.groups_inner;
        offA = g1 * p.gA + g2 * p.gA2; offB = g1 * p.gB + g2 * p.gB2;
        offC = g1 * p.gC + g2 * p.gC2; offRes = g1 * p.gRes + g2 * p.gRes2;
    } else {
        offA = (int64_t)g * p.gA; offB = (int64_t)g * p.gB; offC = (int64_t)g * p.gC; offRes = (int64_t)g * p.gRes;
    }
    // A_alt (row-major forms): the column blocks n0 < alt_ncols read their rows from a second matrix of the same shape (the
    // packed QKV product: q and k project x + pos, v projects x -- transformer.py:216-217 -- with x + pos written by the
    // LayerNorm that produced x); block-uniform, costs nothing in the loop
    const float* __restrict__ A = ((AMODE == A_N) && p.A_alt != nullptr && n0 < p.alt_ncols ? p.A_alt : p.A) + offA;
    const float* __restrict__ Bw = p.Bw + offB;

    uint64_t* stamp = p.stamps ? p.stamps + ((int64_t)blockIdx.z * gridDim.x + blockIdx.x) * 4 : nullptr;
    if (stamp && threadIdx.x == 0) {
        stamp[0] = __builtin_amdgcn_s_memtime();
        // diagnostic: which CU runs this block (HW_ID: cu [11:8], sh [12], se [15:13]; XCC_ID [3:0])
        const unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
        const unsigned xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (31 << 11));
        stamp[1] = ((uint64_t)xcc << 32) | hw;
    }
    unsigned* const amax_out = UNMASKED ? nullptr : p.amax_out;
    unsigned* const fflag = UNMASKED ? nullptr : p.finite_flag;
    const bool track = amax_out != nullptr || fflag != nullptr;
    const unsigned amax_seen = amax_out ? __hip_atomic_load(amax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    const int nk_total = (p.K + BK - 1) / BK;
    const int tps = (nk_total + splitk - 1) / splitk;
    const int kt_begin = split * tps;
    const int kt_end = (kt_begin + tps < nk_total) ? kt_begin + tps : nk_total;
    if (kt_begin >= kt_end) return;      // block-uniform
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wrow0 = (wave / WAVES_N) * WM;
    const int wcol0 = (wave % WAVES_N) * WN;
    int arow[TM], bcol[TN];          // LDS row of this lane's fragment per MFMA tile (swizzled for the transposed forms)
#pragma unroll
    for (int i = 0; i < TM; ++i) arow[i] = (AMODE == A_T) ? swz_row(wrow0 + i * 32 + li) : wrow0 + i * 32 + li;
#pragma unroll
    for (int j = 0; j < TN; ++j) bcol[j] = (BMODE != B_N) ? swz_row(wcol0 + j * 32 + li) : wcol0 + j * 32 + li;
    const int cidx = t & 7;          // N-form staging: which 16-byte chunk of the 32-wide k tile
    const int srow = t >> 3;         // N-form staging row (plus RPP*i)

    // ------------------------------------------------------------------ A operand descriptors
    const float* a_ptr[NLA];
    const float* add_ptr[NLA];
    int a_hi0[NLA], a_wi0[NLA];
    bool a_ok[NLA];
    // A_CONV fast path (block-uniform): when Cin is a multiple of the K tile, a K tile lies inside ONE filter tap, so the
    // tap (r,s) and its channel base are scalars per tile and the per-row work shrinks to a mask-bit test and one add
    // (the general path decomposes k per thread with two integer divisions per K tile).
    // CF: the hot (unmasked) convolution flavour ASSUMES the uniform-tap fast path (the host only selects it then, launch_cfg_b):
    // the general im2col decode and its block-uniform branches are not even compiled into it, so the K step stays ONE basic
    // block and the scheduler can weave the next tiles' loads between the MFMAs.  In this flavour all operand addresses are a
    // uniform base + an unsigned 32-bit byte offset per lane (global_load with an SGPR base: no 64-bit VALU adds per load), and
    // the filter tap of a K tile is tracked incrementally in scalar registers (the reciprocal-multiply decode cost ~15 VALU
    // per tile on uniform values).  Measured on the round-2 build: 108 VALU per K tile and wave in this loop against 24 MFMAs.
    // MEASURED SLOWER (round 3, same box, profiles/r03_slim_loader_ab.json): the convolution launches 155.6 vs 151.4 us, and the
    // row-major flavour below (SA) 92.5 vs 81.7 us although their loops shrank from 183 to 71 and from 45 to 37 vector
    // instructions per K tile -- the loop is not issue-bound, and loads addressed from an SGPR base behind a readfirstlane
    // issue later than the 64-bit VGPR-addressed ones the compiler hoists to the top of the step.  Kept compiled out
    // (ACTMI_GEMM_SLIM=1 builds it) as the record of that experiment.
#ifndef ACTMI_GEMM_SLIM
#define ACTMI_GEMM_SLIM 0
#endif
    constexpr bool CF = ACTMI_GEMM_SLIM != 0 && AMODE == A_CONV && UNMASKED != 0;
    // SA: the same addressing (uniform base + unsigned 32-bit byte offset per lane) for the row-major hot flavours
    constexpr bool SA = ACTMI_GEMM_SLIM != 0 && (AMODE == A_N || AMODE == A_NADD) && BMODE == B_N && UNMASKED != 0;
    const bool conv_fast = CF || (AMODE == A_CONV && (p.Cin % BK) == 0 && p.KH * p.KW <= 32 &&
                                  (int64_t)p.H * p.W * p.Cin < ((int64_t)1 << 30));
    unsigned sa_offb[SA ? NLA : 1], sa_addb[(SA && AMODE == A_NADD) ? NLA : 1];
    unsigned a_tapmask[AMODE == A_CONV ? NLA : 1];
    int a_off0[AMODE == A_CONV ? NLA : 1];
    int a_offx[XS ? NLA : 1];                   // XS: this row's pixel of the second source (floats from Axg), + cidx * 4
    const float* __restrict__ Axg = XS ? p.Ax + (int64_t)g * p.gAx : nullptr;
    const int ktx = XS ? p.kx_begin / BK : 0x7fffffff;      // first K tile of the second source
    const float conv_inv_tpr = (AMODE == A_CONV && conv_fast) ? 1.0f / (float)(p.Cin / BK) : 0.f;
    const float conv_inv_kw = (AMODE == A_CONV || AMODE == A_DGRAD) ? 1.0f / (float)p.KW : 0.f;
    const float conv_inv_ntap = (AMODE == A_CONV) ? 1.0f / (float)(p.KH * p.KW) : 0.f;
    const int dg_cq = (AMODE == A_DGRAD) ? p.K / (p.KH * p.KW) : BK;
    const bool dgrad_fast = AMODE == A_DGRAD && (dg_cq % BK) == 0 && (p.stride == 1 || p.stride == 2) &&
                            (int64_t)p.Ho * p.Wo * dg_cq < ((int64_t)1 << 30);
    const float dg_inv_tpr = (AMODE == A_DGRAD) ? 1.0f / (float)(dg_cq / BK > 0 ? dg_cq / BK : 1) : 0.f;
    unsigned conv_rep = 0;                      // bit r*KW set for every filter row: replicates a column mask over the rows
    if (AMODE == A_CONV && conv_fast)
        for (int r = 0; r < p.KH; ++r) conv_rep |= 1u << (r * p.KW);
    const bool use_add = (AMODE == A_NADD) && n0 < p.add_ncols;       // block-uniform
    // T-form micro tile of A: out group og (4 consecutive m), k group kg (4 consecutive k)
    const int a_og = t % (BM / 4), a_kg = t / (BM / 4);
    if (AMODE == A_T) {
        // nothing per row: addresses are formed per k row in the loader
    } else {
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            const int m = m0 + srow + RPP * i;
            a_ok[i] = m < p.M;
            const int mm = a_ok[i] ? m : 0;
            add_ptr[i] = nullptr;
            a_hi0[i] = a_wi0[i] = 0;
            if (AMODE == A_N || AMODE == A_NADD) {
                const int64_t ar = p.a_rowmap ? p.a_rowmap[mm] : mm;
                a_ptr[i] = A + ar * p.lda;
                if (AMODE == A_NADD) add_ptr[i] = p.A_add + (int64_t)(mm % p.add_mod) * p.ld_add;
                if (SA) {
                    sa_offb[i] = ((unsigned)ar * (unsigned)p.lda + (unsigned)(cidx * 4)) * 4u;
                    if (AMODE == A_NADD) sa_addb[i] = ((unsigned)(mm % p.add_mod) * (unsigned)p.ld_add + (unsigned)(cidx * 4)) * 4u;
                }
            } else if (AMODE == A_CONV) {
                const int hw = p.Ho * p.Wo;
                const int b = mm / hw, rem = mm - b * hw;
                const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                a_ptr[i] = A + (int64_t)b * p.img_stride;
                a_hi0[i] = ho * p.stride - p.pad;
                a_wi0[i] = wo * p.stride - p.pad;
                if (conv_fast) {
                    // bit (r*KW+s): filter tap (r,s) of this output pixel lies inside the image
                    // (closed form, no loops: valid filter rows [r_lo, r_hi) x valid filter columns [q_lo, q_hi))
                    const int r_lo = a_hi0[i] < 0 ? -a_hi0[i] : 0, q_lo = a_wi0[i] < 0 ? -a_wi0[i] : 0;
                    int r_hi = p.H - a_hi0[i], q_hi = p.W - a_wi0[i];
                    r_hi = r_hi < p.KH ? (r_hi > 0 ? r_hi : 0) : p.KH;
                    q_hi = q_hi < p.KW ? (q_hi > 0 ? q_hi : 0) : p.KW;
                    const unsigned colmask = (r_lo < r_hi && q_lo < q_hi) ? (((1u << q_hi) - 1u) & ~((1u << q_lo) - 1u)) : 0u;
                    const unsigned rowsel = (unsigned)((((uint64_t)1 << (r_hi * p.KW)) - 1u) & ~(((uint64_t)1 << (r_lo * p.KW)) - 1u));
                    a_tapmask[i] = a_ok[i] ? (rowsel & (colmask * conv_rep)) : 0u;
                    a_off0[i] = (a_hi0[i] * p.W + a_wi0[i]) * p.Cin + cidx * 4;
                    if (CF) a_off0[i] += b * (int)p.img_stride;          // CF: offsets from the group's base, image included
                    if (XS) a_offx[i] = ((b * p.Hx + ho * p.stride_x) * p.Wx + wo * p.stride_x) * p.Cx + cidx * 4;
                }
            } else {   // A_DGRAD: rows are input pixels (b, hi, wi)
                const int hw = p.H * p.W;
                const int b = mm / hw, rem = mm - b * hw;
                const int hi = rem / p.W, wi = rem - hi * p.W;
                a_ptr[i] = A + (int64_t)b * p.img_stride;
                a_hi0[i] = hi + p.pad;
                a_wi0[i] = wi + p.pad;
            }
        }
    }
    // ------------------------------------------------------------------ B operand descriptors
    const float* b_ptr[NLB];
    bool b_ok[NLB];
    const int b_og = t % (BN / 4), b_kg = t / (BN / 4);
    int wg_r = 0, wg_s = 0, wg_c = 0;        // B_WGRAD: this thread's (r, s, c0) of its 4 out columns
    const float wg_inv_hw = (BMODE == B_WGRAD) ? 1.0f / (float)(p.Ho * p.Wo) : 0.f;
    const float wg_inv_wo = (BMODE == B_WGRAD) ? 1.0f / (float)p.Wo : 0.f;
    unsigned b_offb[(CF || SA) ? NLB : 1];   // CF / SA: byte offset of this lane's row (and 16-byte chunk) from Bw
    if (BMODE == B_N) {
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            const int n = n0 + srow + RPP * i;
            b_ok[i] = n < p.N;
            b_ptr[i] = Bw + (int64_t)(b_ok[i] ? n : 0) * p.ldb;
            if (CF || SA) b_offb[i] = ((unsigned)(b_ok[i] ? n : 0) * (unsigned)p.ldb + (unsigned)(cidx * 4)) * 4u;
        }
    } else if (BMODE == B_WGRAD) {
        const int j0 = n0 + b_og * 4;
        const int rs = j0 / p.Cin;
        wg_c = j0 - rs * p.Cin;
        wg_r = rs / p.KW;
        wg_s = rs - wg_r * p.KW;
    }

    // staging registers of the K tile in flight
    struct Regs {
        f32x4 ra[NLA], rb[NLB], rx[AMODE == A_NADD ? NLA : 1];
        bool ra_ok[NLA], rb_ok[NLB];
    };
    Regs R0;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // CF: the tile the loader was last asked for and its filter tap, kept in scalar registers.  The loader is called with
    // non-decreasing tile numbers that grow by at most one (tail requests are clamped to the last tile).
    int cf_kt = 0, cf_rs = 0, cf_r = 0, cf_q = 0, cf_cb = 0;
    if (CF) {
        const int tpr = p.Cin / BK;
        cf_kt = kt_begin;
        cf_rs = cf_kt / tpr;
        cf_cb = (cf_kt - cf_rs * tpr) * BK;
        cf_r = cf_rs / p.KW;
        cf_q = cf_rs - cf_r * p.KW;
    }
    auto ldb4 = [](const float* base, unsigned offb) { return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + offb); };
    // a block-uniform pointer, read through readfirstlane so that it provably lives in SGPRs (loads then take the saddr form)
    auto ubase = [](const float* q) {
        const uint64_t u = reinterpret_cast<uint64_t>(q);
        return reinterpret_cast<const float*>(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                                              (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u));
    };
    // transposed-storage operands ([contraction][out]): a tile that lies inside the matrix, on a K tile that lies inside the
    // contraction, is four plain 16-byte loads per thread from uniform row bases -- the general form below (per-element
    // bounds, scalar tails: ~200 branches in the compiled loop) only runs on the matrix edges
    const bool at_interior = AMODE == A_T && m0 + BM <= p.M;
    const bool bt_interior = BMODE == B_T && n0 + BN <= p.N && p.B_add == nullptr;
    const unsigned at_offb = AMODE == A_T ? ((unsigned)(a_kg * 4) * (unsigned)p.lda + (unsigned)(m0 + a_og * 4)) * 4u : 0u;
    const unsigned bt_offb = BMODE == B_T ? ((unsigned)(b_kg * 4) * (unsigned)p.ldb + (unsigned)(n0 + b_og * 4)) * 4u : 0u;

    auto load_tile = [&](int kt, auto& R) {
        // ---------------- A
        if (AMODE == A_T) {
            if (a_kg < NPL && at_interior && (kt + 1) * BK <= p.K) {
                const float* __restrict__ rb = ubase(A + (int64_t)kt * BK * p.lda);
#pragma unroll
                for (int j = 0; j < 4; ++j) R.ra[j] = ldb4(rb + (int64_t)j * p.lda, at_offb);
            } else if (a_kg < NPL) {
                const int o = m0 + a_og * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kk = kt * BK + a_kg * 4 + j;
                    f32x4 v = zero4;
                    if (kk < p.K) {
                        const float* src = A + (int64_t)kk * p.lda + o;
                        if (o + 3 < p.M) v = ld4(src);
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (o + e < p.M) v[e] = src[e];
                        }
                    }
                    R.ra[j] = v;
                }
            }
        } else {
            const int k = kt * BK + cidx * 4;
            const bool kok = k < p.K;
            if (SA) {
                const float* __restrict__ ak = ubase(A + kt * BK);
                const float* __restrict__ xk = AMODE == A_NADD ? ubase(p.A_add + kt * BK) : nullptr;
#pragma unroll
                for (int i = 0; i < NLA; ++i) {
                    R.ra[i] = ldb4(ak, sa_offb[i]);
                    if (AMODE == A_NADD) R.rx[i] = ldb4(xk, sa_addb[i]);
                    R.ra_ok[i] = a_ok[i];
                }
            } else if (AMODE == A_N || AMODE == A_NADD) {
                // branch-free: always load from a valid address, zero by select afterwards (keeps the K loop one
                // basic block so that the loads interleave with the MFMA stream)
                const int kc = kok ? k : 0;
#pragma unroll
                for (int i = 0; i < NLA; ++i) {
                    R.ra[i] = ld4(a_ptr[i] + kc);
                    if (AMODE == A_NADD) R.rx[i] = ld4(add_ptr[i] + kc);
                    R.ra_ok[i] = a_ok[i] && kok;
                }
            } else if (CF) {
                // advance the scalar tap state to tile kt (branch-free: kt is the last tile asked for, or the one after it)
                const int adv = kt != cf_kt ? 1 : 0;
                cf_kt = kt;
                cf_cb += adv * BK;
                const int wrap = cf_cb >= p.Cin ? 1 : 0;
                cf_cb = wrap ? 0 : cf_cb;
                cf_rs += wrap;
                cf_q += wrap;
                const int wq = cf_q >= p.KW ? 1 : 0;
                cf_q = wq ? 0 : cf_q;
                cf_r += wq;
                const int delta = (cf_r * p.W + cf_q) * p.Cin + cf_cb;
                const bool is_x = XS && kt >= ktx;                       // second source: block-uniform
                const float* __restrict__ abase = is_x ? Axg : A;
                const int dk = XS ? (kt - ktx) * BK : 0;
#pragma unroll
                for (int i = 0; i < NLA; ++i) {
                    const bool inb = is_x ? a_ok[i] : (((a_tapmask[i] >> cf_rs) & 1u) != 0);
                    const int off = is_x ? a_offx[XS ? i : 0] + dk : a_off0[i] + delta;
                    R.ra[i] = ldb4(abase, inb ? (unsigned)off * 4u : 0u);
                    R.ra_ok[i] = inb;
                }
            } else if (XS && kt >= ktx) {
                // second source (block-uniform): one tap, no padding -- every valid row reads Cx contiguous channels
                const int dk = (kt - ktx) * BK;
#pragma unroll
                for (int i = 0; i < NLA; ++i) {
                    R.ra[i] = ld4(Axg + (a_ok[i] ? a_offx[i] + dk : 0));
                    R.ra_ok[i] = a_ok[i];
                }
            } else if (AMODE == A_CONV && conv_fast) {
                // uniform tap of this K tile (small exact integer divisions via reciprocal multiply: kt < 2^20)
                // K order of the weight rows: (r, s, c) as the reference stores them, or -- k_tap_inner, the engine's split images of
                // the 3x3 convolutions -- (channel block of 32, r, s, channel): then the nine taps of a channel block follow each
                // other, a workgroup cycles through a quarter-to-sixteenth of its input patch at a time and the ~64 tiles an XCD
                // runs together keep their patches in its 4 MB L2 instead of re-fetching every tap through the fabric
                const int tpr = p.Cin / BK;
                const int ntap = p.KH * p.KW;
                int rs, cb;
                if (p.k_tap_inner) {
                    const int cbi = (int)(((float)kt + 0.5f) * conv_inv_ntap);
                    rs = kt - cbi * ntap;
                    cb = cbi * BK;
                } else {
                    rs = (int)(((float)kt + 0.5f) * conv_inv_tpr);
                    cb = (kt - rs * tpr) * BK;
                }
                const int r = (int)(((float)rs + 0.5f) * conv_inv_kw);
                const int q = rs - r * p.KW;
                const int delta = (r * p.W + q) * p.Cin + cb;
#pragma unroll
                for (int i = 0; i < NLA; ++i) {
                    const bool inb = (a_tapmask[i] >> rs) & 1u;
                    const int off = inb ? a_off0[i] + delta : 0;
                    R.ra[i] = ld4(a_ptr[i] + off);
                    R.ra_ok[i] = inb;
                }
            } else if (AMODE == A_CONV) {
                const int rs = k / p.Cin;
                const int c = k - rs * p.Cin;
                const int r = rs / p.KW;
                const int s = rs - r * p.KW;
#pragma unroll
                for (int i = 0; i < NLA; ++i) {
                    const int hi = a_hi0[i] + r, wi = a_wi0[i] + s;
                    const bool inb = a_ok[i] && kok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                    const int64_t off = inb ? ((int64_t)hi * p.W + wi) * p.Cin + c : 0;
                    R.ra[i] = ld4(a_ptr[i] + off);
                    R.ra_ok[i] = inb;
                }
            } else if (dgrad_fast) {
                // A_DGRAD, uniform tap per K tile (Cout % 32 == 0) and stride 1 or 2: branch-free, no divisions
                const int tpr = dg_cq / BK;
                const int rs = (int)(((float)kt + 0.5f) * dg_inv_tpr);
                const int nb = (kt - rs * tpr) * BK + cidx * 4;
                const int r = (int)(((float)rs + 0.5f) * conv_inv_kw);
                const int q = rs - r * p.KW;
                const int sh = p.stride - 1;                 // stride 1 -> shift 0, stride 2 -> shift 1
#pragma unroll
                for (int i = 0; i < NLA; ++i) {
                    const int hn = a_hi0[i] - r, wn = a_wi0[i] - q;
                    const int ho = hn >> sh, wo = wn >> sh;
                    const bool ok = a_ok[i] && hn >= 0 && wn >= 0 && ((hn | wn) & sh) == 0 && ho < p.Ho && wo < p.Wo;
                    const int off = ok ? (ho * p.Wo + wo) * dg_cq + nb : 0;
                    R.ra[i] = ld4(a_ptr[i] + off);
                    R.ra_ok[i] = ok;
                }
            } else {   // A_DGRAD: contraction (r, s, n) over the forward output channels
                const int Cq = p.K / (p.KH * p.KW);
                const int rs = k / Cq;
                const int n = k - rs * Cq;
                const int r = rs / p.KW;
                const int s = rs - r * p.KW;
#pragma unroll
                for (int i = 0; i < NLA; ++i) {
                    const int hn = a_hi0[i] - r, wn = a_wi0[i] - s;
                    f32x4 v = zero4;
                    if (a_ok[i] && kok && hn >= 0 && wn >= 0) {
                        const int ho = hn / p.stride, wo = wn / p.stride;
                        if (ho * p.stride == hn && wo * p.stride == wn && ho < p.Ho && wo < p.Wo)
                            v = ld4(a_ptr[i] + ((int64_t)ho * p.Wo + wo) * Cq + n);
                    }
                    R.ra[i] = v;
                    R.ra_ok[i] = true;
                }
            }
        }
        // ---------------- B
        if (BMODE == B_N && (CF || SA)) {
            // uniform: this K tile's column block (read through readfirstlane so that the base lands in SGPRs and the loads
            // take the saddr + 32-bit offset form)
            const uint64_t bku = reinterpret_cast<uint64_t>(Bw + kt * BK);
            const float* __restrict__ bk = reinterpret_cast<const float*>(
                ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(bku >> 32)) << 32) |
                (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bku));
#pragma unroll
            for (int i = 0; i < NLB; ++i) {
                R.rb[i] = ldb4(bk, b_offb[i]);
                R.rb_ok[i] = b_ok[i];
            }
        } else if (BMODE == B_N) {
            const int k = kt * BK + cidx * 4;
            const bool kok = k < p.K;
            const int kc = kok ? k : 0;
#pragma unroll
            for (int i = 0; i < NLB; ++i) {
                R.rb[i] = ld4(b_ptr[i] + kc);
                R.rb_ok[i] = b_ok[i] && kok;
            }
        } else if (BMODE == B_T) {
            // (measured at the training shapes: with a transposed A beside it this interior path takes the A_T x B_T launches from
            // 345 to 308 us; beside a row-major A the same path is 5 % SLOWER -- A_N x B_T 651 vs 617 us -- so only then)
            if (AMODE == A_T && b_kg < NPL && bt_interior && (kt + 1) * BK <= p.K) {
                const float* __restrict__ rb = ubase(Bw + (int64_t)kt * BK * p.ldb);
#pragma unroll
                for (int j = 0; j < 4; ++j) R.rb[j] = ldb4(rb + (int64_t)j * p.ldb, bt_offb);
            } else if (b_kg < NPL) {
                const int o = n0 + b_og * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kk = kt * BK + b_kg * 4 + j;
                    f32x4 v = zero4;
                    if (kk < p.K) {
                        const float* src = Bw + (int64_t)kk * p.ldb + o;
                        const float* ad = p.B_add ? p.B_add + (int64_t)(kk % p.badd_mod) * p.ld_badd + o : nullptr;
                        if (o + 3 < p.N) {
                            v = ld4(src);
                            if (ad) v += ld4(ad);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (o + e < p.N) v[e] = src[e] + (ad ? ad[e] : 0.f);
                        }
                    }
                    R.rb[j] = v;
                }
            }
        } else {   // B_WGRAD: B[(r,s,c)][m] gathered from the forward conv's NHWC input
            if (b_kg < NPL) {
                const bool ook = (n0 + b_og * 4) < p.N;
                const int hw = p.Ho * p.Wo;
                // pixel index -> (image, row, column) by reciprocal multiply + one-step correction instead of two integer
                // divisions per pixel (the decode was most of this loader's instructions); the 4 pixels are consecutive
                const int mm0 = kt * BK + b_kg * 4;
                int pb = (int)((float)mm0 * wg_inv_hw);
                int prem = mm0 - pb * hw;
                if (prem < 0) { --pb; prem += hw; } else if (prem >= hw) { ++pb; prem -= hw; }
                int pho = (int)((float)prem * wg_inv_wo);
                int pwo = prem - pho * p.Wo;
                if (pwo < 0) { --pho; pwo += p.Wo; } else if (pwo >= p.Wo) { ++pho; pwo -= p.Wo; }
                // branch-free: a load under a divergent branch is waited for on the spot (the four gathers of a step then run
                // one after the other, each a full memory round trip); here every lane always loads from a valid address (the
                // image's first element when its tap is padding or its pixel beyond the contraction) and the staging zeroes it
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int mm = mm0 + j;
                    const int b = pb, ho = pho, wo = pwo;
                    const bool wrapw = pwo + 1 == p.Wo;
                    const bool wraph = wrapw && pho + 1 == p.Ho;
                    pwo = wrapw ? 0 : pwo + 1;
                    pho = wraph ? 0 : (wrapw ? pho + 1 : pho);
                    pb += wraph ? 1 : 0;
                    const int hi = ho * p.stride - p.pad + wg_r, wi = wo * p.stride - p.pad + wg_s;
                    const bool ok = ook && mm < p.K && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                    const int64_t off = ok ? (int64_t)b * p.img_stride + (int64_t)((hi * p.W + wi) * p.Cin + wg_c) : 0;
                    R.rb[j] = ld4(Bw + off);
                    R.rb_ok[j] = ok;
                }
            }
        }
    };
    // fp16-split staging: a thread's float4 (4 consecutive k of one row) becomes 4 hi halfs + 4 lo halfs -- two 8-byte
    // stores into the (hi, lo) planes of its contraction group, at half (chunk & 1) of the 16-byte unit.
    // MASKED = false: no zero-fill selects (legal when K is a multiple of the K tile and the operand is plain row-major:
    // rows beyond M / N are staged from a valid clamped address and only ever reach outputs that are not stored).
    // optional power-of-two operand pre-scales (keep small operands' lo pieces out of the fp16 subnormals; undone through
    // alpha).  Only the MASKED flavour applies them; the host never selects the unmasked one when they are set.
    // a *_scale_dev pointer supplies a scale computed on the device (actmi_op_pow2_scale: power of two that brings the
    // operand's largest magnitude to [2^13, 2^14)) -- the backward pass uses it for gradient operands, whose magnitudes
    // span many decades across the network
    const float pre_a = (p.a_scale != 0.f ? p.a_scale : 1.f) * (p.a_scale_dev ? *p.a_scale_dev : 1.f);
    const float pre_b = BSPLIT ? 1.f : (p.b_scale != 0.f ? p.b_scale : 1.f) * (p.b_scale_dev ? *p.b_scale_dev : 1.f);
    auto store16 = [&](int stage, auto& R, auto maskedc) {
        constexpr bool MASKED = decltype(maskedc)::value;
        f32x4* sa = smem + stage * STAGE;
        f32x4* sb = sa + NPL * PSA;
        uint2* sa8 = reinterpret_cast<uint2*>(sa);
        uint2* sb8 = reinterpret_cast<uint2*>(sb);
        if (AMODE == A_T) {
            if (a_kg < NPL) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 v = f32x4{R.ra[0][i], R.ra[1][i], R.ra[2][i], R.ra[3][i]} * pre_a;
                    uint2 hi, lo;
                    const int row = swz_row(a_og * 4 + i);
                    if (PREC == PREC_BF16) sa8[(((a_kg >> 1) * 2 + 0) * PSA + row) * 2 + (a_kg & 1)] = pack_bf16x4(v);
                    else {
                    split16(v, hi, lo);
                    sa8[(((a_kg >> 1) * 2 + 0) * PSA + row) * 2 + (a_kg & 1)] = hi;
                    sa8[(((a_kg >> 1) * 2 + 1) * PSA + row) * 2 + (a_kg & 1)] = lo;
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NLA; ++i) {
                f32x4 v = R.ra[i];
                if (AMODE == A_NADD) { if (use_add) v += R.rx[i]; }
                if (AMODE == A_CONV || AMODE == A_DGRAD || MASKED) v = R.ra_ok[i] ? v : zero4;
                if (MASKED) v *= pre_a;
                uint2 hi, lo;
                const int row = srow + RPP * i;
                if (PREC == PREC_BF16) sa8[(((cidx >> 1) * 2 + 0) * PSA + row) * 2 + (cidx & 1)] = pack_bf16x4(v);
                else {
                split16(v, hi, lo);
                sa8[(((cidx >> 1) * 2 + 0) * PSA + row) * 2 + (cidx & 1)] = hi;
                sa8[(((cidx >> 1) * 2 + 1) * PSA + row) * 2 + (cidx & 1)] = lo;
                }
            }
        }
        if (BMODE == B_N) {
#pragma unroll
            for (int i = 0; i < NLB; ++i) {
                f32x4 v = (!MASKED || R.rb_ok[i]) ? R.rb[i] : zero4;
                if (MASKED && !BSPLIT) v *= pre_b;
                uint2 hi, lo;
                const int row = srow + RPP * i;
                if (PREC == PREC_BF16) { sb8[(((cidx >> 1) * 2 + 0) * PSB + row) * 2 + (cidx & 1)] = pack_bf16x4(v); continue; }
                if (BSPLIT) {     // weights split ahead of time: each 16-byte group is {4 hi halfs, 4 lo halfs}
                    const uint4 u = __builtin_bit_cast(uint4, v);
                    hi = uint2{u.x, u.y};
                    lo = uint2{u.z, u.w};
                } else split16(v, hi, lo);
                sb8[(((cidx >> 1) * 2 + 0) * PSB + row) * 2 + (cidx & 1)] = hi;
                sb8[(((cidx >> 1) * 2 + 1) * PSB + row) * 2 + (cidx & 1)] = lo;
            }
        } else {
            if (b_kg < NPL) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 v = f32x4{R.rb[0][i], R.rb[1][i], R.rb[2][i], R.rb[3][i]} * pre_b;
                    if (BMODE == B_WGRAD) v = f32x4{R.rb_ok[0] ? v[0] : 0.f, R.rb_ok[1] ? v[1] : 0.f, R.rb_ok[2] ? v[2] : 0.f, R.rb_ok[3] ? v[3] : 0.f};
                    uint2 hi, lo;
                    const int row = swz_row(b_og * 4 + i);
                    if (PREC == PREC_BF16) sb8[(((b_kg >> 1) * 2 + 0) * PSB + row) * 2 + (b_kg & 1)] = pack_bf16x4(v);
                    else {
                    split16(v, hi, lo);
                    sb8[(((b_kg >> 1) * 2 + 0) * PSB + row) * 2 + (b_kg & 1)] = hi;
                    sb8[(((b_kg >> 1) * 2 + 1) * PSB + row) * 2 + (b_kg & 1)] = lo;
                    }
                }
            }
        }
    };
    auto store_tile = [&](int stage, auto& R) {
        f32x4* sa = smem + stage * STAGE;
        f32x4* sb = sa + NPL * PSA;
        if (AMODE == A_T) {
            if (a_kg < NPL) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 v = {R.ra[0][i], R.ra[1][i], R.ra[2][i], R.ra[3][i]};
                    sa[a_kg * PSA + swz_row(a_og * 4 + i)] = v;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NLA; ++i) {
                f32x4 v = R.ra[i];
                if (AMODE == A_NADD) { if (use_add) v += R.rx[i]; }
                v = R.ra_ok[i] ? v : zero4;
                sa[cidx * PSA + srow + RPP * i] = v;
            }
        }
        if (BMODE == B_N) {
#pragma unroll
            for (int i = 0; i < NLB; ++i) sb[cidx * PSB + srow + RPP * i] = R.rb_ok[i] ? R.rb[i] : zero4;
        } else {
            if (b_kg < NPL) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 v = {R.rb[0][i], R.rb[1][i], R.rb[2][i], R.rb[3][i]};
                    if (BMODE == B_WGRAD) v = f32x4{R.rb_ok[0] ? v[0] : 0.f, R.rb_ok[1] ? v[1] : 0.f, R.rb_ok[2] ? v[2] : 0.f, R.rb_ok[3] ? v[3] : 0.f};
                    sb[b_kg * PSB + swz_row(b_og * 4 + i)] = v;
                }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nsteps = kt_end - kt_begin;
    if constexpr (PREC != PREC_F32) {
        // fp16-split (or single bf16 product) main loop.  A 32-deep K tile is only 24 MFMAs (768 pipe cycles) per wave here -- a fifth of the
        // fp32 instruction's time -- so global loads run TWO tiles ahead of the MFMAs (two register stages), the LDS
        // stage one tile ahead, and the loop body is kept ONE basic block (tail loads are clamped to the last tile
        // instead of branched around; a surplus stage store is harmless) so that the scheduler can weave the next
        // tile's conversions and LDS traffic between the MFMAs.
        auto compute16 = [&](int cur) {
            const f32x4* sa = smem + cur * STAGE;
            const f32x4* sb = sa + NPL * PSA;
            f32x4 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int kg = 2 * s2 + lh;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[s2][i] = sa[(kg * 2 + 0) * PSA + arow[i]];
                    if (PREC == PREC_F16X3) al[s2][i] = sa[(kg * 2 + 1) * PSA + arow[i]];
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[s2][j] = sb[(kg * 2 + 0) * PSB + bcol[j]];
                    if (PREC == PREC_F16X3) bl[s2][j] = sb[(kg * 2 + 1) * PSB + bcol[j]];
                }
            }
            if constexpr (PREC == PREC_BF16) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[s2][i]),
                                                                                __builtin_bit_cast(bf16x8, bh[s2][j]), acc[i][j], 0, 0, 0);
                return;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const h16x8 xh = __builtin_bit_cast(h16x8, ah[s2][i]), xl = __builtin_bit_cast(h16x8, al[s2][i]);
                        const h16x8 yh = __builtin_bit_cast(h16x8, bh[s2][j]), yl = __builtin_bit_cast(h16x8, bl[s2][j]);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, yh, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yl, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yh, acc[i][j], 0, 0, 0);
                    }
        };
        auto run = [&](auto maskedc) {
            Regs R1;
            const int kt_last = kt_end - 1;
            auto ld = [&](int kt, auto& R) { load_tile(kt < kt_last ? kt : kt_last, R); };
            constexpr bool HOT = (AMODE == A_N || AMODE == A_NADD || AMODE == A_CONV) && BMODE == B_N;
            auto weave = [&]() {
                // per MFMA: a few conversion / address VALU ops, one LDS access and one global load of the tiles ahead
                if (HOT) {
#pragma unroll
                    for (int r = 0; r < 6 * TM * TN; ++r) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x006, 5, 0);
                        __builtin_amdgcn_sched_group_barrier(0x080, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
            };
            ld(kt_begin, R0);
            ld(kt_begin + 1, R1);
            store16(0, R0, maskedc);
            __syncthreads();
            int s_ = 0;
            for (; s_ + 1 < nsteps; s_ += 2) {
                // tile s_ is in LDS stage 0, tile s_+1 in flight in R1; R0 is free
                ld(kt_begin + s_ + 2, R0);
                compute16(0);
                store16(1, R1, maskedc);
                weave();
                __syncthreads();
                ld(kt_begin + s_ + 3, R1);
                compute16(1);
                store16(0, R0, maskedc);
                weave();
                __syncthreads();
            }
            if (s_ < nsteps) {
                compute16(0);
                __syncthreads();
            }
        };
        // the loop flavour is chosen on the host (launch_cfg): UNMASKED needs K % 32 == 0 and no operand pre-scales
        if constexpr (UNMASKED != 0) run(std::false_type{});
        else run(std::true_type{});
    } else {
    load_tile(kt_begin, R0);
    store_tile(0, R0);
    __syncthreads();
    // One K step.  PF = prefetch the next tile: the global loads are issued right after the first fragment reads
    // so that (with the branch-free loaders) they sit in the same basic block as the MFMAs and interleave with them;
    // the LDS stores of the next stage follow the last MFMA group.  The final step is peeled (PF = false).
    // step s computes LDS stage s&1 while the global loads of tile s+1 are in flight (issued in the shadow of the first
    // MFMAs), then stores that tile into the other stage.  (A two-step-deep register prefetch was measured: no gain --
    // the loop is bound by the CU's L1 fill rate, ~8 B/clk at a 128x128 fp32 tile, not by load latency.)
    auto kstep = [&](int s_, auto par, auto pf) {
        constexpr bool PF = decltype(pf)::value;
        constexpr int cur = decltype(par)::value;
        const int kt = kt_begin + s_;
        const f32x4* sa = smem + cur * STAGE;
        const f32x4* sb = sa + NPL * PSA;
        f32x4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = sa[lh * PSA + arow[i]];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = sb[lh * PSB + bcol[j]];
        constexpr bool HOT = (AMODE == A_N || AMODE == A_NADD || AMODE == A_CONV) && BMODE == B_N;
        if (PF) load_tile(kt + 1, R0);
        auto kblock = [&](auto kbc) {
            constexpr int kb = decltype(kbc)::value;
            if (kb + 1 < NPL / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[(kb + 1) & 1][i] = sa[(2 * (kb + 1) + lh) * PSA + arow[i]];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[(kb + 1) & 1][j] = sb[(2 * (kb + 1) + lh) * PSB + bcol[j]];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kb & 1][i][e], bf[kb & 1][j][e], acc[i][j], 0, 0, 0);
            if (HOT) {
                // issue order inside this k-block: each 64-cycle MFMA is followed by a few address/VALU ops, at most one
                // global load of the next tile and one LDS fragment read of the next k-block (they run in the MFMA's shadow)
#pragma unroll
                for (int r = 0; r < 4 * TM * TN; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x006, kb == 0 ? 10 : 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        kblock(std::integral_constant<int, 0>{});
        kblock(std::integral_constant<int, 1>{});
        kblock(std::integral_constant<int, 2>{});
        kblock(std::integral_constant<int, 3>{});
        static_assert(NPL / 2 == 4, "k-block unroll assumes BK = 32");
        if (PF) store_tile(cur ^ 1, R0);
        __syncthreads();
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int s_ = 0;
    for (; s_ + 2 < nsteps; s_ += 2) {
        kstep(s_, P0{}, std::true_type{});
        kstep(s_ + 1, P1{}, std::true_type{});
    }
    if (nsteps - s_ == 2) {
        kstep(s_, P0{}, std::true_type{});
        kstep(s_ + 1, P1{}, std::false_type{});
    } else {
        kstep(s_, P0{}, std::false_type{});
    }
    }   // PREC_F32

    if (stamp && threadIdx.x == 0) stamp[2] = __builtin_amdgcn_s_memtime();
    // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const float* scale = p.scale ? p.scale + (int64_t)g * p.gSB : nullptr;
    const float* bias = p.bias ? p.bias + (int64_t)g * p.gSB : nullptr;
    const float* res = p.res ? p.res + offRes : nullptr;
    const float* mask = p.mask ? p.mask + (int64_t)g * p.gMask : nullptr;
    const int64_t ldmask = p.ldmask ? p.ldmask : p.ldc;
    float* C = p.C + offC + (int64_t)split * p.split_stride;      // split_stride != 0: one plain slice per split
    float* C2 = p.C2 ? p.C2 + (int64_t)g * p.gC2out : nullptr;
    // a pre-split B image may carry a power-of-two scale (keeps small weights' lo pieces out of fp16 subnormals)
    float alpha = p.alpha != 0.f ? p.alpha : 1.f;
    if (PREC != PREC_F32) {
        if (p.b_scale != 0.f) alpha /= p.b_scale;
        if (p.a_scale != 0.f) alpha /= p.a_scale;
        if (p.a_scale_dev) alpha /= *p.a_scale_dev;
        if (!BSPLIT && p.b_scale_dev) alpha /= *p.b_scale_dev;
    }
    const bool has_mask = mask != nullptr, has_map = p.rowmap != nullptr;
    // fused attention-backward epilogues (actmi.h: epi): a per-row vector, a per-column kill mask, res as a factor
    const int epi = UNMASKED ? 0 : p.epi;          // the hot forward flavour carries neither these epilogues nor the amax
    const float* erow = nullptr;
    const uint8_t* ekill = nullptr;
    if (epi != 0) {
        if (p.groups_inner > 0) {
            const int g1 = g / p.groups_inner, g2 = g % p.groups_inner;
            erow = p.epi_row + g1 * p.gRow + g2 * p.gRow2;
            ekill = p.epi_colkill ? p.epi_colkill + g1 * p.gColkill : nullptr;
        } else {
            erow = p.epi_row + (int64_t)g * p.gRow;
            ekill = p.epi_colkill ? p.epi_colkill + (int64_t)g * p.gColkill : nullptr;
        }
    }
    const bool has_res = res != nullptr && epi != 2;          // epi 2 multiplies by res instead of adding it
    const float drop_scale = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
    // Fast epilogue for the forward-pass cases (bias / FrozenBN affine, optional same-shape residual, optional ReLU): the
    // feature-complete path below costs ~140 instructions per element (per-element branches, 64-bit index arithmetic) and
    // was measured at 45-75k cycles per 128x128 tile, a quarter of a K=512 main loop; this one is a few thousand.
    const bool simple = !has_map && (splitk <= 1 || p.split_stride != 0) && !C2 && !(p.drop_p > 0.f) && p.res_mod == 0 &&
                        (!has_mask || (int64_t)p.M * ldmask < (int64_t)1 << 31) &&
                        (int64_t)p.M * p.ldc < (int64_t)1 << 31 && (!res || (int64_t)p.M * p.ldres < (int64_t)1 << 31);
    unsigned amx = 0;                  // bits of max |stored value| (p.amax_out)
    auto amax_flush = [&]() {
        if (!track) return;
        for (int o = 32; o > 0; o >>= 1) amx = max(amx, (unsigned)__shfl_xor((int)amx, o, 64));
        if (fflag && lane == 0 && amx >= 0x7f800000u) atomicOr(fflag, p.finite_bit);      // inf / NaN bits order above every finite value
        if (!amax_out) return;
        // amax_seen was read when the block started (a lower bound of the running maximum: it only filters): no load
        // latency at the tail of every tile
        if (lane == 0 && amx > amax_seen) atomicMax(amax_out, amx);
    };
    if (simple) {
        const bool relu = p.relu == 1, gelu = p.relu == 2;

        // Vector form: the wave's accumulator tile goes through the (now idle) LDS stage one 32-row band at a time and
        // leaves as 16-byte stores, a row segment of WN floats per WN/4 lanes -- 4x fewer store instructions, each
        // covering whole 128-byte lines.  (The scalar form below writes 4 bytes per lane; its store burst was measured
        // at ~2.7 TB/s chip-wide against ~6 TB/s for contiguous 16-byte stores.)
        const bool vec = n0 + BN <= p.N && (p.ldc & 3) == 0 && ((uintptr_t)C & 15) == 0 &&
                         (!res || ((p.ldres & 3) == 0 && ((uintptr_t)res & 15) == 0)) &&
                         (!has_mask || ((ldmask & 3) == 0 && ((uintptr_t)mask & 15) == 0)) &&
                         (!scale || ((uintptr_t)scale & 15) == 0) && (!bias || ((uintptr_t)bias & 15) == 0);
        if (vec) {
            constexpr int RS = WN + 8;               // row stride (floats): the two lane halves land 32 banks apart
            constexpr int LPR = WN / 4;              // lanes per row in the read-back
            constexpr int RPI = 64 / LPR;            // rows per read-back instruction
            static_assert(4 * 32 * RS * 4 <= 2 * STAGE * 16 || (BM / WM) * (BN / WN) * 32 * RS * 4 <= 2 * STAGE * 16, "epilogue scratch must fit the stage");
            float* scr = reinterpret_cast<float*>(smem) + wave * (32 * RS);
            const int c4 = lane % LPR, r0 = lane / LPR;
            const int ncol = n0 + wcol0 + c4 * 4;
            const f32x4 one4 = {1.f, 1.f, 1.f, 1.f};
            const f32x4 sc4 = scale ? ld4(scale + ncol) : one4;
            const f32x4 bi4 = bias ? ld4(bias + ncol) : zero4;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        scr[((e & 3) + 8 * (e >> 2) + 4 * lh) * RS + j * 32 + li] = acc[i][j][e];
                const int mbase = m0 + wrow0 + i * 32;
#pragma unroll
                for (int q = 0; q < 32 / RPI; ++q) {
                    const int r = q * RPI + r0;
                    const int m = mbase + r;
                    f32x4 v = *reinterpret_cast<const f32x4*>(scr + r * RS + c4 * 4);
                    v = v * alpha * sc4 + bi4;
                    if (m < p.M) {
                        if (epi == 1) {
                            const float rsub = erow[m];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = (ekill && ekill[ncol + e] != 0) ? 0.f : expf(v[e] - rsub);
                        } else if (epi == 2) {
                            const f32x4 pr = ld4(res + (uint32_t)(m * (int)p.ldres + ncol));
                            v = pr * (v - erow[m]) * p.epi_scale;
                        }
                        if (has_res) v += ld4(res + (uint32_t)(m * (int)p.ldres + ncol));
                        if (has_mask) {
                            const f32x4 mk = ld4(mask + (uint32_t)(m * (int)ldmask + ncol));
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
                        }
                        if (relu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                        }
                        if (gelu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.f + erff(v[e] * 0.70710678118654752f));
                        }
                        *reinterpret_cast<f32x4*>(C + (uint32_t)(m * (int)p.ldc + ncol)) = v;
                        if (track) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) amx = max(amx, __float_as_uint(v[e]) & 0x7fffffffu);
                        }
                    }
                }
            }
            amax_flush();
            if (stamp && threadIdx.x == 0) stamp[3] = __builtin_amdgcn_s_memtime();
            return;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wcol0 + j * 32 + li;
            const bool nok = n < p.N;
            const float sc = (scale && nok) ? scale[n] : 1.f;
            const float bi = (bias && nok) ? bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mb = m0 + wrow0 + i * 32 + 4 * lh;
                float rv[16];
                if (has_res) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int m = mb + (e & 3) + 8 * (e >> 2);
                        const int mc = m < p.M ? m : 0;
                        rv[e] = nok ? res[(uint32_t)(mc * (int)p.ldres + n)] : 0.f;
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = mb + (e & 3) + 8 * (e >> 2);
                    float v = acc[i][j][e] * alpha * sc + bi;
                    if (epi != 0 && nok && m < p.M) {
                        if (epi == 1) v = (ekill && ekill[n] != 0) ? 0.f : expf(v - erow[m]);
                        else v = res[(uint32_t)(m * (int)p.ldres + n)] * (v - erow[m]) * p.epi_scale;
                    }
                    if (has_res) v += rv[e];
                    if (has_mask && nok && m < p.M) { if (!(mask[(uint32_t)(m * (int)ldmask + n)] > 0.f)) v = 0.f; }
                    v = relu ? fmaxf(v, 0.f) : v;
                    if (gelu) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
                    if (nok && m < p.M) {
                        C[(uint32_t)(m * (int)p.ldc + n)] = v;
                        if (track) amx = max(amx, __float_as_uint(v) & 0x7fffffffu);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        amax_flush();
        if (stamp && threadIdx.x == 0) stamp[3] = __builtin_amdgcn_s_memtime();
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wcol0 + j * 32 + li;
        const bool nok = n < p.N;
        const float sc = (scale && nok) ? scale[n] : 1.f;
        const float bi = (bias && nok) ? bias[n] : 0.f;
        const float sc2 = (C2 && nok) ? p.scale2[(int64_t)g * p.gSB + n] : 1.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            // phase A: row indices (row-map loads batched)
            int orow[16];
            bool ok[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wrow0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                ok[e] = nok && m < p.M;
                orow[e] = m;
                if (has_map && ok[e]) orow[e] = p.rowmap[m];
            }
            // phase B: residual / mask loads, all in flight together
            float rv[16], mv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wrow0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                rv[e] = 0.f;
                mv[e] = 1.f;
                if (has_res && ok[e]) rv[e] = res[(int64_t)(p.res_mod ? (m % p.res_mod) : m) * p.ldres + n];
                if (has_mask && ok[e]) mv[e] = mask[(int64_t)m * ldmask + n];
            }
            // phase C: combine and store
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (ok[e]) {
                    float v = acc[i][j][e] * alpha;
                    v = scale ? v * sc + bi : v + bi;
                    const int64_t o = (int64_t)orow[e] * p.ldc + n;
                    if (p.drop_p > 0.f) {
                        if (p.relu && !has_res) v = fmaxf(v, 0.f);          // dropout(relu(x)) for the FFN hidden layer
                        v = actmi_keep(p.drop_seed, (uint64_t)(offC + o), p.drop_p) ? v * drop_scale : 0.f;
                    }
                    v += rv[e];
                    if (!(mv[e] > 0.f)) v = 0.f;
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (splitk > 1 && p.split_stride == 0) atomicAdd(&C[o], v);
                    else C[o] = v;
                    if (track) amx = max(amx, __float_as_uint(v) & 0x7fffffffu);
                    if (C2) C2[o] = v * sc2;
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the next tile's loads from being hoisted (register pressure)
        }
    }
    amax_flush();
    if (stamp && threadIdx.x == 0) stamp[3] = __builtin_amdgcn_s_memtime();
}

template <int BM, int BN, int WM, int WN, int AMODE, int BMODE, int PREC, int BSPLIT, int UNMASKED, int XS = 0>
int launch_cfg_u(const GemmArgs& a, hipStream_t st) {
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    static const int lds_pad = getenv("ACTMI_GEMM_LDSPAD") ? atoi(getenv("ACTMI_GEMM_LDSPAD")) : 0;   // tuning aid
    const int smem = 2 * stage_f4<BM, BN, PREC>() * 16 + lds_pad;
    auto kern = gemm_f32_kernel<BM, BN, WM, WN, AMODE, BMODE, PREC, BSPLIT, UNMASKED, XS>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int splitk = a.splitk > 1 ? a.splitk : 1;
    dim3 grid(tiles_m * tiles_n, 1, (a.groups > 0 ? a.groups : 1) * splitk);
    if (prof_enabled()) {
        char nm[128];
        // ACTMI_PROF_SHAPES=1: one profile class per distinct launch shape (bench.py --shapes: the per-shape table)
        static const bool by_shape = getenv("ACTMI_PROF_SHAPES") && getenv("ACTMI_PROF_SHAPES")[0] == '1';
        if (by_shape)
            snprintf(nm, sizeof(nm), "gemm_%s_kernel<%d,%d,%d,%d,%d,%d%s>[M=%d,N=%d,K=%d,g=%d,sk=%d,wgs=%d]", PREC == PREC_BF16 ? "bf16" : PREC ? "f16x3" : "f32",
                     BM, BN, WM, WN, AMODE, BMODE, XS ? ",xs" : "", a.M, a.N, a.K, a.groups, splitk, tiles_m * tiles_n * a.groups * splitk);
        else
            snprintf(nm, sizeof(nm), "gemm_%s_kernel<%d,%d,%d,%d,%d,%d%s>", PREC == PREC_BF16 ? "bf16" : PREC ? "f16x3" : "f32", BM, BN, WM, WN, AMODE, BMODE, XS ? ",xs" : "");
        const double g = a.groups;
        double abytes;
        if (AMODE == A_CONV) abytes = (double)(a.M / (a.Ho * a.Wo)) * a.H * a.W * a.Cin + (XS ? (double)a.M * a.Cx : 0.0);
        else if (AMODE == A_DGRAD) abytes = (double)(a.M / (a.H * a.W)) * a.Ho * a.Wo * (a.K / (a.KH * a.KW));
        else abytes = (double)a.M * a.K;
        const double bbytes = (BMODE == B_WGRAD) ? (double)(a.K / (a.Ho * a.Wo)) * a.H * a.W * a.Cin : (double)a.N * a.K;
        // algorithmic work: 2*M*N*K flops; operands read once + result written once
        prof_begin(nm, 2.0 * a.M * a.N * a.K * g, 4.0 * g * ((double)a.M * a.N + abytes + bbytes), st);
    }
    hipLaunchKernelGGL(kern, grid, dim3((BM / WM) * (BN / WN) * 64), smem, st, a, tiles_m, tiles_n);
    prof_end(st);
    return (int)hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int AMODE, int BMODE, int PREC, int BSPLIT>
int launch_cfg_b(const GemmArgs& a, hipStream_t st) {
    constexpr bool HOT = PREC != PREC_F32 && BMODE == B_N && (AMODE == A_N || AMODE == A_NADD || AMODE == A_CONV);
    if constexpr (HOT) {
        const bool one_a = (a.a_scale == 0.f || a.a_scale == 1.f) && !a.a_scale_dev;
        const bool one_b = BSPLIT || ((a.b_scale == 0.f || a.b_scale == 1.f) && !a.b_scale_dev);
        bool unmasked_ok = (a.K % BK) == 0 && one_a && one_b && a.epi == 0 && !a.amax_out && !a.finite_flag;
        if constexpr (ACTMI_GEMM_SLIM != 0 && (AMODE == A_N || AMODE == A_NADD)) {
            // the unmasked row-major flavours address both operands as uniform base + unsigned 32-bit byte offset
            const int64_t amax_rows = a.a_rowmap ? ((int64_t)1 << 40) : a.M;        // (a row gather could point anywhere)
            unmasked_ok = unmasked_ok && amax_rows * a.lda < ((int64_t)1 << 30) && (int64_t)a.N * a.ldb < ((int64_t)1 << 30) &&
                          (!a.A_add || (int64_t)a.add_mod * a.ld_add < ((int64_t)1 << 30));
        }
        if constexpr (ACTMI_GEMM_SLIM != 0 && AMODE == A_CONV) {
            // the unmasked convolution flavour is compiled for the uniform-tap fast path with 32-bit byte offsets only
            const int64_t nimg = a.Ho > 0 && a.Wo > 0 ? a.M / ((int64_t)a.Ho * a.Wo) : 0;
            unmasked_ok = unmasked_ok && (a.Cin % BK) == 0 && a.KH * a.KW <= 32 && nimg * a.img_stride < ((int64_t)1 << 30) &&
                          (int64_t)a.H * a.W * a.Cin <= a.img_stride && (int64_t)a.N * a.ldb < ((int64_t)1 << 30);
        }
        if constexpr (AMODE == A_CONV && BSPLIT == 1) {
            // second-source form: only built for the hot (unmasked, pre-split weights) flavour; launch_gemm has checked the rest
            if (a.Ax) return unmasked_ok ? launch_cfg_u<BM, BN, WM, WN, AMODE, BMODE, PREC, BSPLIT, 1, 1>(a, st) : -1001;
        }
        if (unmasked_ok) return launch_cfg_u<BM, BN, WM, WN, AMODE, BMODE, PREC, BSPLIT, 1>(a, st);
    }
    if (a.Ax) return -1001;              // no instantiation carries a second source in this precision / operand form
    return launch_cfg_u<BM, BN, WM, WN, AMODE, BMODE, PREC, BSPLIT, 0>(a, st);
}

template <int BM, int BN, int WM, int WN, int AMODE, int BMODE, int PREC>
int launch_cfg(const GemmArgs& a, hipStream_t st) {
    // pre-split weights exist only for the forward operand forms
    constexpr bool CAN_BSPLIT = PREC == PREC_F16X3 && BMODE == B_N && (AMODE == A_N || AMODE == A_NADD || AMODE == A_CONV);
    if constexpr (CAN_BSPLIT) {
        if (a.b_split) return launch_cfg_b<BM, BN, WM, WN, AMODE, BMODE, PREC, 1>(a, st);
    }
    return launch_cfg_b<BM, BN, WM, WN, AMODE, BMODE, PREC, 0>(a, st);
}

// Estimated launch time of a tile shape, in units of "one 128x128 K tile at full speed".  Two workgroups share a CU:
// 512 residency slots; a surplus round with at most one workgroup per CU costs one tile time, a fuller one up to two;
// a launch that never pairs workgroups on a CU (<= 256 tiles) hides nothing (x1.3).  Per tile: a fixed prologue +
// epilogue worth `fixed` K tiles plus nk K tiles at the shape's relative speed `eff` (both measured on the ACT shapes;
// the fp16-split loop spends 5x fewer MFMA cycles per staged byte, so small tiles and the fixed part weigh more).
double tile_cost(int M, int N, int K, int nz, int BM, int BN, double eff, double fixed) {
    const long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN) * nz;
    const long full = tiles / 512, f = tiles - full * 512;
    double rounds = 2.0 * full + (f == 0 ? 0.0 : (f <= 256 ? 1.0 : 1.0 + (double)(f - 256) / 256.0));
    if (tiles <= 256) rounds *= 1.3;
    const double nk = (double)((K + BK - 1) / BK);
    return rounds * ((double)BM * BN / (128.0 * 128.0)) * (fixed + nk / eff);
}

template <int AMODE, int BMODE, int PREC>
int launch_modes(const GemmArgs& a, hipStream_t st) {
    const int nz = a.groups > 0 ? a.groups : 1;
    const int Ks = a.splitk > 1 ? (a.K + a.splitk - 1) / a.splitk : a.K;
    const int nzs = nz * (a.splitk > 1 ? a.splitk : 1);
    const double fixed = PREC ? 10.0 : 2.5;
    const double cL = tile_cost(a.M, a.N, Ks, nzs, 128, 128, 1.00, fixed);
    // f16x3 128x64: the plain row-major and the transposed-A forms fit 168 VGPRs -> three waves per SIMD (0.74 -> 0.83 of
    // the 128x128 rate); the gather forms spill at that budget and stay at two
    const double cM = tile_cost(a.M, a.N, Ks, nzs, 128, 64, PREC ? ((AMODE == A_N || AMODE == A_T) ? 0.83 : 0.74) : 0.95, fixed);
    const double cS = tile_cost(a.M, a.N, Ks, nzs, 64, 64, PREC ? 0.68 : 0.88, fixed);
    double eL = -cL, eM = -cM, eS = -cS;
    if (!PREC) {
        // native fp32 MFMA: the K loop dominates a tile, so whole rounds of 256 tiles describe it well (measured)
        auto eff = [&](int BM, int BN, double factor) {
            const long tiles = (long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * nzs;
            const long rounds = (tiles + 255) / 256;
            return factor * ((double)a.M * a.N * nzs) / ((double)rounds * 256 * BM * BN);
        };
        eL = eff(128, 128, 1.00); eM = eff(128, 64, 0.95); eS = eff(64, 64, 0.88);
    }
    // tuning aid: per-shape tile overrides "M,N,K=L|M|S;..." (ACTMI_TILE_HINTS), e.g. "4808,512,512=S;9616,512,512=M"
    static const std::string hints = getenv("ACTMI_TILE_HINTS") ? getenv("ACTMI_TILE_HINTS") : "";
    if (!hints.empty() && a.tile_hint == 0) {
        char key[64];
        snprintf(key, sizeof(key), "%d,%d,%d=", a.M, a.N, a.K);
        const size_t pos = hints.find(key);
        if (pos != std::string::npos && (pos == 0 || hints[pos - 1] == ';')) {
            const char c = hints[pos + strlen(key)];
            if (c == 'L') return launch_cfg<128, 128, 64, 64, AMODE, BMODE, PREC>(a, st);
            if (c == 'M') return launch_cfg<128, 64, 64, 32, AMODE, BMODE, PREC>(a, st);
            if (c == 'S') return launch_cfg<64, 64, 32, 32, AMODE, BMODE, PREC>(a, st);
        }
    }
    static const char* force = getenv("ACTMI_GEMM_CFG");      // tuning aid: L / M / S
    if (force && force[0] == 'L') return launch_cfg<128, 128, 64, 64, AMODE, BMODE, PREC>(a, st);
    if (force && force[0] == 'M') return launch_cfg<128, 64, 64, 32, AMODE, BMODE, PREC>(a, st);
    if (force && force[0] == 'S') return launch_cfg<64, 64, 32, 32, AMODE, BMODE, PREC>(a, st);
    if (a.tile_hint == 1) return launch_cfg<128, 128, 64, 64, AMODE, BMODE, PREC>(a, st);
    if (a.tile_hint == 2) return launch_cfg<128, 64, 64, 32, AMODE, BMODE, PREC>(a, st);
    if (a.tile_hint == 3) return launch_cfg<64, 64, 32, 32, AMODE, BMODE, PREC>(a, st);
    if (eL >= eM && eL >= eS) return launch_cfg<128, 128, 64, 64, AMODE, BMODE, PREC>(a, st);
    if (eM >= eS) return launch_cfg<128, 64, 64, 32, AMODE, BMODE, PREC>(a, st);
    return launch_cfg<64, 64, 32, 32, AMODE, BMODE, PREC>(a, st);
}

}  // namespace

#ifdef ACTMI_GEMM_TU_BF16
// second translation unit (gemm_bf16.hip includes this file): the bf16 instantiations only, so that the two halves of the
// template zoo compile in parallel
int launch_gemm_bf16(int amode, int bmode, bool has_add, const GemmArgs& a, hipStream_t st) {
    int rc;
#define ACTMI_DISPATCH(PREC)                                                                                       \
    if (amode == A_N && bmode == B_N && has_add) rc = launch_modes<A_NADD, B_N, PREC>(a, st);                      \
    else if (amode == A_N && bmode == B_N) rc = launch_modes<A_N, B_N, PREC>(a, st);                               \
    else if (amode == A_CONV && bmode == B_N) rc = launch_modes<A_CONV, B_N, PREC>(a, st);                         \
    else if (amode == A_DGRAD && bmode == B_N) rc = launch_modes<A_DGRAD, B_N, PREC>(a, st);                       \
    else if (amode == A_N && bmode == B_T) rc = launch_modes<A_N, B_T, PREC>(a, st);                               \
    else if (amode == A_T && bmode == B_T) rc = launch_modes<A_T, B_T, PREC>(a, st);                               \
    else if (amode == A_T && bmode == B_WGRAD) rc = launch_modes<A_T, B_WGRAD, PREC>(a, st);                       \
    else if (amode == A_T && bmode == B_N) rc = launch_modes<A_T, B_N, PREC>(a, st);                               \
    else rc = -1002;
    ACTMI_DISPATCH(PREC_BF16)
#undef ACTMI_DISPATCH
    return rc;
}
#else
int launch_gemm_bf16(int amode, int bmode, bool has_add, const GemmArgs& a, hipStream_t st);

int launch_gemm(const GemmArgs& a_in, hipStream_t st, std::string* err) {
    GemmArgs a = a_in;
    if (a.groups <= 0) a.groups = 1;
    if (a.M <= 0 || a.N <= 0) return 0;
    auto fail = [&](const char* m) { if (err) *err = std::string("gemm: ") + m; return -2; };
    if (a.K <= 0) return fail("K must be positive");
    if (((uintptr_t)a.A & 15) || ((uintptr_t)a.Bw & 15)) return fail("A/B must be 16-byte aligned");
    if (a.splitk > 1 && (a.bias || a.res || a.mask || a.relu || a.C2 || a.scale)) return fail("split-K supports a plain accumulate only");
    if (a.split_stride != 0) {
        if (a.splitk <= 1) return fail("split_stride needs splitk > 1");
        const int nk = (a.K + BK - 1) / BK, tps = (nk + a.splitk - 1) / a.splitk;
        if ((a.splitk - 1) * tps >= nk) return fail("split_stride: every split must own at least one K tile");
    }
    if (a.relu == 2 && (a.rowmap || a.C2 || a.drop_p > 0.f || a.res_mod != 0)) return fail("GELU is only available in the plain epilogue");
    if (a.relu < 0 || a.relu > 2) return fail("bad activation");
    if (a.C2 && !a.scale2) return fail("C2 needs scale2");
    int amode, bmode = a.tb;
    if (a.ta) {
        if (a.mode != 0) return fail("ta requires mode 0");
        if (a.lda & 3) return fail("lda must be a multiple of 4");
        if (a.A_add || a.a_rowmap) return fail("addend / row gather not supported with ta");
        amode = A_T;
    } else {
        if (a.mode == 0) {
            if (a.lda & 3) return fail("lda must be a multiple of 4");
            // a ragged K is fine when the rows are padded to a multiple of 4 with finite values (the B side is
            // zero-filled beyond K, so the pad never contributes)
            if ((a.K & 3) && a.lda < ((a.K + 3) & ~3)) return fail("K % 4 != 0 needs rows padded to a multiple of 4");
            if (a.A_add && ((a.ld_add & 3) || a.add_mod <= 0 || ((uintptr_t)a.A_add & 15))) return fail("bad addend");
            if (a.A_alt && (a.A_add || a.a_rowmap || (a.alt_ncols % 128) || ((uintptr_t)a.A_alt & 15))) return fail("A_alt: no addend / row gather, alt_ncols a multiple of 128");
            amode = A_N;
        } else if (a.mode == 1) {
            if (a.K & 3) return fail("K must be a multiple of 4");
            if (a.Cin & 3) return fail("Cin must be a multiple of 4");
            if (a.Ax) {
                if (a.kx_begin != a.KH * a.KW * a.Cin || a.K != a.kx_begin + a.Cx) return fail("second source: K must be KH*KW*Cin + Cx, kx_begin = KH*KW*Cin");
                if ((a.Cin % BK) || (a.Cx % BK) || a.KH * a.KW > 32) return fail("second source: Cin and Cx must be multiples of 32");
                if (a.stride_x < 1 || (a.Ho - 1) * a.stride_x >= a.Hx || (a.Wo - 1) * a.stride_x >= a.Wx) return fail("second source: the strided pixel grid leaves the map");
                const int64_t nimg = a.M / ((int64_t)a.Ho * a.Wo);
                if (nimg * a.Hx * a.Wx * a.Cx >= ((int64_t)1 << 31) || (int64_t)a.H * a.W * a.Cin >= ((int64_t)1 << 30)) return fail("second source: map too large for 32-bit offsets");
                if ((uintptr_t)a.Ax & 15) return fail("second source must be 16-byte aligned");
            } else if (a.K != a.KH * a.KW * a.Cin) return fail("K != KH*KW*Cin");
            if (a.A_add || a.a_rowmap) return fail("addend not supported in conv mode");
            if (a.M % (a.Ho * a.Wo)) return fail("M must be images*Ho*Wo");
            if (a.k_tap_inner && ((a.Cin % BK) || a.KH * a.KW > 32 || (int64_t)a.H * a.W * a.Cin >= ((int64_t)1 << 30)))
                return fail("k_tap_inner needs Cin % 32 == 0 (the uniform-tap loader)");
            amode = A_CONV;
        } else if (a.mode == 2) {
            if (a.K % (a.KH * a.KW) || ((a.K / (a.KH * a.KW)) & 3)) return fail("dgrad: K must be KH*KW*Cout, Cout % 4 == 0");
            if (a.M % (a.H * a.W)) return fail("dgrad: M must be images*H*W");
            amode = A_DGRAD;
        } else return fail("bad mode");
    }
    if (bmode == 0) {
        if ((a.ldb & 3) || (a.K & 3)) return fail("ldb and K must be multiples of 4 for a contraction-contiguous B");
    } else if (bmode == 1) {
        if (a.ldb & 3) return fail("ldb must be a multiple of 4");
        if (a.B_add && ((a.ld_badd & 3) || a.badd_mod <= 0)) return fail("bad B addend");
    } else if (bmode == 2) {
        if ((a.Cin & 3) || a.N != a.KH * a.KW * a.Cin) return fail("wgrad: N must be KH*KW*Cin, Cin % 4 == 0");
        if (a.K % (a.Ho * a.Wo)) return fail("wgrad: K must be images*Ho*Wo");
        if (!a.ta) return fail("wgrad: A (dY) must be given in [m][n] storage (ta=1)");
    } else return fail("bad tb");
    int rc;
    // precision: explicit in the descriptor, else ACTMI_GEMM_PREC (f32 | f16x3), else the library default
    static const int env_prec = [] {
        const char* e = getenv("ACTMI_GEMM_PREC");
        if (!e) return ACTMI_PREC_DEFAULT_IS;
        return (e[0] == 'f' && e[1] == '3') ? ACTMI_PREC_F32 : ACTMI_PREC_F16X3;
    }();
    const int prec = a.prec ? a.prec : env_prec;
    if (prec != ACTMI_PREC_F32 && prec != ACTMI_PREC_F16X3 && prec != ACTMI_PREC_BF16) return fail("bad prec");
    if (a.b_split && (prec != ACTMI_PREC_F16X3 || bmode != B_N || !(amode == A_N || amode == A_CONV)))
        return fail("b_split needs prec f16x3 and the forward operand forms");
#define ACTMI_DISPATCH(PREC)                                                                                       \
    if (amode == A_N && bmode == B_N && a.A_add) rc = launch_modes<A_NADD, B_N, PREC>(a, st);                      \
    else if (amode == A_N && bmode == B_N) rc = launch_modes<A_N, B_N, PREC>(a, st);                               \
    else if (amode == A_CONV && bmode == B_N) rc = launch_modes<A_CONV, B_N, PREC>(a, st);                         \
    else if (amode == A_DGRAD && bmode == B_N) rc = launch_modes<A_DGRAD, B_N, PREC>(a, st);                       \
    else if (amode == A_N && bmode == B_T) rc = launch_modes<A_N, B_T, PREC>(a, st);                               \
    else if (amode == A_T && bmode == B_T) rc = launch_modes<A_T, B_T, PREC>(a, st);                               \
    else if (amode == A_T && bmode == B_WGRAD) rc = launch_modes<A_T, B_WGRAD, PREC>(a, st);                       \
    else if (amode == A_T && bmode == B_N) rc = launch_modes<A_T, B_N, PREC>(a, st);                               \
    else return fail("operand form combination not instantiated");
    if (prec == ACTMI_PREC_BF16) {
        if (a.Ax) return fail("a second source (Ax) needs prec f16x3");
        rc = launch_gemm_bf16(amode, bmode, a.A_add != nullptr, a, st);
        if (rc == -1002) return fail("operand form combination not instantiated");
    } else if (prec == ACTMI_PREC_F16X3) { ACTMI_DISPATCH(PREC_F16X3) } else { ACTMI_DISPATCH(PREC_F32) }
#undef ACTMI_DISPATCH
    if (rc == -1001) return fail("a second source (Ax) needs prec f16x3, pre-split weights (b_split), K % 32 == 0 and a plain epilogue");
    if (rc != 0 && err) *err = std::string("gemm launch: ") + hipGetErrorString((hipError_t)rc);
    return rc == 0 ? 0 : -3;
}
#endif  // ACTMI_GEMM_TU_BF16
