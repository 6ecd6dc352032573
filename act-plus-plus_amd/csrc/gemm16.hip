// gemm16: the forward GEMM / implicit-GEMM convolution of the inference path on PRE-SPLIT operands (gfx950).
//
// Same arithmetic as gemm.hip's PREC_F16X3 (every fp32 product = hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with fp32
// accumulation, operands split exactly into two fp16 pieces), but BOTH operands already live in memory in the split form
// ("s16": every aligned group of 8 consecutive k of a row is 32 bytes, [8 hi halfs][8 lo halfs] -- the bytes of the fp32
// row, so strides and addressing are those of the fp32 tensor).  Weights are split once at finalize, activations are
// written in this form by the epilogue that produces them.  Nothing is converted in the main loop and nothing is staged
// through registers:
//
//   * PERSISTENT workgroups, one per CU (8 waves), each walking its share of the output tiles.  The operand tiles of ALL its
//     (tile, K tile) steps form ONE continuous stream through a ring of NS LDS stages: the loads for the next output tile's
//     first K tiles are in flight while the current tile finishes, its epilogue runs beside them, and its stores drain under
//     the next tile's MFMAs.  (A one-tile-per-workgroup grid paid ~13 us of pipeline fill + store burst per tile against a
//     12 us K loop at K = 512, all CUs in lockstep: profiles/r02_gemm16_ablation.txt.)
//   * both operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4): one wave instruction moves 8 rows x 128 bytes (a
//     row's whole K tile of 32 fp32-k: full 128-byte lines), destination linear, XOR swizzle applied on the SOURCE chunk
//     index and again on the fragment reads (16 lanes of a ds_read_b128 group hit 16 distinct 16-byte slots);
//   * loads run NS-1 steps ahead across raw s_barriers with counted vmcnt (never drained inside the stream);
//   * the two waves that share a SIMD (wave w and w+4) run half a K tile apart ("ping-pong"): while one issues its MFMAs
//     (and its share of the DMA for a later step) the other reads its fragments; two barriers per K tile.
//
// Tile: BM x BN outputs, (BM, BN) = (128, 128), (256, 128) or (256, 256); wave (g, n) = (w >> 2, w & 3) owns rows
// [g*BM/2, (g+1)*BM/2) x columns [n*BN/4, (n+1)*BN/4): BM/64 x BN/128 MFMA tiles of 32x32, 3 * 2 * that many MFMAs per K tile
// of 32.  The 256 x 256 tile has room for two ring stages only (2 x 64 KB + 32 KB scratch): loads run ONE step ahead, a step
// being 48 MFMAs per wave.
// A forms: plain rows (nn.Linear, MHA projections, FFN, 1x1 input_proj: transformer.py:196-224, detr_vae.py:184) and the NHWC
// implicit im2col of the 3x3 / 1x1 ResNet convolutions with Cin % 32 == 0 (a K tile lies inside one filter tap; padding
// taps read a zero line).  Epilogue: acc * alpha * scale[n] + bias[n] (+ residual: s16 tensor or an f32 table indexed by
// row % res_mod) -> ReLU -> s16 (times the activation scale) or f32 rows, optional row scatter (token layout of input_proj).
//
// All LDS traffic of the kernel is inline asm: the compiler orders every LDS access it can see behind ALL outstanding
// LDS-DMA (s_waitcnt vmcnt(0)), which would drain the ring at every K tile; the DMA those accesses depend on is retired
// by the counted vmcnt + barrier protocol below.
#include "common.h"
#include "split16.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWB = 128;                    // bytes of one operand row per K tile (32 fp32-k)
constexpr int NTHR = 512;
constexpr int SCR_ROWS = 32;                 // epilogue scratch: one 32-row chunk of the tile at a time
constexpr int NUM_CU = 256;
#ifndef ACTMI_G16_GL
#define ACTMI_G16_GL 3
#endif

template <int BM, int BN> struct Cfg {
    // 128 x 128: 4 x 32 KB; 256 x 128: 3 x 48 KB; 256 x 256: 2 x 64 KB -- with the epilogue scratch (one 32-row chunk of the
    // tile: 16 / 32 KB) the two large tiles take exactly the CU's 160 KB
    static constexpr int NS = (BN == 256) ? 2 : (BM == 256) ? 3 : 4;
    static constexpr int STAGE = (BM + BN) * ROWB;
    static constexpr int SCR_BYTES = SCR_ROWS * BN * 4;
    static constexpr int GA = BM / 64;           // A-side DMA instructions per wave and K tile
    static constexpr int GB = BN / 64;           // B-side
    static constexpr int G = GA + GB;
    static constexpr int TM = BM / 64;           // 32x32 MFMA tiles per wave (rows)
    static constexpr int TN = BN / 128;          // (columns)
    static constexpr int NCH = BM / SCR_ROWS;    // epilogue chunks
    static constexpr int TPR = BN / 8;           // epilogue: threads per output row (8 floats each)
    static constexpr int RPP = NTHR / TPR;       // rows per pass
    static constexpr int NPASS = SCR_ROWS / RPP; // passes per chunk
    static constexpr int SMEM = NS * STAGE + SCR_BYTES;
};

template <int N> __device__ __forceinline__ void vmcnt_wait() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void lgkm_wait0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int OFF> __device__ __forceinline__ u32x4 lds_read16(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF> __device__ __forceinline__ void lds_write4(unsigned addr, float x) {
    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(x), "n"(OFF) : "memory");
}
__device__ __forceinline__ u32x4 glb_read16(const void* p) {           // a load the compiler does not see (manual vmcnt)
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void dma16(const void* src, unsigned char* lds_dst) {
    // 64 lanes x 16 bytes: LDS destination = wave-uniform base + lane * 16
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)lds_dst, 16, 0, 0);
}

struct TileId { int g, split, m0, n0; };
template <int V> using IC = std::integral_constant<int, V>;

template <int BM, int BN, int CONV>
__global__ __launch_bounds__(NTHR) void gemm16_kernel(Gemm16Args p, int tiles_m, int tiles_n, int total_tiles) {
    using C = Cfg<BM, BN>;
    constexpr int NS = C::NS, G = C::G, TM = C::TM, TN = C::TN, NPASS = C::NPASS;
    constexpr int SEPI = 2 * NPASS * C::NCH;                   // store instructions per wave in the epilogue of a full tile
    static_assert((NS - 2) * G + SEPI < 64, "vmcnt literal out of range");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
    const unsigned scr0 = lds0 + NS * C::STAGE;

    const int nwg_launch = gridDim.x, wg = blockIdx.x;
    const int splitk = p.splitk > 1 ? p.splitk : 1;
    const int nwg_tile = tiles_m * tiles_n;
    const int nk = (p.K / 32) / splitk;                  // K tiles per output tile (host: divisible)

    // ---- tile of this workgroup in round r.  Within a round the workgroups that share an XCD (equal wg % 8 under
    //      round-robin dispatch: a speed assumption only) take a contiguous run of tile ids, and tile ids walk bands of GM
    //      tile rows column by column: what an XCD runs at once is a compact patch of the output that fits its 4 MB L2.
    auto tile_of = [&](int r) -> TileId {
        const int first = r * nwg_launch;
        const int cnt = (total_tiles - first < nwg_launch) ? total_tiles - first : nwg_launch;       // tiles in this round
        const int xcd = wg & 7, q = cnt >> 3, rem = cnt & 7;
        const int base = (xcd < rem) ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
        const int flat = first + base + (wg >> 3);
        const int zz = flat / nwg_tile, bid = flat - zz * nwg_tile;
        TileId t;
        t.g = zz / splitk;
        t.split = zz - t.g * splitk;
        constexpr int GM = 4;
        const int width = GM * tiles_n;
        const int band = bid / width, f0 = band * GM;
        const int gsz = (tiles_m - f0 < GM) ? tiles_m - f0 : GM;
        const int in_band = bid - band * width;
        t.m0 = (f0 + in_band % gsz) * BM;
        t.n0 = (in_band / gsz) * BN;
        return t;
    };
    // rounds this workgroup takes part in (every round but possibly the last, ragged one)
    int nt = 0;
    for (int r = 0; r * nwg_launch < total_tiles; ++r) {
        const int first = r * nwg_launch;
        const int cnt = (total_tiles - first < nwg_launch) ? total_tiles - first : nwg_launch;
        const int xcd = wg & 7, q = cnt >> 3, rem = cnt & 7;
        if ((wg >> 3) < q + (xcd < rem ? 1 : 0)) ++nt;
    }
    if (nt == 0) return;                                 // workgroup-uniform
    if (p.stamps && wg == 0 && threadIdx.x == 0) p.stamps[0] = __builtin_amdgcn_s_memtime();

    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = wv >> 2, wn = wv & 3;
    const int li = lane & 31, lh = lane >> 5;

    // ---- DMA descriptors of the tile the ISSUE cursor is in.  Instruction q = wv + 8 j of a K tile covers tile rows
    //      8q .. 8q+7 (A rows first, then B rows); lane -> row 8q + lane/8, physical 16-byte chunk lane%8, which holds
    //      LOGICAL chunk (lane%8) ^ ((row/2)%8)
    const unsigned char* a_src[C::GA];
    int a_off[CONV ? C::GA : 1];
    unsigned a_mask[CONV ? C::GA : 1];
    const unsigned char* b_src[C::GB];
    int i_kt0 = 0;                                   // first K tile (of the whole contraction) of the issue cursor's tile
    const unsigned char* zero_line = reinterpret_cast<const unsigned char*>(p.zero_page) + (lane & 7) * 16;
    const int tpr = CONV ? p.Cin / 32 : 1;          // K tiles per filter tap
    unsigned conv_rep = 0;                          // bit r*KW set for every filter row
    if (CONV)
        for (int r = 0; r < p.KH; ++r) conv_rep |= 1u << (r * p.KW);

    auto setup = [&](const TileId& ti) {
        const unsigned char* Ab = reinterpret_cast<const unsigned char*>(p.A) + (int64_t)ti.g * p.gA * 4;
        const unsigned char* Bb = reinterpret_cast<const unsigned char*>(p.Bw) + (int64_t)ti.g * p.gB * 4;
        i_kt0 = ti.split * nk;
#pragma unroll
        for (int j = 0; j < C::GA; ++j) {
            const int row = (wv + 8 * j) * 8 + (lane >> 3);
            const int lch = (lane & 7) ^ ((row >> 1) & 7);
            int m = ti.m0 + row;
            m = m < p.M ? m : p.M - 1;               // rows past M replay row M-1: valid memory, never stored
            if (CONV) {
                const int hw = p.Ho * p.Wo;
                const int b = m / hw, rem = m - b * hw;
                const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
                // bit (r*KW+s): filter tap (r,s) of this output pixel lies inside the image (closed form)
                const int r_lo = hi0 < 0 ? -hi0 : 0, q_lo = wi0 < 0 ? -wi0 : 0;
                int r_hi = p.H - hi0, q_hi = p.W - wi0;
                r_hi = r_hi < p.KH ? (r_hi > 0 ? r_hi : 0) : p.KH;
                q_hi = q_hi < p.KW ? (q_hi > 0 ? q_hi : 0) : p.KW;
                const unsigned colmask = (r_lo < r_hi && q_lo < q_hi) ? (((1u << q_hi) - 1u) & ~((1u << q_lo) - 1u)) : 0u;
                const unsigned rowsel = (unsigned)((((uint64_t)1 << (r_hi * p.KW)) - 1u) & ~(((uint64_t)1 << (r_lo * p.KW)) - 1u));
                a_mask[j] = rowsel & (colmask * conv_rep);
                a_src[j] = Ab + (int64_t)b * p.img_stride * 4;
                a_off[j] = ((hi0 * p.W + wi0) * p.Cin) * 4 + lch * 16;
            } else {
                a_src[j] = Ab + (int64_t)m * p.lda * 4 + lch * 16;
            }
        }
#pragma unroll
        for (int j = 0; j < C::GB; ++j) {
            const int row = (wv + 8 * j) * 8 + (lane >> 3);
            const int lch = (lane & 7) ^ ((row >> 1) & 7);
            int n = ti.n0 + row;
            n = n < p.N ? n : p.N - 1;
            b_src[j] = Bb + (int64_t)n * p.ldb * 4 + lch * 16;
        }
    };
    // DMA instructions of one step are numbered 0 .. G-1 (A rows first); [J0, J1) selects a sub-range (the step's issue is
    // split between the L and the C phase)
    auto issue_part = [&](int kt, int stage, auto j0c, auto j1c) {
        constexpr int J0 = decltype(j0c)::value, J1 = decltype(j1c)::value;
        unsigned char* st = smem + stage * C::STAGE;
        const int ktg = i_kt0 + kt;
        if (CONV) {
            const int rs = ktg / tpr, cb = ktg - rs * tpr;
            const int r = rs / p.KW, s = rs - r * p.KW;
            const int delta = ((r * p.W + s) * p.Cin + cb * 32) * 4;
#pragma unroll
            for (int j = 0; j < C::GA; ++j) {
                if (j < J0 || j >= J1) continue;
                const bool inb = (a_mask[j] >> rs) & 1u;
                const unsigned char* src = inb ? a_src[j] + (a_off[j] + delta) : zero_line;
                dma16(src, st + (wv + 8 * j) * 1024);
            }
        } else {
#pragma unroll
            for (int j = 0; j < C::GA; ++j)
                if (j >= J0 && j < J1) dma16(a_src[j] + (int64_t)ktg * ROWB, st + (wv + 8 * j) * 1024);
        }
#pragma unroll
        for (int j = 0; j < C::GB; ++j)
            if (C::GA + j >= J0 && C::GA + j < J1) dma16(b_src[j] + (int64_t)ktg * ROWB, st + BM * ROWB + (wv + 8 * j) * 1024);
    };
    auto issue = [&](int kt, int stage) { issue_part(kt, stage, IC<0>{}, IC<C::G>{}); };

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
    const int b_row_off = BM * ROWB + (wn * (32 * TN) + li) * ROWB;

    // ---- the stream.  Step c = 0 .. nsteps-1 is (tile c / nk, K tile c % nk); its operands live in ring stage c % NS.
    //      Step c + NS-1 is issued by every wave during its step c: the first GL of its G DMA instructions at the head of
    //      L(c), the rest woven between the MFMAs of C(c) (target: the stage of step c-1, which both groups have read --
    //      group 1, one barrier behind, finished its L(c-1) before the barrier that opens group 0's L(c)).  A DMA
    //      instruction costs the issuing wave ~85 cycles with the SIMD's partner wave idle and ~170-190 beside its MFMA
    //      burst (in-kernel stamps, profiles/r02_gemm16_stamps.txt): all 6 in L made L (1650 cycles) twice the length of C
    //      (850), all 6 in C cost ~140 cycles of matrix-pipe bubble each; the 3 / 3 split measured best (2550 cycles per
    //      K tile of the 256-row tile = 60 % of the MFMA rate).  Before the barrier that precedes group 0's L(c+1) every
    //      wave has waited for its own share of step c+1: group 0 at the end of C(c), group 1 at the end of its L(c).
    const int nsteps = nt * nk;
    int i_step = 0, i_tile = 0, i_kt = 0;          // issue cursor
    {
        const int npre = NS - 1;
        for (int d = 0; d < npre && d < nsteps; ++d) {
            if (i_kt == 0) setup(tile_of(i_tile));
            issue(i_kt, i_step % NS);
            ++i_step;
            if (++i_kt == nk) { i_kt = 0; ++i_tile; }
        }
        const int pend = i_step - 1;                 // steps issued after step 0
        if (pend >= 3) vmcnt_wait<3 * G>();
        else if (pend == 2) vmcnt_wait<2 * G>();
        else if (pend == 1) vmcnt_wait<G>();
        else vmcnt_wait<0>();
    }
    __builtin_amdgcn_s_barrier();                     // step 0 complete in LDS
    uint64_t* stamp = (p.stamps && wg == 0 && t == 0) ? p.stamps : nullptr;
    if (stamp) { stamp[1] = __builtin_amdgcn_s_memtime(); stamp[62] = __builtin_amdgcn_s_memrealtime(); }

    f32x16 acc[TM][TN];
    u32x4 fa[2 / (TN > 1 ? 2 : 1)][TM][2], fb[2 / (TN > 1 ? 2 : 1)][TN][2];
    // steps whose wait must look past the SEPI stores of a full tile's epilogue: the DMA of the first NS-2 steps waited
    // for after an epilogue was issued BEFORE those stores (vmcnt retires in issue order)
    int store_debt = 0;

    auto wait_next_landed = [&](auto steady, int c) {
        constexpr bool STEADY = decltype(steady)::value;
        if (STEADY) {
            // exactly NS-2 steps of this wave are in flight behind step c+1
            if (store_debt > 0) vmcnt_wait<(NS - 2) * G + SEPI>();
            else vmcnt_wait<(NS - 2) * G>();
        } else if (c + 1 < nsteps) {
            // the stream is ending: count the steps behind c+1 (stores possibly among them only make this wait longer)
            const int pend = nsteps - 2 - c;
            if (pend >= 2) vmcnt_wait<2 * G>();
            else if (pend == 1) vmcnt_wait<G>();
            else vmcnt_wait<0>();
        }
    };
    // A step (one K tile of 32) runs as NH phase pairs L / C: NH = 1 reads both 16-deep halves' fragments in one L phase
    // (96 fragment registers at most with the 128-column tiles); the 256 x 256 tile (128 accumulator registers per wave,
    // 256 registers per wave at 2 waves per SIMD) takes the halves one at a time (NH = 2, 48 fragment registers live).
    constexpr int NH = (TN > 1) ? 2 : 1, KS = 2 / NH;
    // DMA instructions of the step issued in half h: all of them in half 0 by default (a full step of lead even with the
    // 2-stage ring); GL of a half's share at the head of its L phase, the rest woven into its C phase
#ifndef ACTMI_G16_H0
#define ACTMI_G16_H0 (C::G)
#endif
    constexpr int GH0 = (NH == 1) ? G : (ACTMI_G16_H0 < G ? ACTMI_G16_H0 : G);
    auto step = [&](int c, auto steady) {
        constexpr bool STEADY = decltype(steady)::value;
        const unsigned st = lds0 + (unsigned)(c % NS) * C::STAGE;
        auto half = [&](auto hc) {
            constexpr int H = decltype(hc)::value;
            constexpr int J0 = (H == 0) ? 0 : GH0, J1 = (H == 0) ? GH0 : G;        // this half's DMA instructions
            constexpr int GLH = (J1 - J0) < ACTMI_G16_GL ? (J1 - J0) : ACTMI_G16_GL;
            constexpr bool LAST = (H == NH - 1);
            // ---------------- L: this wave's share of the DMA for step c + NS - 1, then the fragments of this half.
            //                  Runs at priority 1, raised BEFORE the barrier that opens it: the partner wave on this SIMD is
            //                  then issuing MFMAs back to back, and a wave released from the barrier at equal priority did not
            //                  get to issue anything for the length of that burst (in-kernel stamps: 384 / 550 cycles)
            if (STEADY) {
                if (H == 0 && i_kt == 0) setup(tile_of(i_tile));           // the issue cursor enters a new tile (once per tile)
                issue_part(i_kt, i_step % NS, IC<J0>{}, IC<J0 + GLH>{});
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int sg = H * KS + s;                 // 16-deep half of the K tile
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const unsigned ab = st + a_row_off + co[sg][q], bb = st + b_row_off + co[sg][q];
                    fb[s][0][q] = lds_read16<0>(bb);
                    if (TN > 1) fb[s][1 % TN][q] = lds_read16<32 * ROWB>(bb);
                    fa[s][0][q] = lds_read16<0>(ab);
                    if (TM > 1) fa[s][1 % TM][q] = lds_read16<32 * ROWB>(ab);
                    if (TM > 2) fa[s][2 % TM][q] = lds_read16<64 * ROWB>(ab);
                    if (TM > 3) fa[s][3 % TM][q] = lds_read16<96 * ROWB>(ab);
                }
            }
            if (LAST && grp == 1) wait_next_landed(steady, c);
            lgkm_wait0();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---------------- C: the MFMAs of this half, back to back
            __builtin_amdgcn_s_setprio(0);
            if (STEADY) {
                issue_part(i_kt, i_step % NS, IC<J0 + GLH>{}, IC<J1>{});
                if (LAST) {
                    ++i_step;
                    if (++i_kt == nk) { i_kt = 0; ++i_tile; }
                }
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const h16x8 yh = __builtin_bit_cast(h16x8, fb[s][j][0]), yl = __builtin_bit_cast(h16x8, fb[s][j][1]);
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const h16x8 xh = __builtin_bit_cast(h16x8, fa[s][i][0]), xl = __builtin_bit_cast(h16x8, fa[s][i][1]);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, yh, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yl, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yh, acc[i][j], 0, 0, 0);
                    }
                }
            }
            if (STEADY && J1 - J0 - GLH > 0) {
                constexpr int NC = J1 - J0 - GLH, NMF = 3 * KS * TM * TN, PER = NMF / (NC + 1);
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);       // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x006, 8, 0);         // address arithmetic (VALU / SALU)
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);         // one LDS-DMA (VMEM read)
                }
            }
            __builtin_amdgcn_s_setprio(1);                       // for the L phase that follows the barrier below
            if (LAST) {
                if (grp == 0) wait_next_landed(steady, c);
                if (store_debt > 0) --store_debt;
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        };
        half(IC<0>{});
        if (NH > 1) half(IC<NH - 1>{});
    };

    // ---- epilogue of one finished tile (all 8 waves, aligned).  C layout of a 32x32 MFMA tile: col = lane & 31,
    //      row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5).  The tile leaves in 32-row chunks through a 16 KB LDS scratch as
    //      full rows: 16 lanes x 32 bytes = one 512-byte row segment per row.
    const int c8 = t % C::TPR, r0 = t / C::TPR;
    const unsigned scr_w = scr0 + (unsigned)((4 * lh) * BN + wn * (32 * TN) + li) * 4;       // + ((e&3) + 8(e>>2)) * BN * 4 + 128 j
    const unsigned scr_r = scr0 + (unsigned)(r0 * BN + c8 * 8) * 4;
    auto epilogue = [&](const TileId& ti, auto fullc) {
        constexpr bool FULL = decltype(fullc)::value;      // every row / column of the tile exists: store counts are literals
        const int n = ti.n0 + c8 * 8;
        const bool nok = n < p.N;
        const float alpha = p.alpha != 0.f ? p.alpha : 1.f;
        float sc[8], bi[8];
        {
            f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0, b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
            if (p.scale && nok) {
                const float* scale = p.scale + (int64_t)ti.g * p.gSB + n;
                s0 = *reinterpret_cast<const f32x4*>(scale); s1 = *reinterpret_cast<const f32x4*>(scale + 4);
            }
            if (p.bias && nok) {
                const float* bias = p.bias + (int64_t)ti.g * p.gSB + n;
                b0 = *reinterpret_cast<const f32x4*>(bias); b1 = *reinterpret_cast<const f32x4*>(bias + 4);
            }
            // the s16 output's scale folds into the affine (and into the residual's scale below)
            const float om = (p.c_fmt && p.c_scale != 0.f) ? p.c_scale : 1.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sc[e] = alpha * om * s0[e]; sc[4 + e] = alpha * om * s1[e];
                bi[e] = om * b0[e]; bi[4 + e] = om * b1[e];
            }
        }
        const unsigned char* res = p.res ? reinterpret_cast<const unsigned char*>(p.res) + (int64_t)ti.g * p.gRes * 4 : nullptr;
        unsigned char* Cb = reinterpret_cast<unsigned char*>(p.C) + ((int64_t)ti.g * p.gC + (int64_t)ti.split * p.split_stride) * 4;
        const float out_mul = (p.c_fmt && p.c_scale != 0.f) ? p.c_scale : 1.f;
        const float res_mul = (p.res_scale != 0.f ? p.res_scale : 1.f) * out_mul;
        const bool guard = p.c_fmt && p.flag != nullptr;
        float vmax = 0.f;
        u32x4 rbuf[2][NPASS][2];
        auto res_ptr = [&](int ch, int ps) {
            int m = ti.m0 + ch * SCR_ROWS + ps * C::RPP + r0;
            m = m < p.M ? m : p.M - 1;
            const int mr = p.res_mod ? m % p.res_mod : m;
            return res + ((int64_t)mr * p.ldres + (nok ? n : 0)) * 4;
        };
        auto res_load = [&](int ch) {
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const unsigned char* rp = res_ptr(ch, ps);
                rbuf[ch & 1][ps][0] = glb_read16(rp); rbuf[ch & 1][ps][1] = glb_read16(rp + 16);
            }
        };
        if (res) res_load(0);
#pragma unroll
        for (int ch = 0; ch < C::NCH; ++ch) {
            // the wave group that owns rows [32 ch, 32 ch + 32) writes its accumulator tiles into the scratch
            constexpr int CPG = C::NCH / 2;                     // chunks per group
            if (grp == ch / CPG) {
                const int i = ch % CPG;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const unsigned w = scr_w + j * 128;
                    auto wr = [&](auto ec) {            // immediate offset: row (e & 3) + 8 (e >> 2) of the chunk
                        constexpr int E = decltype(ec)::value;
                        lds_write4<((E & 3) + 8 * (E >> 2)) * BN * 4>(w, acc[i][j][E]);
                    };
                    wr(IC<0>{}); wr(IC<1>{}); wr(IC<2>{}); wr(IC<3>{}); wr(IC<4>{}); wr(IC<5>{}); wr(IC<6>{}); wr(IC<7>{});
                    wr(IC<8>{}); wr(IC<9>{}); wr(IC<10>{}); wr(IC<11>{}); wr(IC<12>{}); wr(IC<13>{}); wr(IC<14>{}); wr(IC<15>{});
                }
            }
            if (res && ch + 1 < C::NCH) res_load(ch + 1);
            lgkm_wait0();                                       // this wave's scratch writes have landed
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            u32x4 w0[NPASS], w1[NPASS];
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                w0[ps] = (ps == 0) ? lds_read16<0>(scr_r) : lds_read16<(NPASS > 1 ? C::RPP * BN * 4 : 0)>(scr_r);
                w1[ps] = (ps == 0) ? lds_read16<16>(scr_r) : lds_read16<(NPASS > 1 ? C::RPP * BN * 4 : 0) + 16>(scr_r);
            }
            if (res) {
                // residual of this chunk: younger vmem of this wave = the next chunk's 2 NPASS loads + the previous chunk's
                // 2 NPASS stores when every chunk stores (FULL); otherwise drain
                if (FULL) {
                    if (ch + 1 < C::NCH) { if (ch > 0) vmcnt_wait<4 * NPASS>(); else vmcnt_wait<2 * NPASS>(); }
                    else { if (ch > 0) vmcnt_wait<2 * NPASS>(); else vmcnt_wait<0>(); }
                } else vmcnt_wait<0>();
            }
            lgkm_wait0();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int m = ti.m0 + ch * SCR_ROWS + ps * C::RPP + r0;
                const f32x4 v0 = __builtin_bit_cast(f32x4, w0[ps]), v1 = __builtin_bit_cast(f32x4, w1[ps]);
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + bi[e];
                if (res) {
                    const u32x4 ra = rbuf[ch & 1][ps][0], rb = rbuf[ch & 1][ps][1];
                    if (p.res_fmt) {
                        const h16x8 rh = __builtin_bit_cast(h16x8, ra), rl = __builtin_bit_cast(h16x8, rb);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += ((float)rh[e] + (float)rl[e]) * res_mul;
                    } else {
                        const f32x4 fa4 = __builtin_bit_cast(f32x4, ra), fb4 = __builtin_bit_cast(f32x4, rb);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] += fa4[e] * out_mul; v[4 + e] += fb4[e] * out_mul; }
                    }
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (FULL || (m < p.M && nok)) {
                    const int64_t orow = p.rowmap ? p.rowmap[m] : m;
                    unsigned char* cp = Cb + (orow * p.ldc + n) * 4;
                    if (p.c_fmt) {
                        if (guard) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) vmax = fmaxf(vmax, fabsf(v[e]));
                        }
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
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();                        // every wave has read the scratch: free for the next chunk
        }
        // range guard of the split form: a stored value beyond the fp16 range (or not finite) raises the handle's flag
        if (guard && !(vmax < 65504.f)) atomicOr(p.flag, 1u);
    };

    // ---- main: per tile, stagger the groups in, run the K loop, re-align, epilogue
    int c = 0;
    for (int tile = 0; tile < nt; ++tile) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        if (grp == 1) __builtin_amdgcn_s_barrier();           // stagger: group 1's phases lag group 0's by one barrier
        const int c_end = c + nk;
        // two separate loops (one per flavour of the step): a single loop that picks the flavour per iteration made the
        // compiler give the flavours different accumulator registers and copy all of them on every back edge (64 v_mov =
        // ~550 cycles per K tile, seen in the stamps as a late start of every L phase)
        const int c_steady = (nsteps - (NS - 1) < c_end) ? nsteps - (NS - 1) : c_end;
        for (; c < c_steady; ++c) step(c, std::true_type{});
        for (; c < c_end; ++c) step(c, std::false_type{});
        if (grp == 0) __builtin_amdgcn_s_barrier();           // balances group 1's extra barrier: the groups are aligned
        if (stamp && tile < 29) stamp[2 + 2 * tile] = __builtin_amdgcn_s_memtime();
        const TileId ti = tile_of(tile);
        // FULL: every thread stores in every chunk and there is no row map (whose loads the compiler would wait for with
        // vmcnt(0)): the store count per wave is the literal SEPI
        const bool full = ti.m0 + BM <= p.M && ti.n0 + BN <= p.N && !p.rowmap;
        if (full && nk >= 2 * NS) {
            epilogue(ti, std::true_type{});
            store_debt = (c < nsteps) ? NS - 2 : 0;
        } else {
            if (full) epilogue(ti, std::true_type{});
            else epilogue(ti, std::false_type{});
            vmcnt_wait<0>();                                  // ragged tile / short K: drain, keep the bookkeeping trivial
            store_debt = 0;
        }
        if (stamp && tile < 29) stamp[3 + 2 * tile] = __builtin_amdgcn_s_memtime();
    }
    if (stamp) stamp[63] = __builtin_amdgcn_s_memrealtime() - stamp[62];
}

template <int BM, int BN, int CONV>
int launch_t(const Gemm16Args& a, hipStream_t st) {
    using C = Cfg<BM, BN>;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    auto kern = gemm16_kernel<BM, BN, CONV>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int splitk = a.splitk > 1 ? a.splitk : 1;
    const int groups = a.groups > 0 ? a.groups : 1;
    const int total = tiles_m * tiles_n * groups * splitk;
    const int nwg = total < NUM_CU ? total : NUM_CU;
    if (prof_enabled()) {
        char nm[128];
        static const bool by_shape = getenv("ACTMI_PROF_SHAPES") && getenv("ACTMI_PROF_SHAPES")[0] == '1';
        if (by_shape)
            snprintf(nm, sizeof(nm), "gemm16_kernel<%d,%d,%d>[M=%d,N=%d,K=%d,g=%d,sk=%d,wgs=%d]", BM, BN, CONV, a.M, a.N, a.K, groups, splitk,
                     total);
        else snprintf(nm, sizeof(nm), "gemm16_kernel<%d,%d,%d>", BM, BN, CONV);
        const double abytes = CONV ? (double)(a.M / (a.Ho * a.Wo)) * a.H * a.W * a.Cin : (double)a.M * a.K;
        prof_begin(nm, 2.0 * a.M * a.N * a.K * groups, 4.0 * groups * ((double)a.M * a.N + abytes + (double)a.N * a.K), st);
    }
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(NTHR), C::SMEM, st, a, tiles_m, tiles_n, total);
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
    constexpr int BN = 128;
    const long tn = (N + BN - 1) / BN;
    const long t128 = (long)((M + 127) / 128) * tn * z, t256 = (long)((M + 255) / 256) * tn * z;
    const double c128 = (double)((t128 + NUM_CU - 1) / NUM_CU) * 128 * 1.08, c256 = (double)((t256 + NUM_CU - 1) / NUM_CU) * 256;
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
        if (a.splitk > nk) return fail("every split must own at least one K tile");
        if (nk % a.splitk) return fail("every split must own the same number of K tiles (K / 32 divisible by splitk)");
    }
    // bm: 128 / 256 = rows of a 128-column tile; 512 = the 256 x 256 tile (half the operand bytes per MFMA of 128 x 128)
    const int bm = a.bm ? a.bm : gemm16_pick_bm(a.M, a.N, a.groups, a.splitk);
    int rc;
    if (bm == 512) rc = a.mode ? launch_t<256, 256, 1>(a, st) : launch_t<256, 256, 0>(a, st);
    else if (bm == 256) rc = a.mode ? launch_t<256, 128, 1>(a, st) : launch_t<256, 128, 0>(a, st);
    else if (bm == 128) rc = a.mode ? launch_t<128, 128, 1>(a, st) : launch_t<128, 128, 0>(a, st);
    else return fail("bm must be 0, 128, 256 or 512");
    if (rc != 0 && err) *err = std::string("gemm16 launch: ") + hipGetErrorString((hipError_t)rc);
    return rc == 0 ? 0 : -3;
}
