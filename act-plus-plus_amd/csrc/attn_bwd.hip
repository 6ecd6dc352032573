// Attention backward without the materialised score matrices (flash style), f16x3 products, for the training step's long
// self-attention calls (encoder: N = 1202 tokens, transformer.py:216-218 under autograd).  The path it replaces wrote
// P = softmax(S) and dS for every (sample, head) to HBM -- 2 x 3 GB per encoder layer at B = 64, written once and read three
// times -- and spent 25 ms of the step on launches bound by that traffic (profiles/r03_train_b64_shapes.json).
//
// Both kernels are the forward kernel of attn.hip turned around: a workgroup of four waves owns 128 rows of one operand (queries
// for dQ, keys for dK / dV), one row per lane; tiles of 64 rows of the other operand pass through LDS; the two score-shaped
// products  S^T = X_tile Y^T  and  T^T = X'_tile Y'^T  are computed TRANSPOSED so that one lane holds one owned row's column of
// both, the softmax weights are recomputed from the saved log-sum-exp (E = exp(S - lse), no running maximum), dS = E * (T - delta)
// * scale is formed in registers, and E / dS go from the accumulator registers straight into the B operand of the output
// products (contraction over the accumulator's row index; the LDS tile of the other operand is staged TRANSPOSED with the same
// slot permutation as V in the forward kernel).
//   dQ kernel (rows = queries):  S = K Q^T, T = V dO^T,  dQ^T += K^T dS
//   dK / dV kernel (rows = keys): S = Q K^T, T = dO V^T,  dV^T += dO^T E',  dK^T += Q^T dS         (E' = dropped weights)
// Seven products instead of five, no P / dS buffers, every output element written once by one lane: bitwise repeatable.
// Precision: dO enters pre-scaled by the power of two the backward's operand-scale machinery already computed for it; E is split
// at a fixed 2^10; dS -- whose magnitude is unknown until it exists -- is scaled per (owned row, 32-row step) by the power of two
// that brings its largest element to [2^13, 2^14), the product goes to a scratch accumulator and is added with the inverse scale.
#include "common.h"
#include "dropout.h"
#include "split16.h"
#include <cstdio>
#include <cstdlib>

namespace {

constexpr int KT = 64;                // rows of the streamed operand per LDS tile
constexpr int DQ_NT = 256;            // threads of the dQ workgroup (three workgroups per CU = three waves per SIMD; six-wave workgroups do not pack: 2.05 vs 1.27 ms)
constexpr int DKV_NT = 512;           // threads of the dK / dV workgroup: 8 waves = 256 owned keys share every staged query tile
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

struct AttnBwdK {
    const float *Q, *K, *V, *dO, *lse, *delta, *dO_scale;
    float *dQ, *dK, *dV;
    int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs;
    const uint8_t* kpm; int64_t kpm_bs;
    int B, H, Nq, Nk;
    float scale, drop_p; uint64_t drop_seed;
    unsigned* amax_out;
};

__device__ __forceinline__ void bwd_block_coords(int& bx, int& by, int& bz) {      // XCD-contiguous block order (attn.hip)
    const int gx = gridDim.x, gy = gridDim.y;
    const int total = gx * gy * (int)gridDim.z;
    const int lin = ((int)blockIdx.z * gy + (int)blockIdx.y) * gx + (int)blockIdx.x;
    const int xcd = lin & 7, q = total >> 3, r = total & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int flat = base + (lin >> 3);
    bx = flat % gx;
    const int rest = flat / gx;
    by = rest % gy;
    bz = rest / gy;
}

// one row of an operand (this lane's owned row): HD floats at `src` (+ lh * 8 per 16-deep step) times `sc`, as split MFMA B fragments
template <int NS>
__device__ __forceinline__ void load_row_frags(const float* src, float sc, h16x8 (&fh)[NS], h16x8 (&fl)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        f32x4 v0 = *reinterpret_cast<const f32x4*>(src + s * 16);
        f32x4 v1 = *reinterpret_cast<const f32x4*>(src + s * 16 + 4);
        v0 *= sc; v1 *= sc;
        uint2 h0, l0, h1, l1;
        split16(v0, h0, l0);
        split16(v1, h1, l1);
        fh[s] = __builtin_bit_cast(h16x8, uint4{h0.x, h0.y, h1.x, h1.y});
        fl[s] = __builtin_bit_cast(h16x8, uint4{l0.x, l0.y, l1.x, l1.y});
    }
}

// stage rows [r0, r0 + KT) of a [rows][HD] operand (row stride rs) as [row][HD hi halfs | HD lo halfs] (A operand of a score product)
template <int HD, int NT = 256>
__device__ __forceinline__ void stage_rows(unsigned char* dst, const float* base, int64_t rs, int r0, int r_end, float sc) {
    constexpr int C4 = HD / 4, KROW = 4 * HD + 16;
    constexpr int NL = (KT * C4 + NT - 1) / NT;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int e = threadIdx.x + NT * i;
        if (e < KT * C4) {
            const int kr = e / C4, c = e - kr * C4;
            const int row = r0 + kr;
            const bool ok = row < r_end;
            f32x4 v = *reinterpret_cast<const f32x4*>(base + (ok ? (int64_t)row * rs + c * 4 : 0));
            v = ok ? v * sc : z;
            uint2 hi, lo;
            split16(v, hi, lo);
            *reinterpret_cast<uint2*>(dst + kr * KROW + c * 8) = hi;
            *reinterpret_cast<uint2*>(dst + kr * KROW + 2 * HD + c * 8) = lo;
        }
    }
}

// the same rows TRANSPOSED: [d][KT rows hi | KT rows lo] with the rows of every 16-group in slot order (0-3, 8-11, 4-7, 12-15):
// the A operand of an output product whose B operand comes out of score-accumulator registers (attn.hip, V in the forward)
template <int HD, int NT = 256>
__device__ __forceinline__ void stage_transposed(unsigned char* dst, const float* base, int64_t rs, int r0, int r_end, float sc) {
    constexpr int C4 = HD / 4, VROW = 4 * KT + 16;
    constexpr int NL = (KT / 2 * C4 + NT - 1) / NT;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int e = threadIdx.x + NT * i;
        if (e < KT / 2 * C4) {
            const int c = ((e >> 5) % (C4 / 4)) * 4 + (e & 3);
            const int kp = ((e >> 5) / (C4 / 4)) * 8 + ((e >> 2) & 7);
            f32x4 pv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int row = r0 + 2 * kp + u;
                const bool ok = row < r_end;
                const f32x4 v = *reinterpret_cast<const f32x4*>(base + (ok ? (int64_t)row * rs + c * 4 : 0));
                pv[u] = ok ? v * sc : z;
            }
            uint2 ha, la, hb, lb;
            split16(pv[0], ha, la);
            split16(pv[1], hb, lb);
            const int kl = 2 * kp, r = kl & 15;
            const int pos = (kl & ~15) | (r & 3) | ((r & 8) >> 1) | ((r & 4) << 1);
            unsigned char* o = dst + (c * 4) * VROW + pos * 2;
            *reinterpret_cast<uint32_t*>(o + 0 * VROW) = (ha.x & 0xffffu) | (hb.x << 16);
            *reinterpret_cast<uint32_t*>(o + 1 * VROW) = (ha.x >> 16) | (hb.x & 0xffff0000u);
            *reinterpret_cast<uint32_t*>(o + 2 * VROW) = (ha.y & 0xffffu) | (hb.y << 16);
            *reinterpret_cast<uint32_t*>(o + 3 * VROW) = (ha.y >> 16) | (hb.y & 0xffff0000u);
            *reinterpret_cast<uint32_t*>(o + 0 * VROW + 2 * KT) = (la.x & 0xffffu) | (lb.x << 16);
            *reinterpret_cast<uint32_t*>(o + 1 * VROW + 2 * KT) = (la.x >> 16) | (lb.x & 0xffff0000u);
            *reinterpret_cast<uint32_t*>(o + 2 * VROW + 2 * KT) = (la.y & 0xffffu) | (lb.y << 16);
            *reinterpret_cast<uint32_t*>(o + 3 * VROW + 2 * KT) = (la.y >> 16) | (lb.y & 0xffff0000u);
        }
    }
}

// S[e] = sum over the head dim of (LDS rows sub*32 .. +32) x (this lane's row fragments): rows in the accumulator, lane's row the column
template <int HD>
__device__ __forceinline__ void score_tile(const unsigned char* rows, int sub, int li, int lh, const h16x8* fh, const h16x8* fl, f32x16& S) {
    constexpr int NS = HD / 16, KROW = 4 * HD + 16;
#pragma unroll
    for (int e = 0; e < 16; ++e) S[e] = 0.f;
    const unsigned char* r = rows + (sub * 32 + li) * KROW + lh * 16;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const h16x8 ah = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(r + s * 32));
        const h16x8 al = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(r + 2 * HD + s * 32));
        S = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, fh[s], S, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, fl[s], S, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, fh[s], S, 0, 0, 0);
    }
}

// 16 accumulator values (rows in slot order) -> split B fragments of the two 16-row groups
__device__ __forceinline__ void acc_to_frags(const f32x16& G, float sc, h16x8 (&gh)[2], h16x8 (&gl)[2]) {
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
        const f32x4 a0 = f32x4{G[8 * g2], G[8 * g2 + 1], G[8 * g2 + 2], G[8 * g2 + 3]} * sc;
        const f32x4 a1 = f32x4{G[8 * g2 + 4], G[8 * g2 + 5], G[8 * g2 + 6], G[8 * g2 + 7]} * sc;
        uint2 h0, l0, h1, l1;
        split16(a0, h0, l0);
        split16(a1, h1, l1);
        gh[g2] = __builtin_bit_cast(h16x8, uint4{h0.x, h0.y, h1.x, h1.y});
        gl[g2] = __builtin_bit_cast(h16x8, uint4{l0.x, l0.y, l1.x, l1.y});
    }
}

// acc (32 d rows x this lane's column) += (transposed LDS tile, d rows dt*32 .., rows sub*32 .. +32) x fragments
__device__ __forceinline__ void out_tile(const unsigned char* tr, int VROW, int dt, int sub, int li, int lh, const h16x8* gh, const h16x8* gl,
                                         f32x16& acc) {
    const unsigned char* r = tr + (dt * 32 + li) * VROW + sub * 64 + lh * 16;
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
        const h16x8 ah = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(r + g2 * 32));
        const h16x8 al = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(r + 2 * KT + g2 * 32));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, gh[g2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, gl[g2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, gh[g2], acc, 0, 0, 0);
    }
}

// power of two s with max * s in [2^13, 2^14) and its inverse, from the largest magnitude of a lane's column (both lane halves)
__device__ __forceinline__ void column_scale(const f32x16& G, float& s, float& inv) {
    float mx = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) mx = fmaxf(mx, fabsf(G[e]));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    int ex = (int)((__float_as_uint(mx) >> 23) & 0xffu);
    ex = ex < 40 ? 40 : (ex > 230 ? 230 : ex);           // zero / denormal columns: any scale will do; inf / nan: stays inf / nan
    s = __uint_as_float((unsigned)(267 - ex) << 23);      // 2^(13 - (ex - 127))
    inv = __uint_as_float((unsigned)(ex - 13) << 23);
}

__device__ __forceinline__ void raise_amax(unsigned* word, float mx) {
    if (!word) return;
    unsigned m = __float_as_uint(mx) & 0x7fffffffu;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(word, m);
}

// ------------------------------------------------------------------------------------------------------------ dQ
template <int HD>
__global__ __launch_bounds__(DQ_NT, 3) void attn_bwd_dq_kernel(AttnBwdK p) {
    constexpr int NS = HD / 16, DT = (HD + 31) / 32, VD = DT * 32;
    constexpr int KROW = 4 * HD + 16, VROW = 4 * KT + 16;
    __shared__ __attribute__((aligned(16))) unsigned char s_k[KT * KROW];
    __shared__ __attribute__((aligned(16))) unsigned char s_v[KT * KROW];
    __shared__ __attribute__((aligned(16))) unsigned char s_kt[VD * VROW];
    __shared__ __attribute__((aligned(16))) uint8_t s_dead[KT];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
    int bx, h, b;
    bwd_block_coords(bx, h, b);
    const int q = bx * (DQ_NT / 2) + wave * 32 + li;
    const bool qok = q < p.Nq;
    const int qs = qok ? q : 0;
    const float sdo = p.dO_scale ? *p.dO_scale : 1.f;
    const float qscale = p.scale * 1.4426950408889634f;
    h16x8 qh[NS], ql[NS], gh_[NS], gl_[NS];
    load_row_frags<NS>(p.Q + (int64_t)b * p.q_bs + (int64_t)qs * p.q_rs + h * HD + lh * 8, qok ? qscale : 0.f, qh, ql);
    load_row_frags<NS>(p.dO + (int64_t)b * p.do_bs + (int64_t)qs * p.do_rs + h * HD + lh * 8, qok ? sdo : 0.f, gh_, gl_);
    const int64_t rowid = ((int64_t)b * p.H + h) * p.Nq + qs;
    const float lse2 = p.lse[rowid] * 1.4426950408889634f;
    const float dlt = p.delta[rowid] * sdo;
    const float* Kb = p.K + (int64_t)b * p.k_bs + h * HD;
    const float* Vb = p.V + (int64_t)b * p.v_bs + h * HD;
    const uint8_t* kpm = p.kpm ? p.kpm + (int64_t)b * p.kpm_bs : nullptr;
    const float dsc = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;

    f32x16 acc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[d][e] = 0.f;
    if (HD < VD) {
        for (int e = t; e < VD * VROW / 4; e += DQ_NT) reinterpret_cast<uint32_t*>(s_kt)[e] = 0u;
        __syncthreads();
    }
    for (int kt0 = 0; kt0 < p.Nk; kt0 += KT) {
        stage_rows<HD, DQ_NT>(s_k, Kb, p.k_rs, kt0, p.Nk, 1.f);
        stage_rows<HD, DQ_NT>(s_v, Vb, p.v_rs, kt0, p.Nk, 1.f);
        stage_transposed<HD, DQ_NT>(s_kt, Kb, p.k_rs, kt0, p.Nk, 1.f);
        if (t < KT) {
            const int key = kt0 + t;
            s_dead[t] = (key >= p.Nk) || (kpm && kpm[key < p.Nk ? key : 0] != 0);
        }
        __syncthreads();
#pragma nounroll
        for (int sub = 0; sub < KT / 32; ++sub) {
            const int kb = kt0 + sub * 32;
            if (kb < p.Nk) {
                f32x16 S, T;
                score_tile<HD>(s_k, sub, li, lh, qh, ql, S);
                score_tile<HD>(s_v, sub, li, lh, gh_, gl_, T);
                uint32_t dead4[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) dead4[g] = *reinterpret_cast<const uint32_t*>(s_dead + sub * 32 + 8 * g + 4 * lh);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kr = sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    const float E = ((dead4[e >> 2] >> (8 * (e & 3))) & 0xffu) ? 0.f : __builtin_amdgcn_exp2f(S[e] - lse2);
                    float dp = T[e];
                    if (p.drop_p > 0.f) dp = actmi_keep(p.drop_seed, (uint64_t)rowid * (uint64_t)p.Nk + (uint64_t)(kt0 + kr), p.drop_p) ? dp * dsc : 0.f;
                    S[e] = E * (dp - dlt) * p.scale;            // dS, carrying the scale of dO
                }
                float gs, ginv;
                column_scale(S, gs, ginv);
                h16x8 fh[2], fl[2];
                acc_to_frags(S, gs, fh, fl);
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    f32x16 tmp;
#pragma unroll
                    for (int e = 0; e < 16; ++e) tmp[e] = 0.f;
                    out_tile(s_kt, VROW, d, sub, li, lh, fh, fl, tmp);
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[d][e] = fmaf(tmp[e], ginv, acc[d][e]);
                }
            }
        }
        __syncthreads();
    }
    const float un = 1.f / sdo;
    float mx = 0.f;
    if (qok) {
        float* o = p.dQ + (int64_t)b * p.dq_bs + (int64_t)q * p.dq_rs + h * HD;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = d * 32 + 8 * g + 4 * lh;
                if (d0 < HD) {
                    const f32x4 v = f32x4{acc[d][4 * g], acc[d][4 * g + 1], acc[d][4 * g + 2], acc[d][4 * g + 3]} * un;
                    *reinterpret_cast<f32x4*>(o + d0) = v;
                    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                }
            }
    }
    raise_amax(p.amax_out, mx);
}

// ------------------------------------------------------------------------------------------------------- dK, dV
template <int HD>
__global__ __launch_bounds__(DKV_NT, 2) void attn_bwd_dkv_kernel(AttnBwdK p) {
    constexpr int NS = HD / 16, DT = (HD + 31) / 32, VD = DT * 32;
    constexpr int KROW = 4 * HD + 16, VROW = 4 * KT + 16;
    constexpr float ESC = 1024.f;                       // fixed split scale of the (dropped) softmax weights
    extern __shared__ __attribute__((aligned(16))) unsigned char dkv_smem[];        // 70 KB at HD = 64: above the static limit
    unsigned char* s_q = dkv_smem;
    unsigned char* s_g = s_q + KT * KROW;
    unsigned char* s_qt = s_g + KT * KROW;
    unsigned char* s_gt = s_qt + VD * VROW;
    float* s_lse = reinterpret_cast<float*>(s_gt + VD * VROW);
    float* s_dlt = s_lse + KT;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
    int bx, h, b;
    bwd_block_coords(bx, h, b);
    const int key = bx * (DKV_NT / 2) + wave * 32 + li;
    const bool kok = key < p.Nk;
    const int ks = kok ? key : 0;
    const uint8_t* kpm = p.kpm ? p.kpm + (int64_t)b * p.kpm_bs : nullptr;
    const bool kdead = !kok || (kpm && kpm[ks] != 0);
    const float sdo = p.dO_scale ? *p.dO_scale : 1.f;
    const float qscale = p.scale * 1.4426950408889634f;
    h16x8 kh[NS], kl[NS], vh[NS], vl[NS];
    load_row_frags<NS>(p.K + (int64_t)b * p.k_bs + (int64_t)ks * p.k_rs + h * HD + lh * 8, kok ? qscale : 0.f, kh, kl);
    load_row_frags<NS>(p.V + (int64_t)b * p.v_bs + (int64_t)ks * p.v_rs + h * HD + lh * 8, kok ? 1.f : 0.f, vh, vl);
    const float* Qb = p.Q + (int64_t)b * p.q_bs + h * HD;
    const float* Gb = p.dO + (int64_t)b * p.do_bs + h * HD;
    const int64_t row0 = ((int64_t)b * p.H + h) * p.Nq;
    const float dsc = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;

    f32x16 aK[DT], aV[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) { aK[d][e] = 0.f; aV[d][e] = 0.f; }
    if (HD < VD) {
        for (int e = t; e < VD * VROW / 4; e += DKV_NT) { reinterpret_cast<uint32_t*>(s_qt)[e] = 0u; reinterpret_cast<uint32_t*>(s_gt)[e] = 0u; }
        __syncthreads();
    }
    for (int qt0 = 0; qt0 < p.Nq; qt0 += KT) {
        stage_rows<HD, DKV_NT>(s_q, Qb, p.q_rs, qt0, p.Nq, 1.f);
        stage_rows<HD, DKV_NT>(s_g, Gb, p.do_rs, qt0, p.Nq, sdo);
        stage_transposed<HD, DKV_NT>(s_qt, Qb, p.q_rs, qt0, p.Nq, 1.f);
        stage_transposed<HD, DKV_NT>(s_gt, Gb, p.do_rs, qt0, p.Nq, sdo);
        if (t < KT) {
            const int qq = qt0 + t;
            const bool ok = qq < p.Nq;
            s_lse[t] = ok ? p.lse[row0 + qq] * 1.4426950408889634f : INFINITY;       // exp2(S - inf) = 0: rows beyond Nq vanish
            s_dlt[t] = ok ? p.delta[row0 + qq] * sdo : 0.f;
        }
        __syncthreads();
#pragma nounroll
        for (int sub = 0; sub < KT / 32; ++sub) {
            const int qb = qt0 + sub * 32;
            if (qb < p.Nq) {
                f32x16 S, T;
                score_tile<HD>(s_q, sub, li, lh, kh, kl, S);
                score_tile<HD>(s_g, sub, li, lh, vh, vl, T);
                f32x16 Ed;
#pragma unroll
                for (int g = 0; g < 4; ++g) {           // the 16 rows of this lane: four groups of four consecutive rows
                    const f32x4 lse4 = *reinterpret_cast<const f32x4*>(s_lse + sub * 32 + 8 * g + 4 * lh);
                    const f32x4 dlt4 = *reinterpret_cast<const f32x4*>(s_dlt + sub * 32 + 8 * g + 4 * lh);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 4 * g + j;
                        const int qr = sub * 32 + j + 8 * g + 4 * lh;
                        const float E = kdead ? 0.f : __builtin_amdgcn_exp2f(S[e] - lse4[j]);
                        float dp = T[e], ed = E;
                        if (p.drop_p > 0.f) {
                            const int qq = qt0 + qr < p.Nq ? qt0 + qr : 0;
                            const bool keep = actmi_keep(p.drop_seed, (uint64_t)(row0 + qq) * (uint64_t)p.Nk + (uint64_t)ks, p.drop_p);
                            dp = keep ? dp * dsc : 0.f;
                            ed = keep ? E * dsc : 0.f;
                        }
                        Ed[e] = ed;
                        S[e] = E * (dp - dlt4[j]) * p.scale;
                    }
                }
                h16x8 fh[2], fl[2];
                acc_to_frags(Ed, ESC, fh, fl);
#pragma unroll
                for (int d = 0; d < DT; ++d) out_tile(s_gt, VROW, d, sub, li, lh, fh, fl, aV[d]);
                float gs, ginv;
                column_scale(S, gs, ginv);
                acc_to_frags(S, gs, fh, fl);
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    f32x16 tmp;
#pragma unroll
                    for (int e = 0; e < 16; ++e) tmp[e] = 0.f;
                    out_tile(s_qt, VROW, d, sub, li, lh, fh, fl, tmp);
#pragma unroll
                    for (int e = 0; e < 16; ++e) aK[d][e] = fmaf(tmp[e], ginv, aK[d][e]);
                }
            }
        }
        __syncthreads();
    }
    const float unk = 1.f / sdo, unv = 1.f / (sdo * ESC);
    float mx = 0.f;
    if (kok) {
        float* ok_ = p.dK + (int64_t)b * p.dk_bs + (int64_t)key * p.dk_rs + h * HD;
        float* ov = p.dV + (int64_t)b * p.dv_bs + (int64_t)key * p.dv_rs + h * HD;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = d * 32 + 8 * g + 4 * lh;
                if (d0 < HD) {
                    const f32x4 a = f32x4{aK[d][4 * g], aK[d][4 * g + 1], aK[d][4 * g + 2], aK[d][4 * g + 3]} * unk;
                    const f32x4 c = f32x4{aV[d][4 * g], aV[d][4 * g + 1], aV[d][4 * g + 2], aV[d][4 * g + 3]} * unv;
                    *reinterpret_cast<f32x4*>(ok_ + d0) = a;
                    *reinterpret_cast<f32x4*>(ov + d0) = c;
                    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3]))));
                    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(c[0]), fabsf(c[1])), fmaxf(fabsf(c[2]), fabsf(c[3]))));
                }
            }
    }
    raise_amax(p.amax_out, mx);
}

}  // namespace

int launch_attention_bwd(const AttnBwdArgs& a, hipStream_t st, std::string* err) {
    auto fail = [&](const char* m) { if (err) *err = std::string("attention backward: ") + m; return -2; };
    if (a.B <= 0 || a.Nq <= 0 || a.Nk <= 0) return 0;
    if (a.HD != 64 && a.HD != 32 && a.HD != 16) return fail("head_dim must be 16, 32 or 64");
    const int64_t strides[] = {a.q_bs, a.q_rs, a.k_bs, a.k_rs, a.v_bs, a.v_rs, a.do_bs, a.do_rs, a.dq_bs, a.dq_rs, a.dk_bs, a.dk_rs, a.dv_bs, a.dv_rs};
    for (int64_t s : strides) if (s & 3) return fail("strides must be multiples of 4 floats");
    const void* ptrs[] = {a.Q, a.K, a.V, a.dO, a.dQ, a.dK, a.dV};
    for (const void* q : ptrs) if (!q || ((uintptr_t)q & 15)) return fail("pointers must be non-null and 16-byte aligned");
    if (!a.lse || !a.delta) return fail("lse and delta are required");
    AttnBwdK k{a.Q, a.K, a.V, a.dO, a.lse, a.delta, a.dO_scale, a.dQ, a.dK, a.dV, a.q_bs, a.q_rs, a.k_bs, a.k_rs, a.v_bs, a.v_rs,
               a.do_bs, a.do_rs, a.dq_bs, a.dq_rs, a.dk_bs, a.dk_rs, a.dv_bs, a.dv_rs, a.kpm, a.kpm_bs, a.B, a.H, a.Nq, a.Nk,
               a.scale, a.drop_p, a.drop_seed, a.amax_out};
    const dim3 gq((a.Nq + DQ_NT / 2 - 1) / (DQ_NT / 2), a.H, a.B), gk((a.Nk + DKV_NT / 2 - 1) / (DKV_NT / 2), a.H, a.B);
    const double f1 = 2.0 * a.B * a.H * (double)a.Nq * a.Nk * a.HD;
    prof_begin("attn_bwd_dq_kernel", 3.0 * f1, 4.0 * a.B * a.H * a.HD * (3.0 * a.Nq + 2.0 * a.Nk), st);
    switch (a.HD) {
        case 64: hipLaunchKernelGGL(attn_bwd_dq_kernel<64>, gq, dim3(DQ_NT), 0, st, k); break;
        case 32: hipLaunchKernelGGL(attn_bwd_dq_kernel<32>, gq, dim3(DQ_NT), 0, st, k); break;
        default: hipLaunchKernelGGL(attn_bwd_dq_kernel<16>, gq, dim3(DQ_NT), 0, st, k); break;
    }
    prof_end(st);
    prof_begin("attn_bwd_dkv_kernel", 4.0 * f1, 4.0 * a.B * a.H * a.HD * (2.0 * a.Nq + 4.0 * a.Nk), st);
    {
        const int VD = ((a.HD + 31) / 32) * 32;
        const size_t lds = (size_t)2 * KT * (4 * a.HD + 16) + (size_t)2 * VD * (4 * KT + 16) + 2 * KT * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            const int cap = 72 * 1024;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess)
                return fail("cannot raise the dynamic LDS limit");
            attr_set = true;
        }
        switch (a.HD) {
            case 64: hipLaunchKernelGGL(attn_bwd_dkv_kernel<64>, gk, dim3(DKV_NT), lds, st, k); break;
            case 32: hipLaunchKernelGGL(attn_bwd_dkv_kernel<32>, gk, dim3(DKV_NT), lds, st, k); break;
            default: hipLaunchKernelGGL(attn_bwd_dkv_kernel<16>, gk, dim3(DKV_NT), lds, st, k); break;
        }
    }
    prof_end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { if (err) *err = std::string("attention backward launch: ") + hipGetErrorString(e); return -3; }
    return 0;
}
