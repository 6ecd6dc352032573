// Exact-f32 multi-head attention, flash style (scores never leave registers), for gfx950.
// Reference semantics: nn.MultiheadAttention as called at transformer.py:217-218 (encoder self-attention,
// N=2+15*20*C tokens), :286-289 (decoder cross-attention, 100 queries x N keys) and the CVAE encoder
// (detr_vae.py:133, 102 tokens with a key-padding mask): softmax(q k^T / sqrt(hd)) v per head.
//
// Mapping to CDNA4: a block is 4 waves, each wave owns 32 query rows; K/V tiles of 64 keys are staged in LDS
// and shared by the 4 waves.  The score tile is computed TRANSPOSED, S^T[key][q] = sum_d K[key][d] Q[q][d], with
// v_mfma_f32_32x32x2_f32: the C/D layout then puts one query per lane (column = lane&31) and 16 of the 32 keys
// in that lane's registers (the other 16 in lane^32), so the row max / row sum of the softmax are 15
// in-register ops plus one cross-half exchange, and the probabilities are already the B operand of the second
// product O^T[d][q] = sum_key V[key][d] P^T[key][q] (contraction over the accumulator's ROW index needs no lane
// movement).  The head-dim contraction order is free, so lane-half h takes d = h*HD/2 + s at step s and reads
// K rows as ds_read_b128 (row stride HD+4 floats: conflict-free); V is read as ds_read_b32 along d.
#include "common.h"
#include "dropout.h"
#include "split16.h"
#include <cstdio>
#include <cstdlib>

namespace {

// f16x3 kernel: the next K/V tile is NOT prefetched into registers -- without those 48 VGPRs the kernel fits 168 and a third
// wave per SIMD hides the global latency instead (measured 136 -> 116 us on the encoder shape; with the prefetch kept,
// three waves spill).
constexpr bool ATTN_NOPF = true;
constexpr int KT = 64;   // keys per LDS tile
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

// XCD-aware block order: the hardware deals consecutive block ids round-robin over the 8 XCDs (each with its own 4 MB L2), so
// the query blocks of one (batch, head) -- which all stream the same K / V -- used to land on all eight and each L2 fetched
// that K / V again (201 MB of fabric reads per encoder launch against 79 MB of operands, PMC round 3).  The bijective remap
// gives every XCD a contiguous run of (batch, head, query block) triples, as gemm.hip does for its tiles.
__device__ __forceinline__ void attn_block_coords(int& bx, int& by, int& bz) {
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

template <int HD>
__global__ __launch_bounds__(256) void attn_f32_kernel(AttnArgs p, int nsplit, int chunk) {
    constexpr int HH = HD / 2;            // k-steps of the QK^T product per lane half
    constexpr int DT = (HD + 31) / 32;    // 32-wide output tiles along d
    constexpr int VD = DT * 32;
    constexpr int KS = HD + 4;
    __shared__ __attribute__((aligned(16))) float s_k[KT * KS];
    __shared__ __attribute__((aligned(16))) float s_v[KT * VD];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    int bx, h, b;
    attn_block_coords(bx, h, b);
    const int split = bx % nsplit;
    const int q = (bx / nsplit) * 128 + wave * 32 + li;
    const int k_begin = split * chunk;
    const int k_end = (k_begin + chunk < p.Nk) ? k_begin + chunk : p.Nk;      // this block's key range
    const bool qok = q < p.Nq;

    // scores are kept in the base-2 domain: s' = s * log2(e), p = 2^(s' - m'), so the exponential is one v_exp_f32
    const float qscale = p.scale * 1.4426950408889634f;
    const float* Qp = p.Q + (int64_t)b * p.q_bs + (int64_t)(qok ? q : 0) * p.q_rs + h * HD + lh * HH;
    float qreg[HH];
#pragma unroll
    for (int s = 0; s < HH; s += 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(Qp + s);
#pragma unroll
        for (int j = 0; j < 4; ++j) qreg[s + j] = qok ? v[j] * qscale : 0.f;
    }
    const float* Kb = p.K + (int64_t)b * p.k_bs + h * HD;
    const float* Vb = p.V + (int64_t)b * p.v_bs + h * HD;
    const uint8_t* kpm = p.kpm ? p.kpm + (int64_t)b * p.kpm_bs : nullptr;

    f32x16 O[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[d][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // zero the V pad columns once (HD < VD only for small heads)
    if (HD < VD) {
        for (int e = t; e < KT * VD; e += 256) s_v[e] = 0.f;
        __syncthreads();
    }

    __shared__ uint8_t s_dead[KT];
    // software pipeline: the K/V rows of tile t+1 are fetched into registers while tile t is being consumed
    constexpr int C4 = HD / 4;
    constexpr int NLD = (KT * C4 + 255) / 256;
    f32x4 pk[NLD], pv[NLD];
    auto fetch = [&](int kt0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = t + 256 * i;
            const int kr = e / C4, c = e - kr * C4;
            const int key = kt0 + kr;
            const bool ok = e < KT * C4 && key < k_end;
            const int64_t koff = ok ? (int64_t)key * p.k_rs + c * 4 : 0;
            const int64_t voff = ok ? (int64_t)key * p.v_rs + c * 4 : 0;
            const f32x4 kv = *reinterpret_cast<const f32x4*>(Kb + koff);
            const f32x4 vv = *reinterpret_cast<const f32x4*>(Vb + voff);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            pk[i] = ok ? kv : z;
            pv[i] = ok ? vv : z;
        }
    };
    fetch(k_begin);
    for (int kt0 = k_begin; kt0 < k_end; kt0 += KT) {
        // ---- stage the prefetched tile
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = t + 256 * i;
            if (e < KT * C4) {
                const int kr = e / C4, c = e - kr * C4;
                *reinterpret_cast<f32x4*>(&s_k[kr * KS + c * 4]) = pk[i];
                *reinterpret_cast<f32x4*>(&s_v[kr * VD + c * 4]) = pv[i];
            }
        }
        const bool ragged = kt0 + KT > k_end || kpm != nullptr;    // block-uniform: masking only where needed
        if (ragged && t < KT) {
            const int key = kt0 + t;
            s_dead[t] = (key >= k_end) || (kpm && kpm[key] != 0);
        }
        __syncthreads();
        if (kt0 + KT < k_end) fetch(kt0 + KT);
#pragma unroll
        for (int sub = 0; sub < KT / 32; ++sub) {
            const int kb = kt0 + sub * 32;
            if (kb < k_end) {
                f32x16 S;
#pragma unroll
                for (int e = 0; e < 16; ++e) S[e] = 0.f;
                const float* krow = &s_k[(sub * 32 + li) * KS + lh * HH];
#pragma unroll
                for (int s = 0; s < HH; s += 4) {
                    const f32x4 kv = *reinterpret_cast<const f32x4*>(krow + s);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        S = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[j], qreg[s + j], S, 0, 0, 0);
                }
                if (ragged) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (s_dead[sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh]) S[e] = -INFINITY;
                }
                if (p.causal) {          // key j is visible to query i only for j <= i (nn.MultiheadAttention attn_mask = triu(1))
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (kb + (e & 3) + 8 * (e >> 2) + 4 * lh > q) S[e] = -INFINITY;
                }
                float mt = S[0];
#pragma unroll
                for (int e = 1; e < 16; ++e) mt = fmaxf(mt, S[e]);
                mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
                const float m_new = fmaxf(m_run, mt);
                const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
                float rs = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    S[e] = __builtin_amdgcn_exp2f(S[e] - m_safe);
                    rs += S[e];
                }
                rs += __shfl_xor(rs, 32, 64);
                l_run = l_run * alpha + rs;          // the normaliser uses the UN-dropped weights (dropout acts on softmax output)
                m_run = m_new;
                if (p.drop_p > 0.f) {
                    const float ds = 1.f / (1.f - p.drop_p);
                    const uint64_t rowbase = (((uint64_t)b * p.H + h) * p.Nq + (qok ? q : 0)) * (uint64_t)p.Nk;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = kb + (e & 3) + 8 * (e >> 2) + 4 * lh;
                        S[e] = actmi_keep(p.drop_seed, rowbase + key, p.drop_p) ? S[e] * ds : 0.f;
                    }
                }
#pragma unroll
                for (int d = 0; d < DT; ++d)
#pragma unroll
                    for (int e = 0; e < 16; ++e) O[d][e] *= alpha;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kl = sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
#pragma unroll
                    for (int d = 0; d < DT; ++d) {
                        const float vv = s_v[kl * VD + d * 32 + li];
                        O[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, S[e], O[d], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (qok) {
        if (nsplit == 1) {
            const float inv = 1.f / l_run;
            float* Op = p.O + (int64_t)b * p.o_bs + (int64_t)q * p.o_rs + h * HD;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = d * 32 + 8 * g + 4 * lh;
                    if (d0 < HD) {
                        f32x4 o = {O[d][4 * g] * inv, O[d][4 * g + 1] * inv, O[d][4 * g + 2] * inv, O[d][4 * g + 3] * inv};
                        *reinterpret_cast<f32x4*>(Op + d0) = o;
                    }
                }
            if (p.lse && lh == 0) p.lse[((int64_t)b * p.H + h) * p.Nq + q] = m_run * 0.6931471805599453f + logf(l_run);
        } else {
            // partial result of this key range: un-normalised O plus (running max, running sum)
            const int64_t row = (((int64_t)split * p.B + b) * p.H + h) * p.Nq + q;
            float* Op = p.ws + row * HD;
            float* ml = p.ws + (int64_t)nsplit * p.B * p.H * p.Nq * HD + row * 2;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = d * 32 + 8 * g + 4 * lh;
                    if (d0 < HD) {
                        f32x4 o = {O[d][4 * g], O[d][4 * g + 1], O[d][4 * g + 2], O[d][4 * g + 3]};
                        *reinterpret_cast<f32x4*>(Op + d0) = o;
                    }
                }
            if (lh == 0) { ml[0] = m_run; ml[1] = l_run; }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// fp16-split variant (same mapping, same outputs): every fp32 product of both matrix products is formed from three
// v_mfma_f32_32x32x16_f16 products of exactly split operands (hi = rn16(x), lo = rn16(x - hi); see gemm.hip PREC_F16X3).
// K is staged as [key][hd hi halfs | hd lo halfs]; V is staged TRANSPOSED, [d][64 keys hi | 64 keys lo], with the keys
// of every 16-group permuted (0-3, 8-11, 4-7, 12-15) so that the 8 keys a lane half owns in the score accumulator
// (rows (e&3) + 8(e>>2) + 4*half) are 16 contiguous bytes: the probabilities go from accumulator registers straight
// into the B operand of the second product, as in the fp32 kernel.  A 32-key step is 24 MFMAs of 32 cycles per wave
// (768) against 64 of 64 cycles (4096) for the fp32 instruction.
template <int HD>
__global__ __launch_bounds__(256, 3) void attn_f16x3_kernel(AttnArgs p, int nsplit, int chunk) {
    constexpr int NS = HD / 16;           // 16-deep steps of the QK^T contraction
    constexpr int DT = (HD + 31) / 32;    // 32-wide output tiles along d
    constexpr int VD = DT * 32;
    constexpr int KROW = 4 * HD + 16;     // bytes per key row of s_k  (hd hi halfs, hd lo halfs, pad: conflict-free b128)
    constexpr int VROW = 4 * KT + 16;     // bytes per d row of s_vt   (64 keys hi, 64 keys lo, pad)
    __shared__ __attribute__((aligned(16))) unsigned char s_k[KT * KROW];
    __shared__ __attribute__((aligned(16))) unsigned char s_vt[VD * VROW];
    __shared__ uint8_t s_dead[KT];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    int bx, h, b;
    attn_block_coords(bx, h, b);
    const int split = bx % nsplit;
    const int q = (bx / nsplit) * 128 + wave * 32 + li;
    const int k_begin = split * chunk;
    const int k_end = (k_begin + chunk < p.Nk) ? k_begin + chunk : p.Nk;
    const bool qok = q < p.Nq;

    // scores in the base-2 domain (one v_exp_f32 per probability); q carries the scale into its split
    const float qscale = p.scale * 1.4426950408889634f;
    const float* Qp = p.Q + (int64_t)b * p.q_bs + (int64_t)(qok ? q : 0) * p.q_rs + h * HD + lh * 8;
    h16x8 qh[NS], ql[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        f32x4 v0 = *reinterpret_cast<const f32x4*>(Qp + s * 16);
        f32x4 v1 = *reinterpret_cast<const f32x4*>(Qp + s * 16 + 4);
        const float sc = qok ? qscale : 0.f;
        v0 *= sc; v1 *= sc;
        uint2 h0, l0, h1, l1;
        split16(v0, h0, l0);
        split16(v1, h1, l1);
        qh[s] = __builtin_bit_cast(h16x8, uint4{h0.x, h0.y, h1.x, h1.y});
        ql[s] = __builtin_bit_cast(h16x8, uint4{l0.x, l0.y, l1.x, l1.y});
    }
    const float* Kb = p.K + (int64_t)b * p.k_bs + h * HD;
    const float* Vb = p.V + (int64_t)b * p.v_bs + h * HD;
    const uint8_t* kpm = p.kpm ? p.kpm + (int64_t)b * p.kpm_bs : nullptr;

    f32x16 O[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[d][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    if (HD < VD) {      // d rows beyond the head dim contribute zeros
        for (int e = t; e < VD * VROW / 4; e += 256) reinterpret_cast<uint32_t*>(s_vt)[e] = 0u;
        __syncthreads();
    }

    // software pipeline: tile t+1 is fetched into registers while tile t is consumed.  K: one float4 (4 d of one key)
    // per item; V: the same 4 d of TWO adjacent keys per item (their halfs pair up into one 32-bit transposed store)
    constexpr int C4 = HD / 4;
    constexpr int NLK = (KT * C4 + 255) / 256;
    constexpr int NLV = (KT / 2 * C4 + 255) / 256;
    f32x4 pk[NLK], pv[NLV][2];
    // V item -> (key pair, 4-wide d chunk): 4 chunks x 8 key pairs per 32 lanes.  The transposed 4-byte stores of a
    // 32-lane group then spread over 16 banks (2-way); chunk-fastest numbering put them on 4 (d rows are 272 B apart,
    // so a chunk's 4 rows only contribute their parity to the bank).
    auto vc = [&](int e) { return ((e >> 5) % (C4 / 4)) * 4 + (e & 3); };
    auto vkp = [&](int e) { return ((e >> 5) / (C4 / 4)) * 8 + ((e >> 2) & 7); };
    auto fetch = [&](int kt0) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NLK; ++i) {
            const int e = t + 256 * i;
            const int kr = e / C4, c = e - kr * C4;
            const int key = kt0 + kr;
            const bool ok = e < KT * C4 && key < k_end;
            const f32x4 kv = *reinterpret_cast<const f32x4*>(Kb + (ok ? (int64_t)key * p.k_rs + c * 4 : 0));
            pk[i] = ok ? kv : z;
        }
#pragma unroll
        for (int i = 0; i < NLV; ++i) {
            const int e = t + 256 * i;
            const int kp = vkp(e), c = vc(e);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int key = kt0 + 2 * kp + u;
                const bool ok = e < KT / 2 * C4 && key < k_end;
                const f32x4 vv = *reinterpret_cast<const f32x4*>(Vb + (ok ? (int64_t)key * p.v_rs + c * 4 : 0));
                pv[i][u] = ok ? vv : z;
            }
        }
    };
    if (!ATTN_NOPF) fetch(k_begin);
    for (int kt0 = k_begin; kt0 < k_end; kt0 += KT) {
        if (ATTN_NOPF) fetch(kt0);
        // ---- stage the prefetched tile (split to fp16 pieces on the way)
#pragma unroll
        for (int i = 0; i < NLK; ++i) {
            const int e = t + 256 * i;
            if (e < KT * C4) {
                const int kr = e / C4, c = e - kr * C4;
                uint2 hi, lo;
                split16(pk[i], hi, lo);
                *reinterpret_cast<uint2*>(s_k + kr * KROW + c * 8) = hi;
                *reinterpret_cast<uint2*>(s_k + kr * KROW + 2 * HD + c * 8) = lo;
            }
        }
#pragma unroll
        for (int i = 0; i < NLV; ++i) {
            const int e = t + 256 * i;
            if (e < KT / 2 * C4) {
                const int kp = vkp(e), c = vc(e);
                uint2 ha, la, hb, lb;
                split16(pv[i][0], ha, la);
                split16(pv[i][1], hb, lb);
                const int kl = 2 * kp, r = kl & 15;
                const int pos = (kl & ~15) | (r & 3) | ((r & 8) >> 1) | ((r & 4) << 1);      // permuted key slot (even)
                unsigned char* dst = s_vt + (c * 4) * VROW + pos * 2;
                // d = 4c+j: low half = key 2kp, high half = key 2kp+1
                *reinterpret_cast<uint32_t*>(dst + 0 * VROW) = (ha.x & 0xffffu) | (hb.x << 16);
                *reinterpret_cast<uint32_t*>(dst + 1 * VROW) = (ha.x >> 16) | (hb.x & 0xffff0000u);
                *reinterpret_cast<uint32_t*>(dst + 2 * VROW) = (ha.y & 0xffffu) | (hb.y << 16);
                *reinterpret_cast<uint32_t*>(dst + 3 * VROW) = (ha.y >> 16) | (hb.y & 0xffff0000u);
                *reinterpret_cast<uint32_t*>(dst + 0 * VROW + 2 * KT) = (la.x & 0xffffu) | (lb.x << 16);
                *reinterpret_cast<uint32_t*>(dst + 1 * VROW + 2 * KT) = (la.x >> 16) | (lb.x & 0xffff0000u);
                *reinterpret_cast<uint32_t*>(dst + 2 * VROW + 2 * KT) = (la.y & 0xffffu) | (lb.y << 16);
                *reinterpret_cast<uint32_t*>(dst + 3 * VROW + 2 * KT) = (la.y >> 16) | (lb.y & 0xffff0000u);
            }
        }
        const bool ragged = kt0 + KT > k_end || kpm != nullptr;    // block-uniform: masking only where needed
        if (ragged && t < KT) {
            const int key = kt0 + t;
            s_dead[t] = (key >= k_end) || (kpm && kpm[key] != 0);
        }
        __syncthreads();
        if (!ATTN_NOPF && kt0 + KT < k_end) fetch(kt0 + KT);
#pragma unroll
        for (int sub = 0; sub < KT / 32; ++sub) {
            const int kb = kt0 + sub * 32;
            if (kb < k_end) {
                f32x16 S;
#pragma unroll
                for (int e = 0; e < 16; ++e) S[e] = 0.f;
                const unsigned char* krow = s_k + (sub * 32 + li) * KROW + lh * 16;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const h16x8 kh = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(krow + s * 32));
                    const h16x8 kl = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(krow + 2 * HD + s * 32));
                    S = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s], S, 0, 0, 0);
                    S = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s], S, 0, 0, 0);
                    S = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], S, 0, 0, 0);
                }
                if (ragged) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (s_dead[sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh]) S[e] = -INFINITY;
                }
                if (p.causal) {          // key j is visible to query i only for j <= i (nn.MultiheadAttention attn_mask = triu(1))
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (kb + (e & 3) + 8 * (e >> 2) + 4 * lh > q) S[e] = -INFINITY;
                }
                float mt = S[0];
#pragma unroll
                for (int e = 1; e < 16; ++e) mt = fmaxf(mt, S[e]);
                mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
                const float m_new = fmaxf(m_run, mt);
                const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
                float rs = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    S[e] = __builtin_amdgcn_exp2f(S[e] - m_safe);
                    rs += S[e];
                }
                rs += __shfl_xor(rs, 32, 64);
                l_run = l_run * alpha + rs;          // the normaliser uses the UN-dropped weights
                m_run = m_new;
                if (p.drop_p > 0.f) {
                    const float ds = 1.f / (1.f - p.drop_p);
                    const uint64_t rowbase = (((uint64_t)b * p.H + h) * p.Nq + (qok ? q : 0)) * (uint64_t)p.Nk;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = kb + (e & 3) + 8 * (e >> 2) + 4 * lh;
                        S[e] = actmi_keep(p.drop_seed, rowbase + key, p.drop_p) ? S[e] * ds : 0.f;
                    }
                }
                // probabilities -> fp16 pieces, already in B-operand position: registers 0-7 are this lane half's 8 keys
                // of the first 16-key group, 8-15 of the second
                h16x8 ph[2], pl[2];
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const f32x4 a0 = {S[8 * g2], S[8 * g2 + 1], S[8 * g2 + 2], S[8 * g2 + 3]};
                    const f32x4 a1 = {S[8 * g2 + 4], S[8 * g2 + 5], S[8 * g2 + 6], S[8 * g2 + 7]};
                    uint2 h0, l0, h1, l1;
                    split16(a0, h0, l0);
                    split16(a1, h1, l1);
                    ph[g2] = __builtin_bit_cast(h16x8, uint4{h0.x, h0.y, h1.x, h1.y});
                    pl[g2] = __builtin_bit_cast(h16x8, uint4{l0.x, l0.y, l1.x, l1.y});
                }
#pragma unroll
                for (int d = 0; d < DT; ++d) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) O[d][e] *= alpha;
                    const unsigned char* vrow = s_vt + (d * 32 + li) * VROW + sub * 64 + lh * 16;
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        const h16x8 vh = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(vrow + g2 * 32));
                        const h16x8 vl = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(vrow + 2 * KT + g2 * 32));
                        O[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[g2], O[d], 0, 0, 0);
                        O[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[g2], O[d], 0, 0, 0);
                        O[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[g2], O[d], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (qok) {
        if (nsplit == 1) {
            const float inv = 1.f / l_run;
            float* Op = p.O + (int64_t)b * p.o_bs + (int64_t)q * p.o_rs + h * HD;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = d * 32 + 8 * g + 4 * lh;
                    if (d0 < HD) {
                        f32x4 o = {O[d][4 * g] * inv, O[d][4 * g + 1] * inv, O[d][4 * g + 2] * inv, O[d][4 * g + 3] * inv};
                        *reinterpret_cast<f32x4*>(Op + d0) = o;
                    }
                }
            if (p.lse && lh == 0) p.lse[((int64_t)b * p.H + h) * p.Nq + q] = m_run * 0.6931471805599453f + logf(l_run);
        } else {
            // partial result of this key range: un-normalised O plus (running max, running sum)
            const int64_t row = (((int64_t)split * p.B + b) * p.H + h) * p.Nq + q;
            float* Op = p.ws + row * HD;
            float* ml = p.ws + (int64_t)nsplit * p.B * p.H * p.Nq * HD + row * 2;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = d * 32 + 8 * g + 4 * lh;
                    if (d0 < HD) {
                        f32x4 o = {O[d][4 * g], O[d][4 * g + 1], O[d][4 * g + 2], O[d][4 * g + 3]};
                        *reinterpret_cast<f32x4*>(Op + d0) = o;
                    }
                }
            if (lh == 0) { ml[0] = m_run; ml[1] = l_run; }
        }
    }
}


// merge the nsplit partial results: m = max m_s, L = sum l_s 2^(m_s - m), O = sum O_s 2^(m_s - m) / L
__global__ void attn_combine_kernel(AttnArgs p, int nsplit, int HD) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // over (b, h, q, d4)
    const int D4 = HD / 4;
    const int64_t total = (int64_t)p.B * p.H * p.Nq * D4;
    if (idx >= total) return;
    const int d4 = (int)(idx % D4);
    const int64_t bhq = idx / D4;
    const int q = (int)(bhq % p.Nq);
    const int h = (int)((bhq / p.Nq) % p.H);
    const int b = (int)(bhq / ((int64_t)p.Nq * p.H));
    const int64_t per = (int64_t)p.B * p.H * p.Nq;
    const float* mlb = p.ws + (int64_t)nsplit * per * HD;
    float m = -INFINITY;
    for (int s = 0; s < nsplit; ++s) m = fmaxf(m, mlb[(s * per + bhq) * 2]);
    const float msafe = (m == -INFINITY) ? 0.f : m;
    float L = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nsplit; ++s) {
        const float w = __builtin_amdgcn_exp2f(mlb[(s * per + bhq) * 2] - msafe);
        L += mlb[(s * per + bhq) * 2 + 1] * w;
        acc += *reinterpret_cast<const f32x4*>(p.ws + (s * per + bhq) * HD + d4 * 4) * w;
    }
    const float inv = 1.f / L;
    *reinterpret_cast<f32x4*>(p.O + (int64_t)b * p.o_bs + (int64_t)q * p.o_rs + h * HD + d4 * 4) = acc * inv;
    if (p.lse && d4 == 0) p.lse[bhq] = msafe * 0.6931471805599453f + logf(L);
}

}  // namespace

int launch_attention(const AttnArgs& a, hipStream_t st, std::string* err) {
    auto fail = [&](const char* m) { if (err) *err = std::string("attention: ") + m; return -2; };
    if (a.B <= 0 || a.Nq <= 0) return 0;
    if (a.Nk <= 0) return fail("Nk must be positive");
    if (a.causal && a.Nq != a.Nk) return fail("causal needs Nq == Nk");
    if ((a.q_rs & 3) || (a.k_rs & 3) || (a.v_rs & 3) || (a.o_rs & 3) || (a.q_bs & 3) || (a.k_bs & 3) || (a.v_bs & 3) ||
        (a.o_bs & 3))
        return fail("strides must be multiples of 4 floats");
    if (((uintptr_t)a.Q & 15) || ((uintptr_t)a.K & 15) || ((uintptr_t)a.V & 15) || ((uintptr_t)a.O & 15))
        return fail("pointers must be 16-byte aligned");
    // precision: explicit in the descriptor, else ACTMI_GEMM_PREC (f32 | f16x3), else native fp32
    static const int env_prec = [] {
        const char* e = getenv("ACTMI_GEMM_PREC");
        if (!e) return ACTMI_PREC_F32;
        return (e[0] == 'f' && e[1] == '3') ? ACTMI_PREC_F32 : ACTMI_PREC_F16X3;
    }();
    const int prec = a.prec ? a.prec : env_prec;
    if (prec != ACTMI_PREC_F32 && prec != ACTMI_PREC_F16X3) return fail("bad prec");
    const bool f16 = prec == ACTMI_PREC_F16X3;
    const int qblocks = (a.Nq + 127) / 128;
    const int tiles = (a.Nk + KT - 1) / KT;
    int nsplit = 1;
    if (a.ws) {
        const long blocks = (long)qblocks * a.H * a.B;
        // grid target of the key split (tuning aid ACTMI_ATTN_SPLIT_TARGET; 768 residency slots at three workgroups per CU)
        static const long target = getenv("ACTMI_ATTN_SPLIT_TARGET") ? atol(getenv("ACTMI_ATTN_SPLIT_TARGET")) : 1024;
        nsplit = (int)((target + blocks - 1) / blocks);
        if (nsplit < 1) nsplit = 1;
        if (nsplit > 8) nsplit = 8;
        if (nsplit > tiles / 2) nsplit = tiles / 2 > 0 ? tiles / 2 : 1;
        while (nsplit > 1 && (int64_t)nsplit * a.B * a.Nq * ((int64_t)a.H * a.HD + 2 * a.H) > a.ws_floats) --nsplit;
    }
    const int chunk = ((tiles + nsplit - 1) / nsplit) * KT;
    dim3 grid(qblocks * nsplit, a.H, a.B);
    if (prof_enabled()) {
        char nm[48];
        snprintf(nm, sizeof(nm), "attn_%s_kernel<%d>", f16 ? "f16x3" : "f32", a.HD);
        prof_begin(nm, 4.0 * a.B * a.H * (double)a.Nq * a.Nk * a.HD,
                   4.0 * a.B * a.H * a.HD * (2.0 * a.Nq + 2.0 * a.Nk), st);
    }
    if (f16) {
        switch (a.HD) {
            case 64: hipLaunchKernelGGL(attn_f16x3_kernel<64>, grid, dim3(256), 0, st, a, nsplit, chunk); break;
            case 32: hipLaunchKernelGGL(attn_f16x3_kernel<32>, grid, dim3(256), 0, st, a, nsplit, chunk); break;
            case 16: hipLaunchKernelGGL(attn_f16x3_kernel<16>, grid, dim3(256), 0, st, a, nsplit, chunk); break;
            default: return fail("head_dim must be 16, 32 or 64");
        }
    } else
    switch (a.HD) {
        case 64: hipLaunchKernelGGL(attn_f32_kernel<64>, grid, dim3(256), 0, st, a, nsplit, chunk); break;
        case 32: hipLaunchKernelGGL(attn_f32_kernel<32>, grid, dim3(256), 0, st, a, nsplit, chunk); break;
        case 16: hipLaunchKernelGGL(attn_f32_kernel<16>, grid, dim3(256), 0, st, a, nsplit, chunk); break;
        default: return fail("head_dim must be 16, 32 or 64");
    }
    if (nsplit > 1) {
        const int64_t total = (int64_t)a.B * a.H * a.Nq * (a.HD / 4);
        hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, nsplit, a.HD);
    }
    prof_end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { if (err) *err = std::string("attention launch: ") + hipGetErrorString(e); return -3; }
    return 0;
}
