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
#include <cstdio>

namespace {

constexpr int KT = 64;   // keys per LDS tile

template <int HD>
__global__ __launch_bounds__(256) void attn_f32_kernel(AttnArgs p) {
    constexpr int HH = HD / 2;            // k-steps of the QK^T product per lane half
    constexpr int DT = (HD + 31) / 32;    // 32-wide output tiles along d
    constexpr int VD = DT * 32;
    constexpr int KS = HD + 4;
    __shared__ __attribute__((aligned(16))) float s_k[KT * KS];
    __shared__ __attribute__((aligned(16))) float s_v[KT * VD];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q = blockIdx.x * 128 + wave * 32 + li;
    const bool qok = q < p.Nq;

    const float* Qp = p.Q + (int64_t)b * p.q_bs + (int64_t)(qok ? q : 0) * p.q_rs + h * HD + lh * HH;
    float qreg[HH];
#pragma unroll
    for (int s = 0; s < HH; s += 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(Qp + s);
#pragma unroll
        for (int j = 0; j < 4; ++j) qreg[s + j] = qok ? v[j] * p.scale : 0.f;
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

    for (int kt0 = 0; kt0 < p.Nk; kt0 += KT) {
        // ---- stage K and V tiles (float4 along d), zero beyond Nk
        constexpr int C4 = HD / 4;
        for (int e = t; e < KT * C4; e += 256) {
            const int kr = e / C4, c = e - kr * C4;
            const int key = kt0 + kr;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (key < p.Nk) {
                kv = *reinterpret_cast<const f32x4*>(Kb + (int64_t)key * p.k_rs + c * 4);
                vv = *reinterpret_cast<const f32x4*>(Vb + (int64_t)key * p.v_rs + c * 4);
            }
            *reinterpret_cast<f32x4*>(&s_k[kr * KS + c * 4]) = kv;
            *reinterpret_cast<f32x4*>(&s_v[kr * VD + c * 4]) = vv;
        }
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < KT / 32; ++sub) {
            const int kb = kt0 + sub * 32;
            if (kb < p.Nk) {
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
                float mt = -INFINITY;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kb + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    bool dead = key >= p.Nk;
                    if (kpm && !dead) dead = kpm[key] != 0;
                    if (dead) S[e] = -INFINITY;
                    mt = fmaxf(mt, S[e]);
                }
                mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
                const float m_new = fmaxf(m_run, mt);
                const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = expf(m_run - m_safe);
                float rs = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    S[e] = expf(S[e] - m_safe);
                    rs += S[e];
                }
                rs += __shfl_xor(rs, 32, 64);
                l_run = l_run * alpha + rs;
                m_run = m_new;
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
        if (p.lse && lh == 0) p.lse[((int64_t)b * p.H + h) * p.Nq + q] = m_run + logf(l_run);
    }
}

}  // namespace

int launch_attention(const AttnArgs& a, hipStream_t st, std::string* err) {
    auto fail = [&](const char* m) { if (err) *err = std::string("attention: ") + m; return -2; };
    if (a.B <= 0 || a.Nq <= 0) return 0;
    if (a.Nk <= 0) return fail("Nk must be positive");
    if ((a.q_rs & 3) || (a.k_rs & 3) || (a.v_rs & 3) || (a.o_rs & 3) || (a.q_bs & 3) || (a.k_bs & 3) || (a.v_bs & 3) ||
        (a.o_bs & 3))
        return fail("strides must be multiples of 4 floats");
    if (((uintptr_t)a.Q & 15) || ((uintptr_t)a.K & 15) || ((uintptr_t)a.V & 15) || ((uintptr_t)a.O & 15))
        return fail("pointers must be 16-byte aligned");
    dim3 grid((a.Nq + 127) / 128, a.H, a.B);
    if (prof_enabled()) {
        char nm[48];
        snprintf(nm, sizeof(nm), "attn_f32_kernel<%d>", a.HD);
        prof_begin(nm, 4.0 * a.B * a.H * (double)a.Nq * a.Nk * a.HD,
                   4.0 * a.B * a.H * a.HD * (2.0 * a.Nq + 2.0 * a.Nk), st);
    }
    switch (a.HD) {
        case 64: hipLaunchKernelGGL(attn_f32_kernel<64>, grid, dim3(256), 0, st, a); break;
        case 32: hipLaunchKernelGGL(attn_f32_kernel<32>, grid, dim3(256), 0, st, a); break;
        case 16: hipLaunchKernelGGL(attn_f32_kernel<16>, grid, dim3(256), 0, st, a); break;
        default: return fail("head_dim must be 16, 32 or 64");
    }
    prof_end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { if (err) *err = std::string("attention launch: ") + hipGetErrorString(e); return -3; }
    return 0;
}
