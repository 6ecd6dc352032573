// Backward-pass kernels of the ACT training step that are not GEMM-shaped: LayerNorm, max-pool, softmax,
// column sums (bias gradients), L1+KL loss, reparametrisation, fused multi-tensor AdamW.
// Reference semantics: torch autograd of the modules in transformer.py / detr_vae.py / policy.py:288-320,
// torch.optim.AdamW as configured at detr/main.py:102-110.
#include "common.h"
#include "dropout.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------- LayerNorm bwd
constexpr int MAXV = 8;

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w;  dw += dy * xhat;  db += dy.
// One wave per row, rows grid-strided; each wave keeps its dw/db partials in registers and adds them once.
template <int NV>     // float4 per lane actually needed (ceil(D/256)): no dead iterations, fewer live registers
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ dy, const float* __restrict__ dx_add,
                                                     float* __restrict__ dx, float* __restrict__ dw,
                                                     float* __restrict__ db, int M, int D, float eps,
                                                     float* __restrict__ part, unsigned* __restrict__ dx_amax) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nw = gridDim.x * 4;
    const int D4 = D >> 2;
    const f32x4* w4 = reinterpret_cast<const f32x4*>(w);
    f32x4 aw[NV], ab[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { aw[i] = f32x4{0.f, 0.f, 0.f, 0.f}; ab[i] = aw[i]; }
    const float invD = 1.f / (float)D;
    unsigned amx = 0;                  // bits of max |dx| (dx_amax: operand scale of the products that read dx next)
    for (int row = wid; row < M; row += nw) {
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + (int64_t)row * D);
        const f32x4* gr = reinterpret_cast<const f32x4*>(dy + (int64_t)row * D);
        f32x4 v[NV], g[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < D4) { v[i] = xr[c]; g[i] = gr[c]; s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
        }
        const float mean = wave_sum(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < D4) { const f32x4 d = v[i] - mean; q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]); }
        const float rstd = 1.f / sqrtf(wave_sum(q) * invD + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < D4) {
                const f32x4 xh = (v[i] - mean) * rstd;
                const f32x4 gw = g[i] * w4[c];
                aw[i] += g[i] * xh;
                ab[i] += g[i];
                v[i] = xh; g[i] = gw;
                s1 += (gw[0] + gw[1]) + (gw[2] + gw[3]);
                s2 += (gw[0] * xh[0] + gw[1] * xh[1]) + (gw[2] * xh[2] + gw[3] * xh[3]);
            }
        }
        const float m1 = wave_sum(s1) * invD, m2 = wave_sum(s2) * invD;
        f32x4* dr = reinterpret_cast<f32x4*>(dx + (int64_t)row * D);
        const f32x4* ar = dx_add ? reinterpret_cast<const f32x4*>(dx_add + (int64_t)row * D) : nullptr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < D4) {
                f32x4 o = (g[i] - m1 - v[i] * m2) * rstd;
                if (ar) o += ar[c];
                dr[c] = o;
#pragma unroll
                for (int e = 0; e < 4; ++e) amx = max(amx, __float_as_uint(o[e]) & 0x7fffffffu);
            }
        }
    }
    if (dx_amax) {
        for (int o = 32; o > 0; o >>= 1) amx = max(amx, (unsigned)__shfl_xor((int)amx, o, 64));
        if (lane == 0 && amx) amax_commit(dx_amax, amx);
    }
    // block-level reduction of the 4 waves' partials through LDS, then ONE atomic per column per block
    // (4096 waves adding to the same 2*D addresses was 15x slower than the row pass itself)
    __shared__ float s_part[2][3][64 * 4 * NV];
    const int wv = threadIdx.x >> 6;
    if (wv > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < D4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { s_part[0][wv - 1][c * 4 + e] = aw[i][e]; s_part[1][wv - 1][c * 4 + e] = ab[i][e]; }
            }
        }
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < D4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = c * 4 + e;
                    const float sw = aw[i][e] + s_part[0][0][k] + s_part[0][1][k] + s_part[0][2][k];
                    const float sb = ab[i][e] + s_part[1][0][k] + s_part[1][1][k] + s_part[1][2][k];
                    if (part) {          // deterministic form: one row of partials per block, summed in block order afterwards
                        part[((int64_t)blockIdx.x * 2 + 0) * D + k] = sw;
                        part[((int64_t)blockIdx.x * 2 + 1) * D + k] = sb;
                    } else {
                        atomicAdd(&dw[k], sw);
                        atomicAdd(&db[k], sb);
                    }
                }
            }
        }
    }
}

// Fixed-order sum of ny partial rows: out[n] += sum_y part[y][n].  Block = 64 columns x 4 row lanes; lane l takes the rows
// y = l (mod 4) in four interleaved chains (16 loads in flight per column), the 16 chain sums are added in a fixed tree:
// bitwise repeatable (float atomics are not), and ~15x faster than one thread walking all ny rows of its column.
__device__ __forceinline__ float ordered_colsum(const float* __restrict__ part, int ny, int64_t ld, int col, int rl, float (*s_l)[64]) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int y = rl;
    for (; y + 12 < ny; y += 16) {
        a0 += part[(int64_t)y * ld + col];
        a1 += part[(int64_t)(y + 4) * ld + col];
        a2 += part[(int64_t)(y + 8) * ld + col];
        a3 += part[(int64_t)(y + 12) * ld + col];
    }
    for (; y < ny; y += 4) a0 += part[(int64_t)y * ld + col];
    s_l[rl][threadIdx.x & 63] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    return (s_l[0][threadIdx.x & 63] + s_l[1][threadIdx.x & 63]) + (s_l[2][threadIdx.x & 63] + s_l[3][threadIdx.x & 63]);
}

// out_a[k] += sum_b part[b][0][k], out_b[k] += sum_b part[b][1][k]   (LayerNorm backward: block partials of d-gamma / d-beta)
__global__ __launch_bounds__(256) void reduce_pairs_kernel(const float* __restrict__ part, int nblocks, int D, float* __restrict__ out_a,
                                                           float* __restrict__ out_b) {
    __shared__ float s_l[4][64];
    const int k = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const bool ok = k < 2 * D;
    const float t = ordered_colsum(part, ok ? nblocks : 0, 2 * (int64_t)D, ok ? k : 0, rl, s_l);
    if (ok && rl == 0) {
        const int which = k / D, col = k - which * D;
        float* o = which ? out_b : out_a;
        o[col] += t;
    }
}

// out[n] += sum_y part[y][n]
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ part, int ny, int N, float* __restrict__ out) {
    __shared__ float s_l[4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const bool ok = n < N;
    const float t = ordered_colsum(part, ok ? ny : 0, N, ok ? n : 0, rl, s_l);
    if (ok && rl == 0) out[n] += t;
}

// ---------------------------------------------------------------------------------------------- max-pool bwd
// dx[pixel] = sum over the (<= 4) windows containing it of dy[window] where the window's FIRST maximum (scan order
// r then s, strict '>', as ATen's max_pool2d) is this pixel.  Gather form: no atomics, bit-reproducible.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ dx, int H, int W, int C4, int Ho, int Wo,
                                                          int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        // 32-bit index arithmetic (the host checks total < 2^31): 64-bit divisions were most of this kernel's time
        const unsigned uidx = (unsigned)idx;
        const unsigned upix = uidx / (unsigned)C4;
        const int c4 = (int)(uidx - upix * (unsigned)C4);
        const unsigned urow = upix / (unsigned)W;
        const int wi = (int)(upix - urow * (unsigned)W);
        const unsigned uimg = urow / (unsigned)H;
        const int hi = (int)(urow - uimg * (unsigned)H);
        const int64_t img = uimg;
        const f32x4* xs = reinterpret_cast<const f32x4*>(x) + img * H * W * C4 + c4;
        const f32x4* gs = reinterpret_cast<const f32x4*>(dy) + img * Ho * Wo * C4 + c4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int ho_lo = hi >> 1, ho_hi = (hi + 1) >> 1;      // windows with 2ho-1 <= hi <= 2ho+1
        const int wo_lo = wi >> 1, wo_hi = (wi + 1) >> 1;
        for (int ho = ho_lo; ho <= ho_hi; ++ho) {
            if (ho >= Ho) continue;
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                if (wo >= Wo) continue;
                f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                int bpos[4] = {-1, -1, -1, -1};
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int h2 = 2 * ho - 1 + r;
                    if ((unsigned)h2 >= (unsigned)H) continue;
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        const int w2 = 2 * wo - 1 + s;
                        if ((unsigned)w2 >= (unsigned)W) continue;
                        const f32x4 v = xs[((int64_t)h2 * W + w2) * C4];
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (v[e] > best[e] || bpos[e] < 0) { best[e] = v[e]; bpos[e] = h2 * W + w2; }
                    }
                }
                const f32x4 g = gs[((int64_t)ho * Wo + wo) * C4];
                const int me = hi * W + wi;
#pragma unroll
                for (int e = 0; e < 4; ++e) if (bpos[e] == me) acc[e] += g[e];
            }
        }
        reinterpret_cast<f32x4*>(dx)[idx] = acc;
    }
}

// ---------------------------------------------------------------------------------------------- column sums
// out[n] += sum_m src[m][n]   (bias gradients; rows split over blockIdx.y, one atomic per column per block)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ src, int64_t ld, float* __restrict__ out,
                                                     int M, int N, int rows_per_block, float* __restrict__ part) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int m0 = blockIdx.y * rows_per_block;
    const int m1 = (m0 + rows_per_block < M) ? m0 + rows_per_block : M;
    float acc = 0.f;
    for (int m = m0; m < m1; ++m) acc += src[(int64_t)m * ld + n];
    if (part) part[(int64_t)blockIdx.y * N + n] = acc;
    else atomicAdd(&out[n], acc);
}

// the same with four columns per thread (N, ld multiples of 4): block = 64 column groups x 4 row lanes over rows_per_block rows,
// four loads in flight per lane, lanes combined in a fixed order; always writes a partial per row block
__global__ __launch_bounds__(256) void colsum_vec_kernel(const f32x4* __restrict__ src, int64_t ld4, int M, int N4, int rows_per_block,
                                                         f32x4* __restrict__ part) {
    __shared__ f32x4 s_l[4][64];
    const int cg = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const bool ok = cg < N4;
    const int m0 = blockIdx.y * rows_per_block;
    const int m1 = (m0 + rows_per_block < M) ? m0 + rows_per_block : M;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 a0 = z, a1 = z, a2 = z, a3 = z;
    if (ok) {
        const f32x4* p = src + cg;
        int m = m0 + rl;
        for (; m + 12 < m1; m += 16) {
            a0 += p[(int64_t)m * ld4];
            a1 += p[(int64_t)(m + 4) * ld4];
            a2 += p[(int64_t)(m + 8) * ld4];
            a3 += p[(int64_t)(m + 12) * ld4];
        }
        for (; m < m1; m += 4) a0 += p[(int64_t)m * ld4];
    }
    s_l[rl][threadIdx.x & 63] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ok && rl == 0)
        part[(int64_t)blockIdx.y * N4 + cg] = (s_l[0][threadIdx.x & 63] + s_l[1][threadIdx.x & 63]) +
                                              (s_l[2][threadIdx.x & 63] + s_l[3][threadIdx.x & 63]);
}

// dst[r][d] (+)= sum_b src[b*bs + r*ld + d]
__global__ void sum_batch_kernel(const float* __restrict__ src, int64_t bs, int64_t ld, float* __restrict__ dst, int B,
                                 int R, int D, int accumulate) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)R * D) return;
    const int r = (int)(idx / D), d = (int)(idx - (int64_t)r * D);
    float acc = accumulate ? dst[idx] : 0.f;
    for (int b = 0; b < B; ++b) acc += src[(int64_t)b * bs + (int64_t)r * ld + d];
    dst[idx] = acc;
}

// ---------------------------------------------------------------------------------------------- attention bwd pieces
// delta[b][h][q] = sum_d dO[b][q][h*HD+d] * O[b][q][h*HD+d]
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ dO, const float* __restrict__ O,
                                                         float* __restrict__ delta, int B, int H, int Nq, int HD) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);       // (b, q, h) flattened with h fastest
    if (row >= (int64_t)B * Nq * H) return;
    const int h = (int)(row % H);
    const int64_t bq = row / H;
    const float* a = dO + bq * (int64_t)H * HD + h * HD;
    const float* o = O + bq * (int64_t)H * HD + h * HD;
    float s = 0.f;
    for (int d = lane; d < HD; d += 64) s += a[d] * o[d];
    s = wave_sum(s);
    if (lane == 0) {
        const int64_t b = bq / Nq, q = bq - b * Nq;
        delta[(b * H + h) * Nq + q] = s;
    }
}

// P[g][q][k] = exp(S - lse[g][q]) (0 for masked / padded keys), in place; S was produced as scale * q.k
__global__ void attn_probs_kernel(float* __restrict__ S, const float* __restrict__ lse, const uint8_t* __restrict__ kpm,
                                  int64_t kpm_bs, int H, int Nq, int Nk, int ldp, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int k = (int)(idx % ldp);
    const int64_t gq = idx / ldp;
    const int64_t g = gq / Nq;
    float v = 0.f;
    if (k < Nk) {
        const bool dead = kpm && kpm[(g / H) * kpm_bs + k] != 0;
        if (!dead) v = expf(S[idx] - lse[gq]);
    }
    S[idx] = v;
}

// the same over 16-byte pieces of the rows (ldp % 4 == 0), 32-bit index arithmetic, bounded grid
__global__ __launch_bounds__(256) void attn_probs_vec_kernel(f32x4* __restrict__ S, const float* __restrict__ lse,
                                                             const uint8_t* __restrict__ kpm, int64_t kpm_bs, unsigned H, unsigned Nq,
                                                             int Nk, unsigned ldp4, unsigned total4) {
    const unsigned step = gridDim.x * 256u;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total4; i += step) {
        const unsigned gq = i / ldp4;
        const int k0 = (int)(i - gq * ldp4) * 4;
        const float l = lse[gq];
        const f32x4 s = S[i];
        const uint8_t* km = kpm ? kpm + (int64_t)(gq / Nq / H) * kpm_bs + k0 : nullptr;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool live = k0 + e < Nk && !(km && km[e] != 0);
            v[e] = live ? expf(s[e] - l) : 0.f;
        }
        S[i] = v;
        if (i + step < i) break;
    }
}

// x[r][c0 .. ld) = 0 for every row (the pad columns of the probability buffers)
__global__ void zero_cols_kernel(float* __restrict__ x, int64_t rows, int ld, int c0) {
    const int w = ld - c0;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * w) return;
    const int64_t r = i / w;
    x[r * ld + c0 + (int)(i - r * w)] = 0.f;
}

// dS = P * (dP - delta[g][q]) * scale, in place of dP
__global__ void attn_ds_kernel(const float* __restrict__ P, float* __restrict__ dP, const float* __restrict__ delta,
                               float scale, int Nk, int ldp, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int k = (int)(idx % ldp);
    const int64_t gq = idx / ldp;
    dP[idx] = (k < Nk) ? P[idx] * (dP[idx] - delta[gq]) * scale : 0.f;
}

// the same over 16-byte pieces; amax_bits (optional) collects the bits of max |dS| (operand scale of the dQ / dK products)
__global__ __launch_bounds__(256) void attn_ds_vec_kernel(const f32x4* __restrict__ P, f32x4* __restrict__ dP,
                                                          const float* __restrict__ delta, float scale, int Nk, unsigned ldp4,
                                                          unsigned total4, unsigned* __restrict__ amax_bits) {
    const unsigned step = gridDim.x * 256u;
    unsigned am = 0;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total4; i += step) {
        const unsigned gq = i / ldp4;
        const int k0 = (int)(i - gq * ldp4) * 4;
        const float dl = delta[gq];
        const f32x4 pr = P[i], d = dP[i];
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = (k0 + e < Nk) ? pr[e] * (d[e] - dl) * scale : 0.f;
            am = max(am, __float_as_uint(v[e]) & 0x7fffffffu);
        }
        dP[i] = v;
        if (i + step < i) break;
    }
    if (amax_bits) {
        for (int o = 32; o > 0; o >>= 1) am = max(am, (unsigned)__shfl_xor((int)am, o, 64));
        if ((threadIdx.x & 63) == 0 && am) amax_commit(amax_bits, am);
    }
}

// ---------------------------------------------------------------------------------------------- losses
// l1 = mean_{b,t,a} |actions - a_hat| * (1 - is_pad)   (policy.py:314-315: mean over ALL elements)
__global__ __launch_bounds__(256) void l1_loss_kernel(const float* __restrict__ a_hat, const float* __restrict__ actions,
                                                      const uint8_t* __restrict__ is_pad, float* __restrict__ part,
                                                      int A, int64_t total) {
    float acc = 0.f;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x)
        if (!is_pad[idx / A]) acc += fabsf(actions[idx] - a_hat[idx]);
    acc = wave_sum(acc);
    __shared__ float s_w[4];
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    // one partial per block (summed in block order by loss_total_kernel: the loss is bitwise repeatable)
    if (threadIdx.x == 0) part[blockIdx.x] = ((s_w[0] + s_w[1]) + (s_w[2] + s_w[3])) / (float)total;
}

// kl = mean_b sum_d -0.5 (1 + logvar - mu^2 - exp(logvar))   (policy.py:386-387)
__global__ void kl_loss_kernel(const float* __restrict__ latent_info, float* __restrict__ losses, int B, int L) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < B * L; i += blockDim.x) {
        const int b = i / L, d = i - b * L;
        const float m = latent_info[b * 2 * L + d], lv = latent_info[b * 2 * L + L + d];
        acc += -0.5f * (1.f + lv - m * m - expf(lv));
    }
    acc = wave_sum(acc);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) losses[1] = (part[0] + part[1] + part[2] + part[3]) / (float)B;
}

__global__ void loss_total_kernel(float* losses, const float* __restrict__ part, int nparts, float kl_weight) {
    float l1 = 0.f;
    for (int i = 0; i < nparts; ++i) l1 += part[i];
    losses[0] = l1;
    losses[2] = l1 + losses[1] * kl_weight;
}

// d a_hat = sign(a_hat - actions) * (1 - is_pad) / (B*Q*A) * gscale
__global__ void l1_bwd_kernel(const float* __restrict__ a_hat, const float* __restrict__ actions,
                              const uint8_t* __restrict__ is_pad, float* __restrict__ d_a_hat, int A, int64_t total,
                              float gscale) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    float g = 0.f;
    if (!is_pad[idx / A]) {
        const float d = a_hat[idx] - actions[idx];
        g = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * gscale / (float)total;
    }
    d_a_hat[idx] = g;
}

// z = mu + exp(logvar/2) * eps   (detr_vae.py:19-22)
__global__ void reparam_kernel(const float* __restrict__ latent_info, const float* __restrict__ eps, float* __restrict__ z,
                               float* __restrict__ mu_out, float* __restrict__ logvar_out, int B, int L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * L) return;
    const int b = i / L, d = i - b * L;
    const float m = latent_info[b * 2 * L + d], lv = latent_info[b * 2 * L + L + d];
    z[i] = m + expf(lv * 0.5f) * eps[i];
    if (mu_out) mu_out[i] = m;
    if (logvar_out) logvar_out[i] = lv;
}

// d latent_info[b] = [ dz + klw*mu/B ,  dz*eps*0.5*exp(lv/2) + klw*(-0.5)(1 - exp(lv))/B ] * (gscale folded in dz / klw)
__global__ void reparam_kl_bwd_kernel(const float* __restrict__ latent_info, const float* __restrict__ eps,
                                      const float* __restrict__ dz, float* __restrict__ d_latent_info, int B, int L,
                                      float klw_scaled) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * L) return;
    const int b = i / L, d = i - b * L;
    const float m = latent_info[b * 2 * L + d], lv = latent_info[b * 2 * L + L + d];
    const float g = dz[i];
    d_latent_info[b * 2 * L + d] = g + klw_scaled * m / (float)B;
    d_latent_info[b * 2 * L + L + d] = g * eps[i] * 0.5f * expf(lv * 0.5f) + klw_scaled * (-0.5f) * (1.f - expf(lv)) / (float)B;
}

// ---------------------------------------------------------------------------------------------- AdamW
// torch.optim.AdamW (single-tensor formulation) over the whole parameter arena.  group[chunk] per 64-float slot:
// 0 = frozen/buffer/no-grad (skipped entirely, like params whose .grad is None), 1 = lr, 2 = lr_backbone.
// flags / skip_mask: the handle's device flag word; when any bit of skip_mask is up (a non-finite loss of this step's forward,
// a weight beyond its split scale) the whole update is skipped ON THE DEVICE -- no host sync, and NaN gradients never reach the
// fp32 master weights or the Adam moments (the host reads the word at its next natural synchronisation and raises)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    const uint8_t* __restrict__ group, int64_t n, float lr, float lr_bb,
                                                    float wd, float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                    const uint32_t* __restrict__ flags, uint32_t skip_mask) {
    if (flags && (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & skip_mask)) return;      // grid-uniform
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint8_t gr = group[i >> 6];
        if (!gr) continue;
        const float l = gr == 2 ? lr_bb : lr;
        float pi = p[i];
        const float gi = g[i];
        pi *= (1.f - l * wd);
        const float mi = m[i] * b1 + gi * (1.f - b1);
        const float vi = v[i] * b2 + gi * gi * (1.f - b2);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (l / bc1) * (mi / denom);
    }
}

// [G][O][(r,s,c)] forward-packed conv weight -> [G][C][(r,s,o)] for the data gradient (n fastest in the contraction)
// flip: the taps reversed (rs -> KK-1-rs): the data gradient of a stride-1 convolution as a forward convolution of dY
__global__ void repack_dgrad_w_kernel(const float* __restrict__ wf, float* __restrict__ wd, int O, int I, int KK,
                                      int64_t total, int flip) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int o = (int)(idx % O);
    int64_t rest = idx / O;
    const int rs = (int)(rest % KK); rest /= KK;
    const int c = (int)(rest % I);
    const int64_t g = rest / I;
    wd[idx] = wf[((g * O + o) * KK + (flip ? KK - 1 - rs : rs)) * I + c];
}

// wgrad comes out as [G][O][(r,s,c)]; the state_dict gradient is OIHW
__global__ void unpack_wgrad_kernel(const float* __restrict__ gp, float* __restrict__ g_oihw, int O, int I, int KH, int KW,
                                    int kpad, int ipack, int64_t total, int64_t g_gp, int64_t g_out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    gp += blockIdx.y * g_gp;                  // blockIdx.y = group (camera): its packed gradient and its OIHW gradient
    g_oihw += blockIdx.y * g_out;
    const int s = (int)(idx % KW);
    int64_t rest = idx / KW;
    const int r = (int)(rest % KH); rest /= KH;
    const int c = (int)(rest % I);
    const int64_t o = rest / I;
    g_oihw[idx] += gp[o * kpad + (r * KW + s) * ipack + c];      // accumulate like autograd
}

// y = x * (mask > 0) * scale[c]; also y_plain = x * (mask > 0)   (ReLU + FrozenBN backward on NHWC maps).
// Four consecutive channels per thread (C % 4 == 0); amax_bits (optional) collects the bits of max |y_scaled| for the
// power-of-two operand scale of the GEMMs that read y_scaled next (train.hip: dyn_scale) -- integer atomicMax, so the
// result does not depend on the order of arrival.
__global__ __launch_bounds__(256) void relu_bn_bwd_kernel(const f32x4* __restrict__ x, const f32x4* __restrict__ add,
                                                          const f32x4* __restrict__ mask, const float* __restrict__ scale,
                                                          f32x4* __restrict__ y_plain, f32x4* __restrict__ y_scaled, unsigned C4,
                                                          unsigned per_group4, unsigned* __restrict__ amax_bits) {
    // blockIdx.y = group (camera: its own FrozenBN scale); a bounded grid walks the group's map, so the amax costs one
    // atomic per wave of a few thousand waves (one per wave of a one-element-per-thread grid: millions on one address)
    const int64_t gbase = (int64_t)blockIdx.y * per_group4;
    const float* sc_g = scale ? scale + (int64_t)blockIdx.y * C4 * 4 : nullptr;
    const unsigned step = gridDim.x * 256u;
    unsigned am = 0;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < per_group4; i += step) {
        const int64_t idx = gbase + i;
        f32x4 v = x[idx];
        if (add) v += add[idx];
        if (mask) {
            const f32x4 mk = mask[idx];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
        }
        if (y_plain) y_plain[idx] = v;
        if (y_scaled) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(sc_g + (i % C4) * 4);
            v = v * sc;
            y_scaled[idx] = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) am = max(am, __float_as_uint(v[e]) & 0x7fffffffu);
        }
        if (i + step < i) break;                               // 32-bit wrap
    }
    if (amax_bits) {
        for (int o = 32; o > 0; o >>= 1) am = max(am, (unsigned)__shfl_xor((int)am, o, 64));
        if ((threadIdx.x & 63) == 0 && am) amax_commit(amax_bits, am);
    }
}

// dz[i] = keep(seed, i) ? dy[i] / (1-p) : 0   (backward of an epilogue dropout; i = element index of the forward output)
__global__ void dropout_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dz, uint64_t seed, float p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dz[i] = actmi_keep(seed, (uint64_t)i, p) ? dy[i] * (1.f / (1.f - p)) : 0.f;
}

// Pd = P * mask / (1-p) written to a second buffer (the dropped weights feed dV = Pd^T dO)
__global__ void attn_drop_kernel(const float* __restrict__ P, float* __restrict__ Pd, uint64_t seed, float p, int Nk, int ldp,
                                 int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int k = (int)(idx % ldp);
    const int64_t gq = idx / ldp;
    Pd[idx] = (k < Nk && actmi_keep(seed, (uint64_t)gq * Nk + k, p)) ? P[idx] * (1.f / (1.f - p)) : 0.f;
}

// dS = P * (dPd * mask/(1-p) - delta) * scale, in place of dPd
__global__ void attn_ds_drop_kernel(const float* __restrict__ P, float* __restrict__ dP, const float* __restrict__ delta,
                                    float scale, uint64_t seed, float p, int Nk, int ldp, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int k = (int)(idx % ldp);
    const int64_t gq = idx / ldp;
    float v = 0.f;
    if (k < Nk) {
        const float g = actmi_keep(seed, (uint64_t)gq * Nk + k, p) ? dP[idx] * (1.f / (1.f - p)) : 0.f;
        v = P[idx] * (g - delta[gq]) * scale;
    }
    dP[idx] = v;
}

}  // namespace

int launch_dropout_bwd(const float* dy, float* dz, uint64_t seed, float p, int64_t n, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dy, dz, seed, p, n);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_attn_drop(const float* P, float* Pd, uint64_t seed, float p, int G, int Nq, int Nk, int ldp, hipStream_t st) {
    const int64_t total = (int64_t)G * Nq * ldp;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(attn_drop_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, P, Pd, seed, p, Nk, ldp, total);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_attn_ds_drop(const float* P, float* dP, const float* delta, float scale, uint64_t seed, float p, int G, int Nq,
                        int Nk, int ldp, hipStream_t st) {
    const int64_t total = (int64_t)G * Nq * ldp;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(attn_ds_drop_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, P, dP, delta, scale, seed, p,
                       Nk, ldp, total);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_ln_bwd(const float* x, const float* w, const float* dy, const float* dx_add, float* dx, float* dw, float* db,
                  int M, int D, float eps, hipStream_t st, float* ws, int64_t ws_floats, unsigned* dx_amax) {
    if ((D & 3) || D > 64 * 4 * MAXV) return -2;
    if (M <= 0) return 0;
    // persistent workgroups (each adds its dw/db partials once): 4 per CU -- with one wave per SIMD (256 workgroups) the
    // dependent load -> reduce -> store chain of a row ran at 0.5 TB/s
    int blocks = (M + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    // with a workspace the dw / db partials of the blocks are summed in a fixed order (bitwise repeatable gradients)
    float* part = (ws && (int64_t)blocks * 2 * D <= ws_floats) ? ws : nullptr;
    prof_begin("ln_bwd_kernel", 0.0, 4.0 * M * D * 3.0, st);
    const int nv = (D / 4 + 63) / 64;
#define ACTMI_LNB(NV) hipLaunchKernelGGL(ln_bwd_kernel<NV>, dim3(blocks), dim3(256), 0, st, x, w, dy, dx_add, dx, dw, db, M, D, eps, part, dx_amax)
    switch (nv) {
        case 1: ACTMI_LNB(1); break;
        case 2: ACTMI_LNB(2); break;
        case 3: ACTMI_LNB(3); break;
        case 4: ACTMI_LNB(4); break;
        default: ACTMI_LNB(MAXV); break;
    }
#undef ACTMI_LNB
    if (part) hipLaunchKernelGGL(reduce_pairs_kernel, dim3((2 * D + 63) / 64), dim3(256), 0, st, part, blocks, D, dw, db);
    prof_end(st);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// gather form on the recorded argmax codes (maxpool_idx_kernel): an input element (hi, wi) lies in at most 2 x 2 windows;
// it receives dy of those whose code names its position
// relu_x / bn_scale (optional): the ReLU + FrozenBN backward of the stem folded in -- dx = pooled gradient * (relu_x > 0) *
// bn_scale[group][c] (group = image / imgs_per_group), amax_bits = bits of max |dx| -- instead of a second pass over the
// largest map of the network
__global__ __launch_bounds__(256) void maxpool_bwd_idx_kernel(const uint8_t* __restrict__ arg, const float* __restrict__ dy,
                                                              float* __restrict__ dx, int H, int W, int C4, int Ho, int Wo,
                                                              unsigned total, const float* __restrict__ relu_x,
                                                              const float* __restrict__ bn_scale, int imgs_per_group,
                                                              unsigned* __restrict__ amax_bits) {
    unsigned amx = 0;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const unsigned pix0 = idx / (unsigned)C4;
        const int c4 = (int)(idx - pix0 * (unsigned)C4);
        const unsigned row = pix0 / (unsigned)W;
        const int wi = (int)(pix0 - row * (unsigned)W);
        const unsigned img = row / (unsigned)H;
        const int hi = (int)(row - img * (unsigned)H);
        const int64_t obase = (int64_t)img * Ho * Wo * C4 + c4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int ho_lo = hi >> 1, ho_hi = (hi + 1) >> 1;      // windows with 2ho-1 <= hi <= 2ho+1
        const int wo_lo = wi >> 1, wo_hi = (wi + 1) >> 1;
        for (int ho = ho_lo; ho <= ho_hi; ++ho) {
            if (ho >= Ho) continue;
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                if (wo >= Wo) continue;
                const int me = (hi - (2 * ho - 1)) * 3 + (wi - (2 * wo - 1));
                const int64_t o = obase + ((int64_t)ho * Wo + wo) * C4;
                const uchar4 a = reinterpret_cast<const uchar4*>(arg)[o];
                const f32x4 g = reinterpret_cast<const f32x4*>(dy)[o];
                if (a.x == me) acc[0] += g[0];
                if (a.y == me) acc[1] += g[1];
                if (a.z == me) acc[2] += g[2];
                if (a.w == me) acc[3] += g[3];
            }
        }
        if (relu_x) {
            const f32x4 xv = reinterpret_cast<const f32x4*>(relu_x)[idx];
            const f32x4 sc = *reinterpret_cast<const f32x4*>(bn_scale + ((int64_t)(img / (unsigned)imgs_per_group) * C4 + c4) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[e] = xv[e] > 0.f ? acc[e] * sc[e] : 0.f;
                amx = max(amx, __float_as_uint(acc[e]) & 0x7fffffffu);
            }
        }
        reinterpret_cast<f32x4*>(dx)[idx] = acc;
    }
    if (amax_bits) {
        for (int o = 32; o > 0; o >>= 1) amx = max(amx, (unsigned)__shfl_xor((int)amx, o, 64));
        if ((threadIdx.x & 63) == 0 && amx) amax_commit(amax_bits, amx);
    }
}

int launch_maxpool_bwd_idx(const uint8_t* arg, const float* dy, float* dx, int nimg, int H, int W, int C, int Ho, int Wo,
                           hipStream_t st, const float* relu_x, const float* bn_scale, int imgs_per_group, unsigned* amax_bits) {
    if (C & 3) return -2;
    const int64_t total = (int64_t)nimg * H * W * (C / 4);
    if (total >= ((int64_t)1 << 31)) return -2;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    prof_begin("maxpool_bwd_idx_kernel", 0.0, 4.0 * nimg * C * ((double)H * W + 1.25 * Ho * Wo), st);
    hipLaunchKernelGGL(maxpool_bwd_idx_kernel, dim3((unsigned)blocks), dim3(256), 0, st, arg, dy, dx, H, W, C / 4, Ho, Wo, (unsigned)total,
                       relu_x, bn_scale, imgs_per_group > 0 ? imgs_per_group : 1, amax_bits);
    prof_end(st);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_maxpool_bwd(const float* x, const float* dy, float* dx, int nimg, int H, int W, int C, int Ho, int Wo,
                       hipStream_t st) {
    if (C & 3) return -2;
    const int64_t total = (int64_t)nimg * H * W * (C / 4);
    if (total >= ((int64_t)1 << 31)) return -2;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    prof_begin("maxpool_bwd_kernel", 0.0, 4.0 * nimg * C * (2.0 * H * W + (double)Ho * Wo), st);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, dy, dx, H, W, C / 4, Ho, Wo, total);
    prof_end(st);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_colsum(const float* src, int64_t ld, float* out, int M, int N, hipStream_t st, float* ws, int64_t ws_floats) {
    if (M <= 0 || N <= 0) return 0;
    prof_begin("colsum_kernel", 0.0, 4.0 * M * N, st);
    const bool vec = (N & 3) == 0 && (ld & 3) == 0 && ((uintptr_t)src & 15) == 0 && ws && ((uintptr_t)ws & 15) == 0;
    int rpb = 256;
    if (vec) {
        // row blocks sized for ~1000 workgroups in all; one partial row per block, then the ordered sum
        const int gx = (N / 4 + 63) / 64;
        while (rpb < 4096 && (int64_t)gx * ((M + rpb - 1) / rpb) > 1024) rpb *= 2;
        const int gy = (M + rpb - 1) / rpb;
        if ((int64_t)gy * N <= ws_floats) {
            hipLaunchKernelGGL(colsum_vec_kernel, dim3(gx, gy), dim3(256), 0, st, reinterpret_cast<const f32x4*>(src), ld / 4, M, N / 4,
                               rpb, reinterpret_cast<f32x4*>(ws));
            hipLaunchKernelGGL(reduce_rows_kernel, dim3((N + 63) / 64), dim3(256), 0, st, ws, gy, N, out);
            prof_end(st);
            return hipGetLastError() == hipSuccess ? 0 : -3;
        }
        rpb = 256;
    }
    dim3 grid((N + 255) / 256, (M + rpb - 1) / rpb);
    float* part = (ws && grid.y > 1 && (int64_t)grid.y * N <= ws_floats) ? ws : nullptr;      // one row block: already ordered
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, st, src, ld, out, M, N, rpb, part);
    if (part) hipLaunchKernelGGL(reduce_rows_kernel, dim3((N + 63) / 64), dim3(256), 0, st, part, (int)grid.y, N, out);
    prof_end(st);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_sum_batch(const float* src, int64_t bs, int64_t ld, float* dst, int B, int R, int D, int accumulate,
                     hipStream_t st) {
    const int64_t total = (int64_t)R * D;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(sum_batch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, bs, ld, dst, B, R, D,
                       accumulate);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_attn_delta(const float* dO, const float* O, float* delta, int B, int H, int Nq, int HD, hipStream_t st) {
    const int64_t rows = (int64_t)B * Nq * H;
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, dO, O, delta, B, H, Nq, HD);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_attn_probs(float* S, const float* lse, const uint8_t* kpm, int64_t kpm_bs, int G, int H, int Nq, int Nk, int ldp,
                      hipStream_t st) {
    const int64_t total = (int64_t)G * Nq * ldp;
    if (total <= 0) return 0;
    prof_begin("attn_probs_kernel", 0.0, 8.0 * total, st);
    if ((ldp & 3) == 0 && ((uintptr_t)S & 15) == 0 && total / 4 < ((int64_t)1 << 32)) {
        const unsigned t4 = (unsigned)(total / 4);
        unsigned blocks = (t4 + 256u * 4u - 1) / (256u * 4u);
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(attn_probs_vec_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<f32x4*>(S), lse, kpm, kpm_bs,
                           (unsigned)H, (unsigned)Nq, Nk, (unsigned)(ldp / 4), t4);
    } else
        hipLaunchKernelGGL(attn_probs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, S, lse, kpm, kpm_bs, H, Nq,
                           Nk, ldp, total);
    prof_end(st);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_zero_cols(float* x, int64_t rows, int ld, int c0, hipStream_t st) {
    if (c0 >= ld || rows <= 0) return 0;
    const int64_t total = rows * (ld - c0);
    hipLaunchKernelGGL(zero_cols_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, rows, ld, c0);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_attn_ds(const float* P, float* dP, const float* delta, float scale, int G, int Nq, int Nk, int ldp,
                   hipStream_t st, unsigned* amax_bits) {
    const int64_t total = (int64_t)G * Nq * ldp;
    if (total <= 0) return 0;
    const bool vec = (ldp & 3) == 0 && ((uintptr_t)P & 15) == 0 && ((uintptr_t)dP & 15) == 0 && total / 4 < ((int64_t)1 << 32);
    if (amax_bits && !vec) return -2;                       // the caller registered a producer-side amax: only this form has it
    prof_begin("attn_ds_kernel", 0.0, 12.0 * total, st);
    if (vec) {
        const unsigned t4 = (unsigned)(total / 4);
        unsigned blocks = (t4 + 256u * 4u - 1) / (256u * 4u);
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(attn_ds_vec_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const f32x4*>(P),
                           reinterpret_cast<f32x4*>(dP), delta, scale, Nk, (unsigned)(ldp / 4), t4, amax_bits);
    } else
        hipLaunchKernelGGL(attn_ds_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, P, dP, delta, scale, Nk, ldp,
                           total);
    prof_end(st);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_losses(const float* a_hat, const float* actions, const uint8_t* is_pad, const float* latent_info, float* losses,
                  int B, int Q, int A, int L, float kl_weight, hipStream_t st) {
    if (hipMemsetAsync(losses, 0, 3 * sizeof(float), st) != hipSuccess) return -3;
    const int64_t total = (int64_t)B * Q * A;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(l1_loss_kernel, dim3(blocks), dim3(256), 0, st, a_hat, actions, is_pad, losses + 4, A, total);
    if (latent_info) hipLaunchKernelGGL(kl_loss_kernel, dim3(1), dim3(256), 0, st, latent_info, losses, B, L);
    hipLaunchKernelGGL(loss_total_kernel, dim3(1), dim3(1), 0, st, losses, losses + 4, blocks, kl_weight);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_l1_bwd(const float* a_hat, const float* actions, const uint8_t* is_pad, float* d_a_hat, int B, int Q, int A,
                  float gscale, hipStream_t st) {
    const int64_t total = (int64_t)B * Q * A;
    hipLaunchKernelGGL(l1_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a_hat, actions, is_pad, d_a_hat,
                       A, total, gscale);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_reparam(const float* latent_info, const float* eps, float* z, float* mu_out, float* logvar_out, int B, int L,
                   hipStream_t st) {
    hipLaunchKernelGGL(reparam_kernel, dim3((B * L + 255) / 256), dim3(256), 0, st, latent_info, eps, z, mu_out, logvar_out,
                       B, L);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---- VQ-ACT latent (detr_vae.py:137-145): per (sample, class) a softmax over vq_dim logits and a one-hot code.  The code is
// either given (parity tests: the reference's multinomial draw) or drawn here by inverse CDF from the counter-based
// generator; the straight-through estimator makes latent_input = latent_out_proj(code) in the forward pass and routes
// d(code) to the probabilities in the backward pass.
__global__ void vq_code_kernel(const float* __restrict__ logits, const float* __restrict__ code_in, uint64_t seed,
                               float* __restrict__ probs, float* __restrict__ code, int n, int VD, float inv_temp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // (sample, class)
    if (i >= n) return;
    const float* lg = logits + (int64_t)i * VD;
    float m = -INFINITY;
    for (int j = 0; j < VD; ++j) m = fmaxf(m, lg[j] * inv_temp);
    float s = 0.f;
    for (int j = 0; j < VD; ++j) s += expf(lg[j] * inv_temp - m);
    const float inv = 1.f / s;
    int pick = VD - 1;
    const float u = code_in ? 0.f : actmi_u01(seed, (uint64_t)i);
    float cum = 0.f;
    bool found = false;
    for (int j = 0; j < VD; ++j) {
        const float pj = expf(lg[j] * inv_temp - m) * inv;
        if (probs) probs[(int64_t)i * VD + j] = pj;
        cum += pj;
        if (!found && cum > u) { pick = j; found = true; }
    }
    for (int j = 0; j < VD; ++j)
        code[(int64_t)i * VD + j] = code_in ? code_in[(int64_t)i * VD + j] : (j == pick ? 1.f : 0.f);
}

// softmax backward per (sample, class): dlogit_j = p_j (g_j - sum_k p_k g_k), g = d(code) through the straight-through path
__global__ void vq_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ g, float* __restrict__ dlogits, int n,
                              int VD) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float dot = 0.f;
    for (int j = 0; j < VD; ++j) dot += probs[(int64_t)i * VD + j] * g[(int64_t)i * VD + j];
    for (int j = 0; j < VD; ++j) dlogits[(int64_t)i * VD + j] = probs[(int64_t)i * VD + j] * (g[(int64_t)i * VD + j] - dot);
}

int launch_vq_code(const float* logits, const float* code_in, uint64_t seed, float* probs, float* code, int B, int VC, int VD,
                   hipStream_t st, float temperature) {
    const int n = B * VC;
    hipLaunchKernelGGL(vq_code_kernel, dim3((n + 255) / 256), dim3(256), 0, st, logits, code_in, seed, probs, code, n, VD,
                       1.f / temperature);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_vq_bwd(const float* probs, const float* g, float* dlogits, int B, int VC, int VD, hipStream_t st) {
    const int n = B * VC;
    hipLaunchKernelGGL(vq_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, st, probs, g, dlogits, n, VD);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_reparam_kl_bwd(const float* latent_info, const float* eps, const float* dz, float* d_latent_info, int B, int L,
                          float klw_scaled, hipStream_t st) {
    hipLaunchKernelGGL(reparam_kl_bwd_kernel, dim3((B * L + 255) / 256), dim3(256), 0, st, latent_info, eps, dz, d_latent_info,
                       B, L, klw_scaled);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_adamw(float* p, const float* g, float* m, float* v, const uint8_t* group, int64_t n, float lr, float lr_bb,
                 float wd, float b1, float b2, float eps, int64_t step, hipStream_t st, const uint32_t* flags, uint32_t skip_mask) {
    const float bc1 = 1.f - powf(b1, (float)step);
    const float bc2 = 1.f - powf(b2, (float)step);
    prof_begin("adamw_kernel", 0.0, 28.0 * (double)n, st);
    hipLaunchKernelGGL(adamw_kernel, dim3(256 * 8), dim3(256), 0, st, p, g, m, v, group, n, lr, lr_bb, wd, b1, b2, eps, bc1,
                       sqrtf(bc2), flags, skip_mask);
    prof_end(st);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_repack_dgrad_w(const float* wf, float* wd, int G, int O, int I, int KK, hipStream_t st, int flip) {
    const int64_t total = (int64_t)G * I * KK * O;
    hipLaunchKernelGGL(repack_dgrad_w_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, wf, wd, O, I, KK, total, flip);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_unpack_wgrad(const float* gp, float* g_oihw, int O, int I, int KH, int KW, int kpad, int ipack, hipStream_t st, int G,
                        int64_t g_gp, int64_t g_out) {
    const int64_t total = (int64_t)O * I * KH * KW;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3((unsigned)((total + 255) / 256), G > 0 ? G : 1), dim3(256), 0, st, gp, g_oihw, O, I,
                       KH, KW, kpad, ipack, total, g_gp, g_out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_relu_bn_bwd(const float* x, const float* add, const float* mask, const float* scale, float* y_plain,
                       float* y_scaled, int G, int64_t per_group, int C, hipStream_t st, unsigned* amax_bits) {
    const int64_t total = (int64_t)G * per_group;
    // NHWC maps with C % 4 == 0 (base_width multiples of 4); one group's map below 2^32 16-byte pieces
    if ((C & 3) || (per_group & 3) || per_group / 4 >= ((int64_t)1 << 32) || G > 65535) return -2;
    if (total == 0) return 0;
    const int64_t pg4 = per_group / 4;
    int64_t bx = (pg4 + 256 * 8 - 1) / (256 * 8);              // ~8 pieces per thread
    const int64_t cap = 4096 / G > 0 ? 4096 / G : 1;
    if (bx > cap) bx = cap;
    if (bx < 1) bx = 1;
    prof_begin("relu_bn_bwd_kernel", 0.0, 16.0 * total, st);
    hipLaunchKernelGGL(relu_bn_bwd_kernel, dim3((unsigned)bx, (unsigned)G), dim3(256), 0, st,
                       reinterpret_cast<const f32x4*>(x), reinterpret_cast<const f32x4*>(add), reinterpret_cast<const f32x4*>(mask),
                       scale, reinterpret_cast<f32x4*>(y_plain), reinterpret_cast<f32x4*>(y_scaled), (unsigned)(C / 4), (unsigned)pg4,
                       amax_bits);
    prof_end(st);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
