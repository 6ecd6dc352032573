// Counter-based dropout mask shared by forward and backward kernels: keep(seed, element index) is a pure function, so
// the backward pass regenerates exactly the mask the forward pass used (no mask storage).  nn.Dropout semantics
// (transformer.py:197-203, 258-266 and the attention-weight dropout inside nn.MultiheadAttention): kept elements are
// scaled by 1/(1-p).  The stream differs from torch's Philox stream, so runs are statistically, not bit-wise,
// comparable with the reference when p > 0.
#pragma once
#include <stdint.h>

__host__ __device__ __forceinline__ uint32_t actmi_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// uniform in [0,1) with 24 bits from (seed, 64-bit element index)
__host__ __device__ __forceinline__ float actmi_u01(uint64_t seed, uint64_t idx) {
    const uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
    uint32_t h = actmi_mix32(lo ^ (uint32_t)seed);
    h = actmi_mix32(h ^ hi ^ (uint32_t)(seed >> 32) ^ 0x9E3779B9U);
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

__host__ __device__ __forceinline__ bool actmi_keep(uint64_t seed, uint64_t idx, float p) { return actmi_u01(seed, idx) >= p; }

__host__ __device__ __forceinline__ uint64_t actmi_site_seed(uint64_t seed, uint32_t site) {
    const uint64_t z = seed + 0x9E3779B97F4A7C15ULL * (uint64_t)(site + 1);
    return ((uint64_t)actmi_mix32((uint32_t)z ^ 0xA511E9B3U) << 32) | actmi_mix32((uint32_t)(z >> 32) ^ (uint32_t)z);
}
