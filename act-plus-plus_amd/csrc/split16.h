// Exact fp16 two-piece split of fp32 values (shared by the GEMM staging path and the weight pre-split kernel so that
// both produce identical bits).
#pragma once
#include <hip/hip_runtime.h>

typedef float actmi_f32x4 __attribute__((ext_vector_type(4)));

// exact two-piece fp16 split of 4 floats (8 VALU): hi = rn16(x) as packed halfs, lo = rn16(x - hi); the remainder is
// formed by v_fma_mix_f32 straight from the packed half (x - hi is exact in fp32)
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split16(const actmi_f32x4 v, uint2& hi, uint2& lo) {
    const f32x2 v01 = {v[0], v[1]}, v23 = {v[2], v[3]};
    const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector(v01, h16x2));
    const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_convertvector(v23, h16x2));
    float r0, r1, r2, r3;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h01), "v"(v[0]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h01), "v"(v[1]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(h23), "v"(v[2]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(h23), "v"(v[3]));
    const f32x2 r01 = {r0, r1}, r23 = {r2, r3};
    hi = uint2{h01, h23};
    lo = uint2{__builtin_bit_cast(unsigned, __builtin_convertvector(r01, h16x2)),
               __builtin_bit_cast(unsigned, __builtin_convertvector(r23, h16x2))};
}

