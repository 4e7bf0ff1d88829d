// Shared by the f16x3 GEMM kernels (afi_gemm_f16.h) and the Winograd transforms that write their planes already split (winograd.hip):
// the power-of-two operand scale and the two-piece fp16 split.
#pragma once
#include "afi_common.h"
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// (AfiF16Bound: afi_common.h)

// the power of two s with bound * s in [2^14, 2^15) (bound > 0, finite); bounds below 2^-100 (and zero) take s = 2^114
__device__ __forceinline__ float afi_f16_scale(float bound) {
    unsigned e = (__float_as_uint(bound) >> 23) & 0xffu;
    e = e < 27u ? 27u : e;
    return __uint_as_float((268u - e) << 23);
}
__device__ __forceinline__ float afi_pow2_inverse(float s) { return __uint_as_float((254u << 23) - __float_as_uint(s)); }   // s a normal power of two

// a pair of values and their (power-of-two) scale -> packed fp16 pieces of x * s (x0 in the low half).  The residual is formed by one
// fused multiply-add per element straight from the packed hi (v_fma_mix_f32 reads either half of it as an fp16 source): x * s is exact,
// so fma(x, s, -hi) is the same number as (x * s) - hi
__device__ __forceinline__ void afi_split2_f16_pair(float x0, float x1, float s, unsigned& hi, unsigned& lo) {
    // four instructions per pair (left to itself hipcc forms hi twice, packed and per element: seven): v_fma_mixlo/hi_f16 round
    // fma(f32, f32, f16-or-f32) to fp16 into the low / high half of the destination and keep the other half
    unsigned h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(h));
    hi = h; lo = l;
}
// the same split from full-rate conversions (v_cvt_pk_f16_f32, v_cvt_f32_f16, v_sub_f32): nine instructions per pair
__device__ __forceinline__ void afi_split2_f16_pair_cvt(float x0, float x1, float s, unsigned& hi, unsigned& lo) {
    const float a = x0 * s, b = x1 * s;
    const f16x2 h = __builtin_convertvector((f32x2){a, b}, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    const float r0 = a - (float)h[0], r1 = b - (float)h[1];
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){r0, r1}, f16x2));
}
__device__ __forceinline__ void afi_split2_f16_x8_cvt(f32x4 v0, f32x4 v1, float s, u32x4& h, u32x4& l) {
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    afi_split2_f16_pair_cvt(v0[0], v0[1], s, h0, l0);
    afi_split2_f16_pair_cvt(v0[2], v0[3], s, h1, l1);
    afi_split2_f16_pair_cvt(v1[0], v1[1], s, h2, l2);
    afi_split2_f16_pair_cvt(v1[2], v1[3], s, h3, l3);
    h = u32x4{h0, h1, h2, h3};
    l = u32x4{l0, l1, l2, l3};
}
// eight values (two float4) -> the hi and lo MFMA operands / 16-byte LDS rows
__device__ __forceinline__ void afi_split2_f16_x8(f32x4 v0, f32x4 v1, float s, u32x4& h, u32x4& l) {
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    afi_split2_f16_pair(v0[0], v0[1], s, h0, l0);
    afi_split2_f16_pair(v0[2], v0[3], s, h1, l1);
    afi_split2_f16_pair(v1[0], v1[1], s, h2, l2);
    afi_split2_f16_pair(v1[2], v1[3], s, h3, l3);
    h = u32x4{h0, h1, h2, h3};
    l = u32x4{l0, l1, l2, l3};
}
