// bf16 split helpers shared by the batched Winograd GEMMs (afi_gemm_bf16.h) and the small-map kernels (smallmap.hip):
//   x = hi + mid + lo exactly, hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (round-to-nearest-even; both residuals are exact in
//   fp32), and the transposed LDS fragment read of the k-slow operands.
#pragma once
#include "afi_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x2 afi_pack_bf16(f32x4 v) {
    const bf16x4 h = __builtin_convertvector(v, bf16x4);
    return __builtin_bit_cast(u32x2, h);
}
__device__ __forceinline__ bf16x8 afi_pack8_bf16(f32x4 lo, f32x4 hi) {
    const bf16x4 a = __builtin_convertvector(lo, bf16x4), b = __builtin_convertvector(hi, bf16x4);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ f32x4 afi_bf16_residual(f32x4 v) {          // v - float(bf16(v)), exact in fp32
    const bf16x4 h = __builtin_convertvector(v, bf16x4);
    return v - __builtin_convertvector(h, f32x4);
}

// The same split on a PAIR of values with one v_cvt_pk_bf16_f32 per part (the vector forms above convert element by element when the
// packed result is unpacked again: 8.5 instead of 5.5 vector instructions per element): hi / mid / lo are packed bf16 pairs, x0 in the low half.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned afi_cvt_pk_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
__device__ __forceinline__ void afi_split3_pair(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = afi_cvt_pk_bf16(x0, x1);
    float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);          // exact in fp32
    mid = afi_cvt_pk_bf16(r0, r1);
    r0 -= __uint_as_float(mid << 16); r1 -= __uint_as_float(mid & 0xffff0000u);
    lo = afi_cvt_pk_bf16(r0, r1);
}

// ... with the four subtractions kept as v_sub_f32: beside MFMAs a v_pk_add_f32 (which -O3 forms out of the two adjacent subtractions) costs
// about three plain ones (MI355X_MICROARCH.md, 'price of one filler beside MFMAs'); for loops whose vector issue runs in the shadow of MFMAs
__device__ __forceinline__ void afi_split3_pair_np(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = afi_cvt_pk_bf16(x0, x1);
    float r0 = x0 - __uint_as_float(hi << 16);
    asm volatile("" : "+v"(r0));
    float r1 = x1 - __uint_as_float(hi & 0xffff0000u);
    asm volatile("" : "+v"(r1));
    mid = afi_cvt_pk_bf16(r0, r1);
    r0 -= __uint_as_float(mid << 16);
    asm volatile("" : "+v"(r0));
    r1 -= __uint_as_float(mid & 0xffff0000u);
    lo = afi_cvt_pk_bf16(r0, r1);
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 afi_tr_frag(const unsigned char* base, int off_lo, int off_hi) {
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + off_lo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + off_hi));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

