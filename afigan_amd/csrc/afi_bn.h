// The BatchNorm affine of the discriminator blocks and its LeakyReLU, as ONE definition for every kernel that evaluates them: the stand-alone
// apply pass and the backward's recomputed masks (elementwise.hip), and the Winograd input transforms that apply them on load (winograd.hip).
#pragma once
#include "afi_common.h"

// z = ((x - mean) * invstd) * gamma + beta with every operation rounded on its own (no FMA contraction): the forward's activation and the
// backward's recomputed LeakyReLU' mask must take the SAME side of zero for every element, so both evaluate this one function on the same
// fp32 operands (and a host restatement in plain fp32 tensor ops reproduces it bit for bit: tests/d_parity_util.py).
__device__ __forceinline__ f32x4 afi_bn_affine(f32x4 v, f32x4 mu, f32x4 is, f32x4 ga, f32x4 be) {
#pragma clang fp contract(off)
    f32x4 t = v - mu;
    t = t * is;
    t = t * ga;
    return t + be;
}
// the activation of a block: LeakyReLU(slope) of the affine (feature_patch_discriminator.py:35-38)
__device__ __forceinline__ f32x4 afi_bn_lrelu(f32x4 v, f32x4 mu, f32x4 is, f32x4 ga, f32x4 be, float slope) {
    v = afi_bn_affine(v, mu, is, ga, be);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
    return v;
}
