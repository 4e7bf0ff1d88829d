// Fused epilogue of the pixel GEMMs (and of the Winograd output transform): shared device code, internal to the library.
#pragma once
#include "afi_common.h"

__device__ __forceinline__ float afi_lrelu(float v) { return v > 0.f ? v : v * AFI_LRELU_SLOPE; }

// bilinear x2, align_corners=False (generator_rdb.py:125): s = max(0.5*o - 0.25, 0)
__device__ __forceinline__ void afi_bil_idx(int o, int L, int& i0, int& i1, float& lam) {
    float s = fmaxf(0.5f * (float)o - 0.25f, 0.f);
    i0 = (int)s;                 // s >= 0 so truncation == floor
    lam = s - (float)i0;
    i1 = min(i0 + 1, L - 1);
}


// Fused epilogue of one float4 of accumulators (4 consecutive output columns of one GEMM row):
//   v = alpha*acc + bias + beta*O_old + r1s*R1 (direct or bilinear x2) + r2s*R2 ; activation ; * lrelu'(Z) ; (pixel-shuffle) store
// POST (generic-tap kernels only): the post-activation form  out = post_scale * act(...) + r2s * R2, act(...) -> O2
template <bool POST = false>
__device__ __forceinline__ void afi_epilogue_store(const AfiPixGemm& p, int img, int y, int x, int col, f32x4 accv) {
    int phase = 0, ch = col;
    if (p.o_up == 2) { phase = col / p.CoutPhase; ch = col - phase * p.CoutPhase; }
    const int yo = y * p.o_up + (phase >> 1), xo = x * p.o_up + (phase & 1);
    if (yo >= p.oH || xo >= p.oW) return;
    float* dst = p.O.p + (long long)img * p.O.sN + (long long)yo * p.O.sH + (long long)xo * p.O.sW + ch;
    f32x4 v = p.alpha * accv;
    if (p.bias) v += *(const f32x4*)(p.bias + ch);
    if (p.beta != 0.f) v += p.beta * *(const f32x4*)dst;
    if (p.R1.p && ch >= p.r1_lo && ch < p.r1_hi) {
        if (p.r1_bilinear) {
            int by0, by1, bx0, bx1; float ly, lx;
            afi_bil_idx(y, p.H >> 1, by0, by1, ly); afi_bil_idx(x, p.W >> 1, bx0, bx1, lx);
            const float* rb = p.R1.p + (long long)img * p.R1.sN + ch;
            const f32x4 x00 = *(const f32x4*)(rb + (long long)by0 * p.R1.sH + (long long)bx0 * p.R1.sW);
            const f32x4 x01 = *(const f32x4*)(rb + (long long)by0 * p.R1.sH + (long long)bx1 * p.R1.sW);
            const f32x4 x10 = *(const f32x4*)(rb + (long long)by1 * p.R1.sH + (long long)bx0 * p.R1.sW);
            const f32x4 x11 = *(const f32x4*)(rb + (long long)by1 * p.R1.sH + (long long)bx1 * p.R1.sW);
            const f32x4 top = x00 * (1.f - lx) + x01 * lx;
            const f32x4 bot = x10 * (1.f - lx) + x11 * lx;
            v += p.r1s * (top * (1.f - ly) + bot * ly);
        } else {
            v += p.r1s * *(const f32x4*)(p.R1.p + (long long)img * p.R1.sN + (long long)yo * p.R1.sH + (long long)xo * p.R1.sW + ch);
        }
    }
    if ((!POST || !p.r2_post) && p.R2.p && ch >= p.r2_lo && ch < p.r2_hi)
        v += p.r2s * *(const f32x4*)(p.R2.p + (long long)img * p.R2.sN + (long long)yo * p.R2.sH + (long long)xo * p.R2.sW + ch);
    if (p.lrelu) {
        const float slope = (p.lrelu == 1) ? AFI_LRELU_SLOPE : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
    }
    if (p.Z.p && ch >= p.z_lo && ch < p.z_hi) {
        const f32x4 z = *(const f32x4*)(p.Z.p + (long long)img * p.Z.sN + (long long)yo * p.Z.sH + (long long)xo * p.Z.sW + ch);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= (z[j] > 0.f ? 1.f : AFI_LRELU_SLOPE);
    }
    if (POST && p.r2_post) {                              // out = post_scale * act(...) + r2s * R2; the activated value goes to O2
        if (p.O2.p) *(f32x4*)(p.O2.p + (long long)img * p.O2.sN + (long long)yo * p.O2.sH + (long long)xo * p.O2.sW + ch) = v;
        v *= p.post_scale;
        if (p.R2.p) v += p.r2s * *(const f32x4*)(p.R2.p + (long long)img * p.R2.sN + (long long)yo * p.R2.sH + (long long)xo * p.R2.sW + ch);
    }
    *(f32x4*)dst = v;
}

// The common case of the big Winograd convs (D forward: bias only; D / G data gradients: the LeakyReLU' mask over every channel;
// plain conv + activation): no beta, residuals or pixel shuffle.  A lean body keeps the 16 inlined copies in an output transform small
// (the general one above makes that kernel ~5500 instructions long).
__device__ __forceinline__ bool afi_epilogue_is_simple(const AfiPixGemm& p) {
    return p.o_up == 1 && p.beta == 0.f && !p.R1.p && !p.R2.p && !p.r2_post && p.oH >= p.H && p.oW >= p.W &&
           (!p.Z.p || (p.z_lo == 0 && p.z_hi >= p.Ncols));
}
__device__ __forceinline__ f32x4 afi_epilogue_store_simple(const AfiPixGemm& p, int img, int y, int x, int col, f32x4 accv) {
    const long long pix = (long long)img * p.O.sN + (long long)y * p.O.sH + (long long)x * p.O.sW + col;
    f32x4 v = p.alpha * accv;
    if (p.bias) v += *(const f32x4*)(p.bias + col);
    if (p.lrelu) {
        const float slope = (p.lrelu == 1) ? AFI_LRELU_SLOPE : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
    }
    if (p.Z.p) {
        const f32x4 z = *(const f32x4*)(p.Z.p + (long long)img * p.Z.sN + (long long)y * p.Z.sH + (long long)x * p.Z.sW + col);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= (z[j] > 0.f ? 1.f : AFI_LRELU_SLOPE);
    }
    *(f32x4*)(p.O.p + pix) = v;
    return v;                                               // (the stored value: what fused statistics accumulate)
}

