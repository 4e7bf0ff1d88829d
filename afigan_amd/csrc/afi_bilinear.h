// Bilinear x2 index map (align_corners = False, generator_rdb.py:125) and one output element of its transpose: shared by the standalone
// kernels (elementwise.hip) and the small-map backward's prologue launch (smallmap.hip), so both produce the same bits.
#pragma once
#include "afi_common.h"

__device__ __forceinline__ void afi_bil_idx2(int o, int L, int& i0, int& i1, float& lam) {
    float s = fmaxf(0.5f * (float)o - 0.25f, 0.f);
    i0 = (int)s; lam = s - (float)i0; i1 = min(i0 + 1, L - 1);
}
// float4 i of bilinear2x^T(dout[N,2H,2W,C] dense) on the [N,H,W,C] grid
__device__ __forceinline__ f32x4 afi_bilinear2x_bwd_elem(const float* __restrict__ dout, int H, int W, int C, long long i) {
    const int C4 = C / 4;
    const int c = (int)(i % C4) * 4; long long r = i / C4;
    const int xi = (int)(r % W); r /= W; const int yi = (int)(r % H); const int n = (int)(r / H);
    f32x4 acc = {0, 0, 0, 0};
    for (int yo = max(2 * yi - 2, 0); yo <= min(2 * yi + 2, 2 * H - 1); ++yo) {
        int y0, y1; float ly; afi_bil_idx2(yo, H, y0, y1, ly);
        const float wy = (y0 == yi ? 1.f - ly : 0.f) + (y1 == yi ? ly : 0.f);
        if (wy == 0.f) continue;
        for (int xo = max(2 * xi - 2, 0); xo <= min(2 * xi + 2, 2 * W - 1); ++xo) {
            int x0, x1; float lx; afi_bil_idx2(xo, W, x0, x1, lx);
            const float wx = (x0 == xi ? 1.f - lx : 0.f) + (x1 == xi ? lx : 0.f);
            if (wx == 0.f) continue;
            acc += (wy * wx) * *(const f32x4*)(dout + (((long long)n * 2 * H + yo) * 2 * W + xo) * C + c);
        }
    }
    return acc;
}
