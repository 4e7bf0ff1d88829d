// Conv-transpose weight pack / gradient unpack body (kernels: elementwise.hip; the pack also rides in the small-map forward's image launch,
// smallmap.hip) -- one definition, so every user produces the same bits.
#pragma once
#include "afi_common.h"

// W  [Cin][Cout][6][6]  (torch ConvTranspose2d layout, generator_rdb.py:101-105)
// Wp [(phase*Cout + co)][tap][ci],  phase = 2a+c, tap = 3(dy+1)+(dx+1),  ky = a+2-2dy, kx = c+2-2dx
// Both directions go through an LDS tile of 32 input channels x 8 output channels x 36 taps so that the reads AND the writes
// are contiguous (W is contiguous along the 36 taps of one (ci, co), Wp along ci): a direct gather ran at ~0.6 TB/s (33 us for
// the 256x256 layer, on the critical path of every generator forward), the tiled form is bandwidth-bound.
#define AFI_CT_CI 32
#define AFI_CT_CO 2                                          // 8 -> 2: 1024 blocks for the 256x256 layer (the pass is latency-bound: 21 / 28 us with 256 blocks)
#define AFI_CT_LD (AFI_CT_CO * 36 + 1)
template <bool UNPACK>
__device__ __forceinline__ void afi_convT_repack_body(const float* __restrict__ src, float* __restrict__ dst, int Cin, int Cout, int bx, int by,
                                                      float (*T)[AFI_CT_LD] /* LDS [AFI_CT_CI][AFI_CT_LD]: [ci][co_l*36 + ky*6 + kx] */) {
    const int ci0 = bx * AFI_CT_CI, co0 = by * AFI_CT_CO, tid = threadIdx.x;
    // pack: src = W [Cin][Cout][6][6], dst = Wp [(phase*Cout + co)][tap][ci];  unpack: src = dWp, dst = dW (accumulated)
    // packed side: thread -> (ci lane, row), row = (phase, co_l, tap)
    auto packed_pass = [&](auto&& f) {
        const int ci = tid & (AFI_CT_CI - 1);
#pragma unroll 3
        for (int row = tid / AFI_CT_CI; row < 4 * AFI_CT_CO * 9; row += 256 / AFI_CT_CI) {
            const int tap = row % 9, r = row / 9, co_l = r % AFI_CT_CO, phase = r / AFI_CT_CO;
            const int a = phase >> 1, c = phase & 1, dy = tap / 3 - 1, dx = tap % 3 - 1;
            const int ky = a + 2 - 2 * dy, kx = c + 2 - 2 * dx;
            if (ci0 + ci < Cin && co0 + co_l < Cout)
                f(T[ci][co_l * 36 + ky * 6 + kx], (((long long)phase * Cout + co0 + co_l) * 9 + tap) * Cin + ci0 + ci);
        }
    };
    // torch side: thread -> consecutive floats of the 8*36-float run of one ci
    auto torch_pass = [&](auto&& f) {
#pragma unroll 3
        for (int i = tid; i < AFI_CT_CI * AFI_CT_CO * 36; i += 256) {
            const int ci = i / (AFI_CT_CO * 36), j = i - ci * (AFI_CT_CO * 36);
            if (ci0 + ci < Cin && co0 + j / 36 < Cout) f(T[ci][j], ((long long)(ci0 + ci) * Cout + co0) * 36 + j);
        }
    };
    if (!UNPACK) {
        torch_pass([&](float& t, long long off) { t = src[off]; });
        __syncthreads();
        packed_pass([&](float& t, long long off) { dst[off] = t; });
    } else {
        packed_pass([&](float& t, long long off) { t = src[off]; });
        __syncthreads();
        torch_pass([&](float& t, long long off) { dst[off] += t; });      // dW += (accumulating gradient buffer)
    }
}
