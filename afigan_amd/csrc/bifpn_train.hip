// Backward pieces of a BiFPN node on gfx950 (bifpn_sr.py:531-733 under autograd; bifpn_layers/{wrappers,activations}.py):
//   fusion + swish backward, depthwise 3x3 weight gradient, zero-padded 3x3/2 max-pool with its argmax and the gather-form backward.
// The depthwise INPUT gradient is the forward kernel run with the taps reversed (afi_dwconv3x3_fwd), the pointwise conv and the norm
// use the GEMM / BatchNorm kernels of the other files.  All of these are HBM / L2-bound float4 passes over pixel-major tensors;
// every reduction is two-stage with a fixed summation order (no atomics): results are bit-reproducible run to run.
#include "../../include/afigan_hip.h"
#include "afi_common.h"

__device__ __forceinline__ float afi_bt_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

#define AFI_BT_MAX_BLOCKS 1024

static unsigned afi_bt_grid(long long work_items) {
    long long g = (work_items + 255) / 256;
    if (g > AFI_BT_MAX_BLOCKS) g = AFI_BT_MAX_BLOCKS;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ---------------------------------------------------------------- out = swish(s), s = w0*a + w1*b (+ w2*c): backward
//   ds = dout * sig(s) * (1 + s * (1 - sig(s)))       (MemoryEfficientSwish.backward, activations.py)
//   da = w0 * ds, db = w1 * ds, dc = w2 * ds          each written only when its pointer is given
//   dw_k = sum ds * {a, b, c}                          per-block partials [blocks][4], summed by the finalize kernel in block order
// One pass: 3-4 reads and up to 3 writes of n floats.
__global__ __launch_bounds__(256) void afi_fuse_swish_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                                 const float* __restrict__ w, const float* __restrict__ dout, float* __restrict__ da,
                                                                 float* __restrict__ db, float* __restrict__ dc, float* __restrict__ partial,
                                                                 long long n4) {
    __shared__ float red[3][4];
    const float w0 = w[0], w1 = w[1], w2 = c ? w[2] : 0.f;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 av = ((const f32x4*)a)[i], bv = ((const f32x4*)b)[i], g = ((const f32x4*)dout)[i];
        f32x4 cv = {0.f, 0.f, 0.f, 0.f};
        if (c) cv = ((const f32x4*)c)[i];
        f32x4 ds;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = w0 * av[j] + w1 * bv[j] + w2 * cv[j];
            const float sg = 1.f / (1.f + __expf(-s));
            ds[j] = g[j] * sg * (1.f + s * (1.f - sg));
            s0 += ds[j] * av[j]; s1 += ds[j] * bv[j]; s2 += ds[j] * cv[j];
        }
        if (da) ((f32x4*)da)[i] = w0 * ds;
        if (db) ((f32x4*)db)[i] = w1 * ds;
        if (dc) ((f32x4*)dc)[i] = w2 * ds;
    }
    s0 = afi_bt_wave_sum(s0); s1 = afi_bt_wave_sum(s1); s2 = afi_bt_wave_sum(s2);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wv] = s0; red[1][wv] = s1; red[2][wv] = s2; }
    __syncthreads();
    if (threadIdx.x < 3)
        partial[(long long)blockIdx.x * 4 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}
// dw[k] = sum over blocks, fp64, fixed order: 64 lanes per weight, each summing every 64th block's partial (independent loads in flight), the
// lanes meeting through LDS in lane order -- bit-reproducible, and 40 us -> a few (the one-thread-per-weight walk was 2048 dependent loads:
// 2.3 ms of a BiFPN training pass over its 56 nodes, profiles/r06/kernel_stats_bifpn_train_5iters.csv of the first pass)
__global__ __launch_bounds__(256) void afi_fuse_swish_bwd_finalize_kernel(const float* __restrict__ partial, int blocks, int nw, float* __restrict__ dw) {
    __shared__ double red[4][64];
    const int k = threadIdx.x & 3, ln = threadIdx.x >> 2;
    double s = 0.0;
    if (k < nw)
        for (int i = ln; i < blocks; i += 64) s += (double)partial[(long long)i * 4 + k];
    red[k][ln] = s;
    __syncthreads();
    if (ln == 0 && k < nw) {
        double t = 0.0;
        for (int j = 0; j < 64; ++j) t += red[k][j];
        dw[k] = (float)t;
    }
}
extern "C" long long afi_fuse_swish_bwd_scratch_floats(void) { return (long long)AFI_BT_MAX_BLOCKS * 4; }
extern "C" int afi_fuse_swish_bwd(const float* a, const float* b, const float* c_or_null, const float* w_dev, const float* dout, float* da_or_null,
                                  float* db_or_null, float* dc_or_null, float* dw_or_null, long long n, float* scratch, void* stream) {
    if (n <= 0 || (n & 3) || !a || !b || !w_dev || !dout || !scratch) return AFI_ERR_BAD_ARG;
    if (dc_or_null && !c_or_null) return AFI_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = afi_bt_grid(n >> 2);
    hipLaunchKernelGGL(afi_fuse_swish_bwd_kernel, dim3(blocks), dim3(256), 0, st, a, b, c_or_null, w_dev, dout, da_or_null, db_or_null, dc_or_null, scratch,
                       n >> 2);
    if (dw_or_null)
        hipLaunchKernelGGL(afi_fuse_swish_bwd_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)scratch, (int)blocks, c_or_null ? 3 : 2, dw_or_null);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// ---------------------------------------------------------------- depthwise 3x3 (zero pad 1) weight gradient
//   dw[t][c] = sum_{n,y,x} dy[n][y][x][c] * x[n][y + t/3 - 1][x + t%3 - 1][c]          w: [9][C] tap-major, as the forward takes it
// Block = 32 channel quads x 8 pixel lanes, grid (C/128, chunks of pixels); 9 float4 accumulators per thread; partials [chunks][9][C]
// are summed per (tap, channel) by the finalize kernel in chunk order.  x's nine taps of neighbouring pixels hit L2.
#define AFI_DWW_MAX_CHUNKS 256
__global__ __launch_bounds__(256) void afi_dwconv3x3_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, int N, int H, int W, int C,
                                                                  int rows_per_chunk, float* __restrict__ partial) {
    __shared__ f32x4 red[8][32];
    const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 128 + cq * 4;
    const bool cok = c < C;
    const long long P = (long long)N * H * W;
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (cok) {
        const long long r0 = (long long)blockIdx.y * rows_per_chunk;
        const long long r1 = (r0 + rows_per_chunk < P) ? r0 + rows_per_chunk : P;
        for (long long r = r0 + rl; r < r1; r += 8) {
            const int xx = (int)(r % W); const int yy = (int)((r / W) % H);
            const f32x4 g = *(const f32x4*)(dy + r * C + c);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int y2 = yy + t / 3 - 1, x2 = xx + t % 3 - 1;
                if ((unsigned)y2 < (unsigned)H && (unsigned)x2 < (unsigned)W)
                    acc[t] += g * *(const f32x4*)(x + (r + (long long)(t / 3 - 1) * W + (t % 3 - 1)) * C + c);
            }
        }
    }
    float* dst = partial + (long long)blockIdx.y * 9 * C;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        red[rl][cq] = acc[t];
        __syncthreads();
        if (rl == 0 && cok) {
            f32x4 s = red[0][cq];
#pragma unroll
            for (int i = 1; i < 8; ++i) s += red[i][cq];
            *(f32x4*)(dst + (long long)t * C + c) = s;
        }
        __syncthreads();
    }
}
__global__ void afi_dwconv3x3_wgrad_finalize_kernel(const float* __restrict__ partial, int chunks, int C, float* __restrict__ dw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // (tap, channel)
    if (i >= 9 * C) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;              // four independent chains (fixed assignment of chunks: bit-reproducible), four loads in flight
    int k = 0;
    for (; k + 3 < chunks; k += 4) {
        s0 += (double)partial[(long long)k * 9 * C + i];
        s1 += (double)partial[(long long)(k + 1) * 9 * C + i];
        s2 += (double)partial[(long long)(k + 2) * 9 * C + i];
        s3 += (double)partial[(long long)(k + 3) * 9 * C + i];
    }
    for (; k < chunks; ++k) s0 += (double)partial[(long long)k * 9 * C + i];
    dw[i] = (float)((s0 + s1) + (s2 + s3));
}
extern "C" long long afi_dwconv3x3_wgrad_scratch_floats(int C) { return (long long)AFI_DWW_MAX_CHUNKS * 9 * C; }
extern "C" int afi_dwconv3x3_wgrad(const float* dy, const float* x, int N, int H, int W, int C, float* dw9c, float* scratch, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || !dy || !x || !dw9c || !scratch) return AFI_ERR_BAD_ARG;
    if (C & 3) return AFI_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const long long P = (long long)N * H * W;
    long long want = (P + 63) / 64;
    if (want > AFI_DWW_MAX_CHUNKS) want = AFI_DWW_MAX_CHUNKS;
    const int rpc = (int)((P + want - 1) / want);
    const int chunks = (int)((P + rpc - 1) / rpc);
    hipLaunchKernelGGL(afi_dwconv3x3_wgrad_kernel, dim3(afi_cdiv(C, 128), chunks), dim3(256), 0, st, dy, x, N, H, W, C, rpc, scratch);
    hipLaunchKernelGGL(afi_dwconv3x3_wgrad_finalize_kernel, dim3(afi_cdiv(9 * C, 256)), dim3(256), 0, st, (const float*)scratch, chunks, C, dw9c);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// ---------------------------------------------------------------- MaxPool2d(3, 2, "static_same") with its argmax, and the backward
// Forward as afi_maxpool3s2_same_kernel, keeping per element the tap t = 3*dy + dx (0..8) of the FIRST maximum in scan order -- the
// element torch's max_pool2d routes the gradient to (strict '>' while scanning); the zero pad row / column takes part, and a window won
// by a pad element sends its gradient nowhere, as F.pad's backward drops it.
__global__ void afi_maxpool3s2_same_idx_kernel(const AfiView x, int N, int H, int W, int C, int Ho, int Wo, float* __restrict__ out,
                                               unsigned* __restrict__ idx) {
    const int C4 = C >> 2;
    const long long total = (long long)N * Ho * Wo * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4; long long r = i / C4;
        const int ox = (int)(r % Wo); r /= Wo; const int oy = (int)(r % Ho); const int n = (int)(r / Ho);
        f32x4 m = {-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
        unsigned am = 0;                                      // four argmax bytes
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int y2 = 2 * oy + t / 3, x2 = 2 * ox + t % 3;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (y2 < H && x2 < W) v = *(const f32x4*)(x.p + (long long)n * x.sN + (long long)y2 * x.sH + (long long)x2 * x.sW + c);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (v[j] > m[j] || t == 0) { m[j] = v[j]; am = (am & ~(0xFFu << (8 * j))) | ((unsigned)t << (8 * j)); }
        }
        *(f32x4*)(out + i * 4) = m;
        idx[i] = am;
    }
}
// dx[n][y][x][c] = sum over the (at most four) windows containing (y, x) whose argmax is this element of dout[window]
__global__ void afi_maxpool3s2_same_bwd_kernel(const float* __restrict__ dout, const unsigned* __restrict__ idx, int N, int H, int W, int C, int Ho, int Wo,
                                               float* __restrict__ dx) {
    const int C4 = C >> 2;
    const long long total = (long long)N * H * W * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4); long long r = i / C4;
        const int xx = (int)(r % W); r /= W; const int yy = (int)(r % H); const int n = (int)(r / H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int oy1 = yy >> 1, ox1 = xx >> 1;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oy = oy1 - a;
            const int ty = yy - 2 * oy;                        // 0/1 for a == 0, 2/3 for a == 1
            if (oy < 0 || oy >= Ho || ty > 2) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ox = ox1 - b;
                const int tx = xx - 2 * ox;
                if (ox < 0 || ox >= Wo || tx > 2) continue;
                const long long o = (((long long)n * Ho + oy) * Wo + ox) * C4 + c4;
                const unsigned am = idx[o];
                const f32x4 g = *(const f32x4*)(dout + o * 4);
                const unsigned t = (unsigned)(ty * 3 + tx);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (((am >> (8 * j)) & 0xFFu) == t) acc[j] += g[j];
            }
        }
        *(f32x4*)(dx + i * 4) = acc;
    }
}
extern "C" int afi_maxpool3s2_same_fwd_idx(afi_view_t xv, int N, int H, int W, int C, float* out, unsigned char* idx, void* stream) {
    if (N <= 0 || H < 2 || W < 2 || C <= 0 || !out || !idx) return AFI_ERR_BAD_ARG;
    if ((C & 3) || (((unsigned long long)idx) & 3)) return AFI_ERR_UNSUPPORTED;
    AfiView x; x.p = (float*)xv.p; x.sN = xv.sN; x.sH = xv.sH; x.sW = xv.sW;
    const int Ho = (H - 2) / 2 + 1, Wo = (W - 2) / 2 + 1;
    hipLaunchKernelGGL(afi_maxpool3s2_same_idx_kernel, dim3(afi_bt_grid((long long)N * Ho * Wo * (C >> 2))), dim3(256), 0, (hipStream_t)stream, x, N, H, W,
                       C, Ho, Wo, out, (unsigned*)idx);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
extern "C" int afi_maxpool3s2_same_bwd(const float* dout, const unsigned char* idx, int N, int H, int W, int C, float* dx, void* stream) {
    if (N <= 0 || H < 2 || W < 2 || C <= 0 || !dout || !idx || !dx) return AFI_ERR_BAD_ARG;
    if ((C & 3) || (((unsigned long long)idx) & 3)) return AFI_ERR_UNSUPPORTED;
    const int Ho = (H - 2) / 2 + 1, Wo = (W - 2) / 2 + 1;
    hipLaunchKernelGGL(afi_maxpool3s2_same_bwd_kernel, dim3(afi_bt_grid((long long)N * H * W * (C >> 2))), dim3(256), 0, (hipStream_t)stream, dout,
                       (const unsigned*)idx, N, H, W, C, Ho, Wo, dx);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
