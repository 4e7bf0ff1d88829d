// HBM-bound kernels of the AFI-GAN hot path on gfx950: layout changes, conv-transpose weight (un)packing,
// BatchNorm statistics / apply / backward, column sums (bias grads), the 1024->1 stencil of the last
// discriminator conv, BCE-with-logits, L1, bilinear x2 (standalone + backward) and multi-tensor SGD.
// All of them are float4-vectorised along the contiguous channel dimension of the pixel-major layout;
// reductions use wavefront shuffles (64 lanes) + one LDS hop per block.
#include "afi_common.h"
#include "afi_bilinear.h"
#include "afi_convt_pack.h"
#include "afi_bn.h"
#include <stdlib.h>
#include <string.h>

#define AFI_BN_EPS 1e-5f
#define AFI_BN_MOMENTUM 0.1f

__device__ __forceinline__ float afi_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------- layout: NCHW <-> NHWC
// in [N][C][P] -> out [N][P][C]   (P = H*W); 32x32 LDS tile, +1 pad
__global__ __launch_bounds__(256) void afi_nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int P) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = in + (long long)n * C * P;
    float* dst = out + (long long)n * C * P;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = c0 + ty + 8 * i, pp = p0 + tx;
        if (c < C && pp < P) tile[ty + 8 * i][tx] = src[(long long)c * P + pp];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int pp = p0 + ty + 8 * i, c = c0 + tx;
        if (c < C && pp < P) dst[(long long)pp * C + c] = tile[tx][ty + 8 * i];
    }
}
__global__ __launch_bounds__(256) void afi_nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int P) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = in + (long long)n * C * P;
    float* dst = out + (long long)n * C * P;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int pp = p0 + ty + 8 * i, c = c0 + tx;
        if (c < C && pp < P) tile[ty + 8 * i][tx] = src[(long long)pp * C + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = c0 + ty + 8 * i, pp = p0 + tx;
        if (c < C && pp < P) dst[(long long)c * P + pp] = tile[tx][ty + 8 * i];
    }
}

int afi_launch_nchw_to_nhwc(const float* in, float* out, int N, int C, int P, hipStream_t st) {
    if (N <= 0 || C <= 0 || P <= 0) return AFI_ERR_BAD_ARG;
    hipLaunchKernelGGL(afi_nchw_to_nhwc_kernel, dim3(afi_cdiv(P, 32), afi_cdiv(C, 32), N), dim3(256), 0, st, in, out, C, P);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
int afi_launch_nhwc_to_nchw(const float* in, float* out, int N, int C, int P, hipStream_t st) {
    if (N <= 0 || C <= 0 || P <= 0) return AFI_ERR_BAD_ARG;
    hipLaunchKernelGGL(afi_nhwc_to_nchw_kernel, dim3(afi_cdiv(P, 32), afi_cdiv(C, 32), N), dim3(256), 0, st, in, out, C, P);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// ---------------------------------------------------------------- conv-transpose weight pack / grad unpack (body: afi_convt_pack.h)
template <bool UNPACK>
__global__ __launch_bounds__(256) void afi_convT_repack_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cin, int Cout) {
    __shared__ float T[AFI_CT_CI][AFI_CT_LD];
    afi_convT_repack_body<UNPACK>(src, dst, Cin, Cout, blockIdx.x, blockIdx.y, T);
}
int afi_launch_convT_pack(const float* W, float* Wp, int Cin, int Cout, hipStream_t st) {
    hipLaunchKernelGGL(afi_convT_repack_kernel<false>, dim3((Cin + AFI_CT_CI - 1) / AFI_CT_CI, (Cout + AFI_CT_CO - 1) / AFI_CT_CO), dim3(256), 0, st, W, Wp, Cin, Cout);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
int afi_launch_convT_unpack_grad(const float* dWp, float* dW, int Cin, int Cout, hipStream_t st) {
    hipLaunchKernelGGL(afi_convT_repack_kernel<true>, dim3((Cin + AFI_CT_CI - 1) / AFI_CT_CI, (Cout + AFI_CT_CO - 1) / AFI_CT_CO), dim3(256), 0, st, dWp, dW, Cin, Cout);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// Dense-block growth convs, weight gradients of a block in ONE GEMM (nets.hip: afi_generator_bwd): the four gradients dy_1..dy_4 are the
// adjacent G-channel slices [C, C + 4G) of the block's gradient buffer and every conv reads a prefix of the block's activation buffer, so
// dWp[4G][3][3][L] = dy[C : C + 4G] (x) cat[0 : L] holds all four -- rows (k-1)G .. kG, columns c < C + (k-1)G are conv k's gradient; the rest
// (a conv paired with channels that come after it) is never read.  This kernel adds the valid part into the four parameter gradients.
__global__ __launch_bounds__(256) void afi_rdb_wgrad_unpack_kernel(const float* __restrict__ dWp, float* dw1, float* dw2, float* dw3, float* dw4, int C, int G, float alpha) {
    const int L = C + 4 * G;
    const long long total = 4LL * G * 9 * (L >> 2);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % (L >> 2)) * 4;
        const long long rt = e / (L >> 2);                   // row * 9 + tap
        const int row = (int)(rt / 9), tap = (int)(rt - 9LL * row);
        const int k = row / G, o = row - k * G;              // conv k + 1
        const int cin = C + k * G;
        if (c4 >= cin) continue;
        float* dw = k == 0 ? dw1 : (k == 1 ? dw2 : (k == 2 ? dw3 : dw4));
        if (!dw) continue;
        const f32x4 v = *(const f32x4*)(dWp + rt * L + c4);
        f32x4* dst = (f32x4*)(dw + ((long long)o * 9 + tap) * cin + c4);
        f32x4 d = *dst;
        d += alpha * v;
        *dst = d;
    }
}
// the same for several dense blocks in one launch (small maps: a launch per block costs more than the work): blockIdx.y = block
struct AfiRdbUnpackMulti { const float* dWp; long long stride; float* dw[8][4]; };
__device__ __forceinline__ void afi_rdb_wgrad_unpack_multi_body(const AfiRdbUnpackMulti& t, int C, int G, float alpha, int bx, int nbx, int r) {
    const float* __restrict__ dWp = t.dWp + (long long)r * t.stride;
    const int L = C + 4 * G;
    const long long total = 4LL * G * 9 * (L >> 2);
    for (long long e = (long long)bx * blockDim.x + threadIdx.x; e < total; e += (long long)nbx * blockDim.x) {
        const int c4 = (int)(e % (L >> 2)) * 4;
        const long long rt = e / (L >> 2);
        const int row = (int)(rt / 9), tap = (int)(rt - 9LL * row);
        const int k = row / G, o = row - k * G;
        const int cin = C + k * G;
        if (c4 >= cin) continue;
        float* dw = t.dw[r][k];
        if (!dw) continue;
        const f32x4 v = *(const f32x4*)(dWp + rt * L + c4);
        f32x4* dst = (f32x4*)(dw + ((long long)o * 9 + tap) * cin + c4);
        f32x4 d = *dst;
        d += alpha * v;
        *dst = d;
    }
}
__global__ __launch_bounds__(256) void afi_rdb_wgrad_unpack_multi_kernel(const AfiRdbUnpackMulti t, int C, int G, float alpha) {
    afi_rdb_wgrad_unpack_multi_body(t, C, G, alpha, blockIdx.x, gridDim.x, blockIdx.y);
}
int afi_launch_rdb_wgrad_unpack_multi(const float* dWp, long long stride, float* const (*dw)[4], int nblocks, int C, int G, float alpha, hipStream_t st) {
    if (!dWp || nblocks <= 0 || nblocks > 8 || C <= 0 || G <= 0 || (C & 3) || (G & 3)) return AFI_ERR_BAD_ARG;
    AfiRdbUnpackMulti t;
    t.dWp = dWp; t.stride = stride;
    for (int r = 0; r < 8; ++r)
        for (int k = 0; k < 4; ++k) t.dw[r][k] = r < nblocks ? dw[r][k] : nullptr;
    const long long total = 4LL * G * 9 * ((C + 4 * G) >> 2);
    long long blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(afi_rdb_wgrad_unpack_multi_kernel, dim3((unsigned)blocks, (unsigned)nblocks), dim3(256), 0, st, t, C, G, alpha);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
int afi_launch_rdb_wgrad_unpack(const float* dWp, float* const dw[4], int C, int G, float alpha, hipStream_t st) {
    if (!dWp || C <= 0 || G <= 0 || (C & 3) || (G & 3)) return AFI_ERR_BAD_ARG;
    const long long total = 4LL * G * 9 * ((C + 4 * G) >> 2);
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(afi_rdb_wgrad_unpack_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dWp, dw[0], dw[1], dw[2], dw[3], C, G, alpha);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------------
// grouped bias gradients: db[c] += alpha * sum_rows g[row][c] for up to 8 matrices in one launch (fp32 atomics: a few thousand adds)
// ------------------------------------------------------------------------------------------------
#define AFI_CS_MAXP 8
struct AfiColsumGroup {
    int nprob;
    int blk_start[AFI_CS_MAXP + 1];
    struct { const float* g; float* db; long long P, ld; int C; int rows_per_blk; float alpha; int pad; } d[AFI_CS_MAXP];
};
__device__ __forceinline__ void afi_colsum_group_body(const AfiColsumGroup& grp, const int b) {
    __shared__ f32x4 red[16][16];
    int pi = 0;
    while (pi + 1 < grp.nprob && b >= grp.blk_start[pi + 1]) ++pi;
    const auto& d = grp.d[pi];
    const int lbk = b - grp.blk_start[pi];
    const int ccs = (d.C + 63) / 64;                       // 64-channel column groups
    const int cg = lbk % ccs, rc = lbk / ccs;
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = cg * 64 + cq * 4;
    const long long r0 = (long long)rc * d.rows_per_blk;
    const long long r1 = (r0 + d.rows_per_blk < d.P) ? r0 + d.rows_per_blk : d.P;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (c < d.C)
        for (long long r = r0 + rl; r < r1; r += 16) s += *(const f32x4*)(d.g + r * d.ld + c);
    red[rl][cq] = s;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int cc = threadIdx.x >> 2, j = threadIdx.x & 3;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) v += red[i][cc][j];
        const int ch = cg * 64 + cc * 4 + j;
        if (ch < d.C) atomicAdd(d.db + ch, d.alpha * v);
    }
}
__global__ __launch_bounds__(256) void afi_colsum_group_kernel(const AfiColsumGroup grp) { afi_colsum_group_body(grp, blockIdx.x); }
static int afi_colsum_group_fill(const AfiColsumProb* probs, int n, AfiColsumGroup& grp, int& blocks) {
    if (n > AFI_CS_MAXP) return AFI_ERR_BAD_ARG;
    grp.nprob = n;
    blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (probs[i].C & 3) return AFI_ERR_UNSUPPORTED;
        const int rpb = 128;                               // rows per block: 850 rows -> 7 blocks per column group
        grp.d[i].g = probs[i].g; grp.d[i].db = probs[i].db; grp.d[i].P = probs[i].P; grp.d[i].ld = probs[i].ld;
        grp.d[i].C = probs[i].C; grp.d[i].rows_per_blk = rpb; grp.d[i].alpha = probs[i].alpha; grp.d[i].pad = 0;
        grp.blk_start[i] = blocks;
        blocks += ((probs[i].C + 63) / 64) * afi_cdiv(probs[i].P, rpb);
    }
    for (int i = n; i <= AFI_CS_MAXP; ++i) grp.blk_start[i] = blocks;
    return AFI_OK;
}
int afi_launch_colsum_group(const AfiColsumProb* probs, int n, hipStream_t st) {
    if (n <= 0) return AFI_OK;
    AfiColsumGroup grp;
    int blocks = 0;
    { const int rc = afi_colsum_group_fill(probs, n, grp, blocks); if (rc != AFI_OK) return rc; }
    hipLaunchKernelGGL(afi_colsum_group_kernel, dim3((unsigned)blocks), dim3(256), 0, st, grp);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// The tail of a small-map generator backward in ONE launch: the grouped bias gradients, the conv-transpose gradient's unpack and the dense
// blocks' packed growth-conv gradients' unpack -- three launches of 5 .. 9 us each whose work is a fraction of that (the first depends on
// nothing of the other two; the unpacks read what the grouped weight-gradient launch in front of this one wrote).  Blocks [0, nb_cs) run the
// column sums, [nb_cs, nb_cs + nb_ct) the conv-transpose tiles (ctx x cty), the rest the dense blocks (nb_rdb per block).  Same bodies as the
// three kernels above: same bits.
struct AfiGenBwdTail {
    AfiColsumGroup cs; int nb_cs;
    const float* ct_src; float* ct_dst; int ct_Cin, ct_Cout, ct_bx, nb_ct;
    AfiRdbUnpackMulti rdb; int rdb_C, rdb_G, nb_rdb, n_rdb; float rdb_alpha; int pad_;
};
__global__ __launch_bounds__(256) void afi_g_bwd_tail_kernel(const AfiGenBwdTail t) {
    int b = blockIdx.x;                                     // (every branch below is uniform per block)
    if (b < t.nb_cs) { afi_colsum_group_body(t.cs, b); return; }
    b -= t.nb_cs;
    if (b < t.nb_ct) {
        __shared__ float T[AFI_CT_CI][AFI_CT_LD];
        afi_convT_repack_body<true>(t.ct_src, t.ct_dst, t.ct_Cin, t.ct_Cout, b % t.ct_bx, b / t.ct_bx, T);
        return;
    }
    b -= t.nb_ct;
    afi_rdb_wgrad_unpack_multi_body(t.rdb, t.rdb_C, t.rdb_G, t.rdb_alpha, b % t.nb_rdb, t.nb_rdb, b / t.nb_rdb);
}
// any of the three parts may be absent (n_cs = 0, dWpT = null, nblocks = 0)
int afi_launch_g_bwd_tail(const AfiColsumProb* cs, int n_cs, const float* dWpT, float* dWT, int Cin, int Cout,
                          const float* dWp, long long stride, float* const (*dw)[4], int nblocks, int C, int G, float alpha, hipStream_t st) {
    AfiGenBwdTail t;
    memset(&t, 0, sizeof(t));
    if (n_cs > 0) { const int rc = afi_colsum_group_fill(cs, n_cs, t.cs, t.nb_cs); if (rc != AFI_OK) return rc; }
    if (dWpT) {
        if (!dWT || Cin <= 0 || Cout <= 0) return AFI_ERR_BAD_ARG;
        t.ct_src = dWpT; t.ct_dst = dWT; t.ct_Cin = Cin; t.ct_Cout = Cout;
        t.ct_bx = (Cin + AFI_CT_CI - 1) / AFI_CT_CI;
        t.nb_ct = t.ct_bx * ((Cout + AFI_CT_CO - 1) / AFI_CT_CO);
    }
    if (nblocks > 0) {
        if (!dWp || nblocks > 8 || C <= 0 || G <= 0 || (C & 3) || (G & 3)) return AFI_ERR_BAD_ARG;
        t.rdb.dWp = dWp; t.rdb.stride = stride;
        for (int r = 0; r < 8; ++r)
            for (int k = 0; k < 4; ++k) t.rdb.dw[r][k] = r < nblocks ? dw[r][k] : nullptr;
        const long long total = 4LL * G * 9 * ((C + 4 * G) >> 2);
        long long blocks = (total + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        t.rdb_C = C; t.rdb_G = G; t.nb_rdb = (int)blocks; t.n_rdb = nblocks; t.rdb_alpha = alpha;
    }
    const long long grid = (long long)t.nb_cs + t.nb_ct + (long long)t.nb_rdb * t.n_rdb;
    if (grid <= 0) return AFI_OK;
    hipLaunchKernelGGL(afi_g_bwd_tail_kernel, dim3((unsigned)grid), dim3(256), 0, st, t);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// ... and the data-gradient side of the same idea: the growth convs' weights on the block input x, Wx[(k-1)G + o][tap][c] = w_k[o][tap][c], c < C
__global__ __launch_bounds__(256) void afi_rdb_xpart_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ w3,
                                                                 const float* __restrict__ w4, float* __restrict__ out, int C, int G) {
    const long long total = 4LL * G * 9 * (C >> 2);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % (C >> 2)) * 4;
        const long long rt = e / (C >> 2);
        const int row = (int)(rt / 9), tap = (int)(rt - 9LL * row);
        const int k = row / G, o = row - k * G;
        const int cin = C + k * G;
        const float* w = k == 0 ? w1 : (k == 1 ? w2 : (k == 2 ? w3 : w4));
        *(f32x4*)(out + rt * C + c4) = *(const f32x4*)(w + ((long long)o * 9 + tap) * cin + c4);
    }
}
int afi_launch_rdb_xpart_pack(const float* const w[4], float* out, int C, int G, hipStream_t st) {
    if (!out || !w[0] || !w[1] || !w[2] || !w[3] || C <= 0 || G <= 0 || (C & 3) || (G & 3)) return AFI_ERR_BAD_ARG;
    const long long total = 4LL * G * 9 * (C >> 2);
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(afi_rdb_xpart_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w[0], w[1], w[2], w[3], out, C, G);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// in-place LeakyReLU on nch channels of a pixel-major view (the slice of growth conv 1 after the batched forward step: nets.hip)
__global__ __launch_bounds__(256) void afi_lrelu_slice_kernel(AfiView v, int N, int H, int W, int nch) {
    const int c4n = nch >> 2;
    const long long total = (long long)N * H * W * c4n;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % c4n) * 4;
        long long pix = e / c4n;
        const int x = (int)(pix % W); pix /= W;
        const int y = (int)(pix % H); const int n = (int)(pix / H);
        f32x4* p = (f32x4*)(v.p + (long long)n * v.sN + (long long)y * v.sH + (long long)x * v.sW + c4);
        f32x4 t = *p;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = t[j] > 0.f ? t[j] : t[j] * AFI_LRELU_SLOPE;
        *p = t;
    }
}
int afi_launch_lrelu_slice(AfiView v, int N, int H, int W, int nch, hipStream_t st) {
    if (!v.p || N <= 0 || H <= 0 || W <= 0 || nch <= 0 || (nch & 3)) return AFI_ERR_BAD_ARG;
    const long long total = (long long)N * H * W * (nch >> 2);
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(afi_lrelu_slice_kernel, dim3((unsigned)blocks), dim3(256), 0, st, v, N, H, W, nch);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// (the BatchNorm affine afi_bn_affine and its LeakyReLU: afi_bn.h)

// ---------------------------------------------------------------- per-channel reductions over pixels
// Generic two-value column reduction over a dense [P][C] matrix (ld = C, C % 4 == 0).
//   MODE 0: s0 = sum (x - K),      s1 = sum (x - K)^2         K = x[0][c]      (BatchNorm statistics)
//   MODE 1: s0 = sum g,            s1 = sum g * xhat          xhat = (x-mean)*invstd  (BatchNorm backward)
//   MODE 2: s0 = sum g             (bias gradient)
//   MODE 3: MODE 1 with g first multiplied by the LeakyReLU' mask of the normalised value (slope where z <= 0): the gradient arrives
//           w.r.t. the ACTIVATION and the mask is recomputed from x here, instead of being streamed through the producing conv's epilogue
// Block = 256 threads = 32 channel quads x 8 pixel lanes; grid = (C/128, chunks). Partials [chunks][2][C].
#define AFI_RED_MAX_CHUNKS 256
template <int MODE>
__global__ __launch_bounds__(256) void afi_colred_partial_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 long long P, int C, long long ld, int rows_per_chunk, float* __restrict__ partial,
                                                                 const float* __restrict__ gamma = nullptr, const float* __restrict__ beta = nullptr,
                                                                 float slope = 1.f) {
    __shared__ f32x4 red[2][8][32];
    const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 128 + cq * 4;
    const bool cok = c < C;
    f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
    if (cok) {
        f32x4 k = {0, 0, 0, 0}, mu = {0, 0, 0, 0}, is = {0, 0, 0, 0}, ga = {0, 0, 0, 0}, be = {0, 0, 0, 0};
        if (MODE == 0) k = *(const f32x4*)(x + c);
        if (MODE == 1 || MODE == 3) { mu = *(const f32x4*)(mean + c); is = *(const f32x4*)(invstd + c); }
        if (MODE == 3) { ga = *(const f32x4*)(gamma + c); be = *(const f32x4*)(beta + c); }
        const long long r0 = (long long)blockIdx.y * rows_per_chunk;
        const long long r1 = (r0 + rows_per_chunk < P) ? r0 + rows_per_chunk : P;
        for (long long r = r0 + rl; r < r1; r += 8) {
            if (MODE == 0) {
                f32x4 v = *(const f32x4*)(x + r * ld + c) - k;
                s0 += v; s1 += v * v;
            } else if (MODE == 1) {
                f32x4 gv = *(const f32x4*)(g + r * ld + c);
                f32x4 xh = (*(const f32x4*)(x + r * ld + c) - mu) * is;
                s0 += gv; s1 += gv * xh;
            } else if (MODE == 3) {
                f32x4 gv = *(const f32x4*)(g + r * ld + c);
                const f32x4 xv = *(const f32x4*)(x + r * ld + c);
                const f32x4 z = afi_bn_affine(xv, mu, is, ga, be);
#pragma unroll
                for (int j = 0; j < 4; ++j) gv[j] = z[j] > 0.f ? gv[j] : gv[j] * slope;
                const f32x4 xh = (xv - mu) * is;
                s0 += gv; s1 += gv * xh;
            } else {
                s0 += *(const f32x4*)(g + r * ld + c);
            }
        }
    }
    red[0][rl][cq] = s0; red[1][rl][cq] = s1;
    __syncthreads();
    if (rl == 0 && cok) {
#pragma unroll
        for (int i = 1; i < 8; ++i) { s0 += red[0][i][cq]; s1 += red[1][i][cq]; }
        float* dst = partial + (long long)blockIdx.y * 2 * C;
        *(f32x4*)(dst + c) = s0;
        *(f32x4*)(dst + C + c) = s1;
    }
}

// Sum the per-chunk partials [chunks][2][C] of one channel: a 256-thread block owns 32 channels, 8 threads per channel take
// every 8th chunk (independent loads in flight) and are combined through LDS in a fixed order (bit-reproducible).  The serial
// one-thread-per-channel loop this replaces cost ~35 us per call (256 dependent L2 round trips), ~90 calls per stage-1 step.
#define AFI_FIN_CH 32
__device__ __forceinline__ bool afi_chunk_sums(const float* __restrict__ partial, int chunks, int C, bool two, int& c, float& s0, float& s1) {
    __shared__ float red[2][8][AFI_FIN_CH];
    const int cl = threadIdx.x & (AFI_FIN_CH - 1), ln = threadIdx.x / AFI_FIN_CH;
    c = blockIdx.x * AFI_FIN_CH + cl;
    float a0 = 0.f, a1 = 0.f;
    if (c < C)
        for (int i = ln; i < chunks; i += 8) {
            a0 += partial[(long long)i * 2 * C + c];
            if (two) a1 += partial[(long long)i * 2 * C + C + c];
        }
    red[0][ln][cl] = a0; red[1][ln][cl] = a1;
    __syncthreads();
    if (ln != 0 || c >= C) return false;
    s0 = 0.f; s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0 += red[0][j][cl]; s1 += red[1][j][cl]; }
    return true;
}

// BatchNorm batch statistics in fp64.  The normalised value decides a LeakyReLU mask, and one flipped mask element moves a whole
// gradient tensor by ~1e-3 (tools/d_parity_probe.py: with identical masks the backward is exact to 2e-6), so the statistics are kept
// as accurate as torch's CPU path, whose accumulation type for float is double: per-thread sums, the chunk partials, the finalize
// and 1/sqrt(var + eps) are all fp64 (the one-pass fp32 form measured 0.9-1.6e-6 relative in var, 5-8x the reference's deviation).
// Shifted by K = x[0][c] so that s1/P - d^2 does not cancel.  The pass stays bandwidth-bound (12 fp64 ops per 16 bytes).
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void afi_bn_stats_partial_kernel(const float* __restrict__ x, long long P, int C, long long ld, int rows_per_chunk,
                                                                   double* __restrict__ partial) {
    __shared__ f64x4 red[2][8][32];
    const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 128 + cq * 4;
    const bool cok = c < C;
    f64x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
    if (cok) {
        const f32x4 k = *(const f32x4*)(x + c);
        const long long r0 = (long long)blockIdx.y * rows_per_chunk;
        const long long r1 = (r0 + rows_per_chunk < P) ? r0 + rows_per_chunk : P;
        for (long long r = r0 + rl; r < r1; r += 8) {
            const f32x4 v = *(const f32x4*)(x + r * ld + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) { const double d = (double)v[j] - (double)k[j]; s0[j] += d; s1[j] += d * d; }
        }
    }
    red[0][rl][cq] = s0; red[1][rl][cq] = s1;
    __syncthreads();
    if (rl == 0 && cok) {
#pragma unroll
        for (int i = 1; i < 8; ++i) { s0 += red[0][i][cq]; s1 += red[1][i][cq]; }
        double* dst = partial + (long long)blockIdx.y * 2 * C;
        *(f64x4*)(dst + c) = s0;
        *(f64x4*)(dst + C + c) = s1;
    }
}
// CH channels per 256-thread block, 256 / CH lanes per channel (each sums every (256 / CH)-th row, independent loads in flight; the lanes
// meet through LDS in a fixed order: bit-reproducible).  CH = 32 for the <= 256 rows of the separate statistics pass, CH = 8 (32 lanes per
// channel, 4x the blocks) for the up-to-1024 rows the Winograd output transforms leave.
template <int CH>
__global__ void afi_bn_stats_finalize64_kernel(const double* __restrict__ partial, int chunks, const float* __restrict__ x0, long long P, int C,
                                               float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ var_out,
                                               float* __restrict__ running_mean, float* __restrict__ running_var,
                                               long long* __restrict__ num_batches_tracked, float eps, float momentum) {
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;     // torch BatchNorm2d train mode
    constexpr int LN = 256 / CH;
    __shared__ double red[2][LN][CH];
    const int cl = threadIdx.x & (CH - 1), ln = threadIdx.x / CH;
    const int c = blockIdx.x * CH + cl;
    double a0 = 0.0, a1 = 0.0;
    if (c < C) {
        int i = ln;
        for (; i + 3 * LN < chunks; i += 4 * LN) {            // four rows in flight per lane
            const double p0 = partial[(long long)i * 2 * C + c], q0 = partial[(long long)i * 2 * C + C + c];
            const double p1 = partial[(long long)(i + LN) * 2 * C + c], q1 = partial[(long long)(i + LN) * 2 * C + C + c];
            const double p2 = partial[(long long)(i + 2 * LN) * 2 * C + c], q2 = partial[(long long)(i + 2 * LN) * 2 * C + C + c];
            const double p3 = partial[(long long)(i + 3 * LN) * 2 * C + c], q3 = partial[(long long)(i + 3 * LN) * 2 * C + C + c];
            a0 += p0; a0 += p1; a0 += p2; a0 += p3;
            a1 += q0; a1 += q1; a1 += q2; a1 += q3;
        }
        for (; i < chunks; i += LN) { a0 += partial[(long long)i * 2 * C + c]; a1 += partial[(long long)i * 2 * C + C + c]; }
    }
    red[0][ln][cl] = a0; red[1][ln][cl] = a1;
    __syncthreads();
    if (ln != 0 || c >= C) return;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int j = 0; j < LN; ++j) { s0 += red[0][j][cl]; s1 += red[1][j][cl]; }        // fixed order: bit-reproducible
    const double inv_n = 1.0 / (double)P;
    const double d = s0 * inv_n;                    // mean - K
    const double m = (x0 ? (double)x0[c] : 0.0) + d;     // (x0 == nullptr: unshifted sums, e.g. the partials of the Winograd output transforms)
    double var = s1 * inv_n - d * d;                // biased
    var = var > 0.0 ? var : 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (var_out) var_out[c] = (float)var;
    if (running_mean) {
        const double unb = var * ((double)P / (double)(P > 1 ? P - 1 : 1));
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * m);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
    }
}

// BatchNorm statistics finalize: mean / invstd for this call + running-stat update (momentum 0.1, unbiased var)
__global__ void afi_bn_stats_finalize_kernel(const float* __restrict__ partial, int chunks, const float* __restrict__ x0, long long P, int C,
                                             float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ var_out,
                                             float* __restrict__ running_mean, float* __restrict__ running_var,
                                             long long* __restrict__ num_batches_tracked, float eps, float momentum) {
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;     // torch BatchNorm2d train mode
    int c; float s0, s1;
    if (!afi_chunk_sums(partial, chunks, C, true, c, s0, s1)) return;
    const float inv_n = 1.f / (float)P;
    const float d = s0 * inv_n;                     // mean - K
    const float m = x0[c] + d;
    float var = s1 * inv_n - d * d;                 // biased
    var = fmaxf(var, 0.f);
    mean[c] = m;
    invstd[c] = rsqrtf(var + eps);
    if (var_out) var_out[c] = var;
    if (running_mean) {
        const float unb = var * ((float)P / (float)(P > 1 ? P - 1 : 1));
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
    }
}

// y = lrelu_slope((x - mean) * invstd * gamma + beta); slope 1 = the plain affine (BiFPN's norms have no activation behind them)
// largest magnitude of what a pass WRITES, raised into a zero-filled slot as a by-product (the f16x3 arithmetic of the Winograd GEMMs scales
// its operands by a power of two derived from it, csrc/afi_gemm_f16.h; the slot is then known BEFORE the next convolution's transforms
// run, which lets them write their planes already split into fp16 pieces).  One conditional atomic max per block.
__device__ __forceinline__ float afi_ew_amax4(float m, f32x4 v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
}
__device__ __forceinline__ void afi_ew_amax_publish(float m, float* slot) {
    __shared__ float red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const unsigned bits = __float_as_uint(m);
        if (bits > __hip_atomic_load((const unsigned*)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) (void)atomicMax((unsigned*)slot, bits);
    }
}
template <bool AMAX>
__global__ void afi_bn_apply_lrelu_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ mean,
                                          const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                          long long P, int C, float slope, float* amax) {
    const long long total4 = P * C / 4;
    const int C4 = C / 4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float am = 0.f;
    auto apply = [slope, &am](f32x4 v, f32x4 mu, f32x4 is, f32x4 ga, f32x4 be) {
        const f32x4 r = afi_bn_lrelu(v, mu, is, ga, be, slope);
        if (AMAX) am = afi_ew_amax4(am, r);
        return r;
    };
    if (stride % C4 == 0) {
        // a thread stays on one channel group: its four parameter vectors are loaded once, and four independent 16-B loads are
        // kept in flight per thread (a streaming kernel needs ~37 KB outstanding per CU to cover the HBM latency)
        const int c = (int)(i % C4) * 4;
        const f32x4 mu = *(const f32x4*)(mean + c), is = *(const f32x4*)(invstd + c);
        const f32x4 ga = *(const f32x4*)(gamma + c), be = *(const f32x4*)(beta + c);
        for (; i + 3 * stride < total4; i += 4 * stride) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load((const f32x4*)(x + (i + u * stride) * 4));
#pragma unroll
            for (int u = 0; u < 4; ++u) *(f32x4*)(y + (i + u * stride) * 4) = apply(v[u], mu, is, ga, be);
        }
        for (; i < total4; i += stride) *(f32x4*)(y + i * 4) = apply(*(const f32x4*)(x + i * 4), mu, is, ga, be);
        if (AMAX) afi_ew_amax_publish(am, amax);
        return;
    }
    for (; i < total4; i += stride) {
        const int c = (int)(i % C4) * 4;
        *(f32x4*)(y + i * 4) = apply(*(const f32x4*)(x + i * 4), *(const f32x4*)(mean + c), *(const f32x4*)(invstd + c),
                                     *(const f32x4*)(gamma + c), *(const f32x4*)(beta + c));
    }
    if (AMAX) afi_ew_amax_publish(am, amax);
}

// The largest magnitude of a tensor given as a view ([N][H][W][C], C contiguous, any pixel strides): one streaming pass, one conditional atomic
// per block into a zero-filled slot.  For the discriminator's FIRST conv under f16x3: its input arrives from outside the library (guide
// features, or the interpolator's output cropped by _reshape_stage1), so no producer has published its maximum; with this pass in front the
// first block's planes are written pre-split like every other block's and its GEMM can take the k-step-local sums.
__global__ __launch_bounds__(256) void afi_view_absmax_kernel(const AfiView x, int N, int H, int W, int C, float* amax) {
    const int C4 = C >> 2;
    const long long total = (long long)N * H * W * C4;
    float am = 0.f;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C4) * 4;
        long long r = e / C4;
        const int xx = (int)(r % W); r /= W;
        const int yy = (int)(r % H); const int n = (int)(r / H);
        am = afi_ew_amax4(am, __builtin_nontemporal_load((const f32x4*)(x.p + (long long)n * x.sN + (long long)yy * x.sH + (long long)xx * x.sW + c)));
    }
    afi_ew_amax_publish(am, amax);
}
int afi_launch_view_absmax(AfiView x, int N, int H, int W, int C, float* amax, hipStream_t st) {
    if (!x.p || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || !amax) return AFI_ERR_BAD_ARG;
    long long grid = ((long long)N * H * W * (C >> 2) + 255) / 256;
    grid = grid > 2048 ? 2048 : (grid < 1 ? 1 : grid);      // 256 CUs x 8 blocks, grid-stride the rest (afi_ew_grid, defined further down)
    hipLaunchKernelGGL(afi_view_absmax_kernel, dim3((unsigned)grid), dim3(256), 0, st, x, N, H, W, C, amax);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// The largest magnitude of an activation y = lrelu(affine(c)) that NO kernel has evaluated yet, from the per-channel minimum / maximum of the
// conv output c (fp32 rows [rows][2][C] the Winograd output transforms leave beside their statistics partials): the pinned affine of
// afi_bn.h is monotonic per channel in c, every one of its fp32 operations is monotonic under rounding, and LeakyReLU is increasing, so the
// extreme activations of a channel are the affine's values at its two extreme conv outputs -- attained values, hence the EXACT maximum the
// apply pass would have published.  One 256-thread block per 8 channels (32 lanes per channel over the rows), one conditional atomic per block.
__global__ __launch_bounds__(256) void afi_bn_act_amax_kernel(const float* __restrict__ mm, int rows, int C, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float slope, float* amax) {
    const int cl = threadIdx.x & 7, ln = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    float mn = INFINITY, mx = -INFINITY;
    if (c < C)
        for (int i = ln; i < rows; i += 32) { mn = fminf(mn, mm[(long long)i * 2 * C + c]); mx = fmaxf(mx, mm[(long long)i * 2 * C + C + c]); }
    float am = 0.f;
    if (c < C && mn <= mx) {
        const f32x4 mu = {mean[c], 0.f, 0.f, 0.f}, is = {invstd[c], 0.f, 0.f, 0.f}, ga = {gamma[c], 0.f, 0.f, 0.f}, be = {beta[c], 0.f, 0.f, 0.f};
        const f32x4 lo = afi_bn_lrelu(f32x4{mn, 0.f, 0.f, 0.f}, mu, is, ga, be, slope), hi = afi_bn_lrelu(f32x4{mx, 0.f, 0.f, 0.f}, mu, is, ga, be, slope);
        am = fmaxf(fabsf(lo[0]), fabsf(hi[0]));
    }
    afi_ew_amax_publish(am, amax);
}
int afi_launch_bn_act_amax(const float* mm, int rows, int C, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                           float* amax, hipStream_t st) {
    if (!mm || rows <= 0 || C <= 0 || !amax) return AFI_ERR_BAD_ARG;
    hipLaunchKernelGGL(afi_bn_act_amax_kernel, dim3(afi_cdiv(C, 8)), dim3(256), 0, st, mm, rows, C, mean, invstd, gamma, beta, slope, amax);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// BatchNorm backward finalize: dgamma += sum g*xhat ; dbeta += sum g ; stash the two sums for the apply pass
__global__ void afi_bn_bwd_finalize_kernel(const float* __restrict__ partial, int chunks, int C, float gscale,
                                           float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ sums) {
    int c; float s0, s1;
    if (!afi_chunk_sums(partial, chunks, C, true, c, s0, s1)) return;
    sums[c] = s0; sums[C + c] = s1;
    if (dbeta) dbeta[c] += gscale * s0;
    if (dgamma) dgamma[c] += gscale * s1;
}
// dx = gamma * invstd * (g - sum_g/P - xhat * sum_gx/P)   (in place on g allowed)
// MASK: g is the gradient w.r.t. the activation; it is first multiplied by the LeakyReLU' mask recomputed from x (see MODE 3 above)
template <bool MASK, bool AMAX = false>
__global__ void afi_bn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ dx,
                                        const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                        const float* __restrict__ sums, long long P, int C, const float* __restrict__ beta, float slope, float* amax,
                                        long long Pn) {
    // Pn: the number of rows the two sums were taken over (= P, or the batch of ALL ranks when the caller all-reduced them: SyncBatchNorm)
    const long long total4 = P * C / 4;
    const int C4 = C / 4;
    const float inv_n = 1.f / (float)Pn;
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float am = 0.f;
    auto apply = [inv_n, slope, &am](f32x4 gv, f32x4 xv, f32x4 mu, f32x4 is, f32x4 ga, f32x4 be, f32x4 sg, f32x4 sgx) {
        if (MASK) {
            const f32x4 z = afi_bn_affine(xv, mu, is, ga, be);
#pragma unroll
            for (int j = 0; j < 4; ++j) gv[j] = z[j] > 0.f ? gv[j] : gv[j] * slope;
        }
        const f32x4 xh = (xv - mu) * is;
        const f32x4 r = ga * is * (gv - sg * inv_n - xh * (sgx * inv_n));
        if (AMAX) am = afi_ew_amax4(am, r);
        return r;
    };
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if (stride % C4 == 0) {          // one channel group per thread: parameters hoisted, two element pairs in flight (see bn_apply)
        const int c = (int)(i % C4) * 4;
        const f32x4 mu = *(const f32x4*)(mean + c), is = *(const f32x4*)(invstd + c), ga = *(const f32x4*)(gamma + c);
        const f32x4 be = MASK ? *(const f32x4*)(beta + c) : zero;
        const f32x4 sg = *(const f32x4*)(sums + c), sgx = *(const f32x4*)(sums + C + c);
        for (; i + stride < total4; i += 2 * stride) {
            const f32x4 g0 = __builtin_nontemporal_load((const f32x4*)(g + i * 4)), x0 = __builtin_nontemporal_load((const f32x4*)(x + i * 4));
            const f32x4 g1 = __builtin_nontemporal_load((const f32x4*)(g + (i + stride) * 4)), x1 = __builtin_nontemporal_load((const f32x4*)(x + (i + stride) * 4));
            *(f32x4*)(dx + i * 4) = apply(g0, x0, mu, is, ga, be, sg, sgx);
            *(f32x4*)(dx + (i + stride) * 4) = apply(g1, x1, mu, is, ga, be, sg, sgx);
        }
        for (; i < total4; i += stride)
            *(f32x4*)(dx + i * 4) = apply(*(const f32x4*)(g + i * 4), *(const f32x4*)(x + i * 4), mu, is, ga, be, sg, sgx);
        if (AMAX) afi_ew_amax_publish(am, amax);
        return;
    }
    for (; i < total4; i += stride) {
        const int c = (int)(i % C4) * 4;
        *(f32x4*)(dx + i * 4) = apply(*(const f32x4*)(g + i * 4), *(const f32x4*)(x + i * 4), *(const f32x4*)(mean + c), *(const f32x4*)(invstd + c),
                                      *(const f32x4*)(gamma + c), MASK ? *(const f32x4*)(beta + c) : zero, *(const f32x4*)(sums + c), *(const f32x4*)(sums + C + c));
    }
    if (AMAX) afi_ew_amax_publish(am, amax);
}
// bias gradient finalize: db += alpha * sum
__global__ void afi_colsum_finalize_kernel(const float* __restrict__ partial, int chunks, int C, float alpha, float* __restrict__ db) {
    int c; float s0, s1;
    if (!afi_chunk_sums(partial, chunks, C, false, c, s0, s1)) return;
    db[c] += alpha * s0;
}

static void afi_red_geometry(long long P, int& chunks, int& rows_per_chunk) {
    long long want = (P + 63) / 64;
    if (want > AFI_RED_MAX_CHUNKS) want = AFI_RED_MAX_CHUNKS;
    if (want < 1) want = 1;
    rows_per_chunk = (int)((P + want - 1) / want);
    chunks = (int)((P + rows_per_chunk - 1) / rows_per_chunk);
}
extern "C" long long afi_reduce_scratch_floats(int C) { return (long long)AFI_RED_MAX_CHUNKS * 4 * C + 2 * C; }   // partials [chunks][2][C] as fp64 (statistics) or fp32, + [2][C] sums

static unsigned afi_ew_grid(long long work_items) {
    long long g = (work_items + 255) / 256;
    if (g > 2048) g = 2048;     // 256 CUs x 8 blocks, grid-stride the rest
    if (g < 1) g = 1;
    return (unsigned)g;
}

int afi_launch_bn_stats(const float* x, long long P, int C, float* mean, float* invstd, float* var_out,
                        float* running_mean, float* running_var, float* scratch, hipStream_t st, long long* num_batches_tracked, float eps,
                        float momentum, bool fp64) {
    if (eps < 0.f) eps = AFI_BN_EPS;                        // negative = torch's BatchNorm2d defaults (the discriminator's norms)
    if (momentum < 0.f) momentum = AFI_BN_MOMENTUM;
    if (P <= 0 || C <= 0 || (C & 3)) return AFI_ERR_BAD_ARG;
    int chunks, rpc; afi_red_geometry(P, chunks, rpc);
    if (fp64 && (((uintptr_t)scratch) & 7) == 0) {         // (fp64 = false: the fp32 one-pass form, AFI_OPT_BN_STATS_FP64 = 0)
        hipLaunchKernelGGL(afi_bn_stats_partial_kernel, dim3(afi_cdiv(C, 128), chunks), dim3(256), 0, st, x, P, C, (long long)C, rpc, (double*)scratch);
        hipLaunchKernelGGL(afi_bn_stats_finalize64_kernel<AFI_FIN_CH>, dim3(afi_cdiv(C, AFI_FIN_CH)), dim3(256), 0, st, (const double*)scratch, chunks, x, P, C, mean,
                           invstd, var_out, running_mean, running_var, num_batches_tracked, eps, momentum);
        return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
    }
    hipLaunchKernelGGL((afi_colred_partial_kernel<0>), dim3(afi_cdiv(C, 128), chunks), dim3(256), 0, st, x, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, P, C, (long long)C, rpc, scratch, (const float*)nullptr, (const float*)nullptr, 1.f);
    hipLaunchKernelGGL(afi_bn_stats_finalize_kernel, dim3(afi_cdiv(C, AFI_FIN_CH)), dim3(256), 0, st, scratch, chunks, x, P, C, mean, invstd,
                       var_out, running_mean, running_var, num_batches_tracked, eps, momentum);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
// statistics from fp64 partial rows [rows][2][C] (sum, sum of squares; unshifted) that another kernel accumulated -- the Winograd output
// transforms in their STATS form -- summed in row order; same outputs and running-buffer updates as afi_launch_bn_stats
int afi_launch_bn_stats_from_partials(const double* partial, int rows, long long P, int C, float* mean, float* invstd, float* var_out, float* running_mean,
                                      float* running_var, hipStream_t st, long long* num_batches_tracked, float eps, float momentum) {
    if (eps < 0.f) eps = AFI_BN_EPS;
    if (momentum < 0.f) momentum = AFI_BN_MOMENTUM;
    if (!partial || rows <= 0 || P <= 0 || C <= 0 || (C & 3)) return AFI_ERR_BAD_ARG;
    hipLaunchKernelGGL(afi_bn_stats_finalize64_kernel<8>, dim3(afi_cdiv(C, 8)), dim3(256), 0, st, partial, rows, (const float*)nullptr, P, C, mean, invstd,
                       var_out, running_mean, running_var, num_batches_tracked, eps, momentum);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
// amax (optional): raised to the largest magnitude of y (zero-filled by the caller)
int afi_launch_bn_apply_lrelu(const float* x, float* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                              long long P, int C, hipStream_t st, float slope, float* amax) {
    if (P <= 0 || C <= 0 || (C & 3)) return AFI_ERR_BAD_ARG;
    if (amax) hipLaunchKernelGGL(afi_bn_apply_lrelu_kernel<true>, dim3(afi_ew_grid(P * C / 4)), dim3(256), 0, st, x, y, mean, invstd, gamma, beta, P, C, slope, amax);
    else hipLaunchKernelGGL(afi_bn_apply_lrelu_kernel<false>, dim3(afi_ew_grid(P * C / 4)), dim3(256), 0, st, x, y, mean, invstd, gamma, beta, P, C, slope, amax);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
// mask_beta != nullptr: g is the gradient w.r.t. lrelu_slope(BN(x)) and the LeakyReLU' mask is recomputed from x (mask_beta = the norm's beta)
// amax (optional): raised to the largest magnitude of dx (zero-filled by the caller)
int afi_launch_bn_bwd(const float* g, const float* x, float* dx, const float* mean, const float* invstd, const float* gamma,
                      float* dgamma, float* dbeta, float gscale, long long P, int C, float* scratch, hipStream_t st, const float* mask_beta,
                      float slope, float* amax) {
    if (P <= 0 || C <= 0 || (C & 3)) return AFI_ERR_BAD_ARG;
    int chunks, rpc; afi_red_geometry(P, chunks, rpc);
    float* sums = scratch + (long long)AFI_RED_MAX_CHUNKS * 4 * C;
    const dim3 rgrid(afi_cdiv(C, 128), chunks), agrid(afi_ew_grid(P * C / 4));
    if (mask_beta) hipLaunchKernelGGL((afi_colred_partial_kernel<3>), rgrid, dim3(256), 0, st, x, g, mean, invstd, P, C, (long long)C, rpc, scratch, gamma, mask_beta, slope);
    else hipLaunchKernelGGL((afi_colred_partial_kernel<1>), rgrid, dim3(256), 0, st, x, g, mean, invstd, P, C, (long long)C, rpc, scratch, (const float*)nullptr, (const float*)nullptr, 1.f);
    hipLaunchKernelGGL(afi_bn_bwd_finalize_kernel, dim3(afi_cdiv(C, AFI_FIN_CH)), dim3(256), 0, st, scratch, chunks, C, gscale, dgamma, dbeta, sums);
    if (mask_beta && amax) hipLaunchKernelGGL((afi_bn_bwd_apply_kernel<true, true>), agrid, dim3(256), 0, st, g, x, dx, mean, invstd, gamma, sums, P, C, mask_beta, slope, amax, P);
    else if (mask_beta) hipLaunchKernelGGL((afi_bn_bwd_apply_kernel<true>), agrid, dim3(256), 0, st, g, x, dx, mean, invstd, gamma, sums, P, C, mask_beta, slope, amax, P);
    else if (amax) hipLaunchKernelGGL((afi_bn_bwd_apply_kernel<false, true>), agrid, dim3(256), 0, st, g, x, dx, mean, invstd, gamma, sums, P, C, (const float*)nullptr, 1.f, amax, P);
    else hipLaunchKernelGGL((afi_bn_bwd_apply_kernel<false>), agrid, dim3(256), 0, st, g, x, dx, mean, invstd, gamma, sums, P, C, (const float*)nullptr, 1.f, amax, P);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
// ... with the two sums already accumulated, as fp64 rows [rows][2][C], by the kernel that WROTE g (the data gradient's Winograd output
// transform: AfiPixGemm::bstats): the finalizer adds the rows in a fixed order, the apply pass is afi_launch_bn_bwd's
__global__ __launch_bounds__(256) void afi_bn_bwd_finalize64_kernel(const double* __restrict__ partial, int rows, int C, float gscale, float* __restrict__ dgamma,
                                                                    float* __restrict__ dbeta, float* __restrict__ sums) {
    __shared__ double red[2][32][8];
    const int cl = threadIdx.x & 7, ln = threadIdx.x >> 3;  // eight channels per block, 32 lanes per channel over the rows
    const int c = blockIdx.x * 8 + cl;
    double a0 = 0.0, a1 = 0.0;
    if (c < C)
        for (int i = ln; i < rows; i += 32) { a0 += partial[(long long)i * 2 * C + c]; a1 += partial[(long long)i * 2 * C + C + c]; }
    red[0][ln][cl] = a0; red[1][ln][cl] = a1;
    __syncthreads();
    if (ln != 0 || c >= C) return;
    double s0 = 0.0, s1 = 0.0;
    for (int j = 0; j < 32; ++j) { s0 += red[0][j][cl]; s1 += red[1][j][cl]; }
    sums[c] = (float)s0; sums[C + c] = (float)s1;
    if (dbeta) dbeta[c] += gscale * (float)s0;
    if (dgamma) dgamma[c] += gscale * (float)s1;
}
int afi_launch_bn_bwd_from_partials(const double* partial, int rows, const float* g, const float* x, float* dx, const float* mean, const float* invstd,
                                    const float* gamma, float* dgamma, float* dbeta, float gscale, long long P, int C, float* scratch, hipStream_t st,
                                    const float* mask_beta, float slope, float* amax) {
    if (!partial || rows <= 0 || P <= 0 || C <= 0 || (C & 3) || !scratch || !mask_beta) return AFI_ERR_BAD_ARG;
    float* sums2C = scratch + (long long)AFI_RED_MAX_CHUNKS * 4 * C;         // (where afi_launch_bn_bwd keeps them: afi_reduce_scratch_floats)
    hipLaunchKernelGGL(afi_bn_bwd_finalize64_kernel, dim3(afi_cdiv(C, 8)), dim3(256), 0, st, partial, rows, C, gscale, dgamma, dbeta, sums2C);
    const dim3 agrid(afi_ew_grid(P * C / 4));
    if (amax) hipLaunchKernelGGL((afi_bn_bwd_apply_kernel<true, true>), agrid, dim3(256), 0, st, g, x, dx, mean, invstd, gamma, sums2C, P, C, mask_beta, slope, amax, P);
    else hipLaunchKernelGGL((afi_bn_bwd_apply_kernel<true>), agrid, dim3(256), 0, st, g, x, dx, mean, invstd, gamma, sums2C, P, C, mask_beta, slope, amax, P);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
// The two halves of afi_launch_bn_bwd for a caller that exchanges the sums between ranks in between (SyncBatchNorm, bifpn_sr.py:210):
// sums2C[0..C) = sum_rows g, sums2C[C..2C) = sum_rows g * xhat (and dbeta / dgamma += them: those stay per-rank, as torch's SyncBatchNorm keeps them);
// then dx = gamma * invstd * (g - sums[0] / Pn - xhat * sums[1] / Pn) with the caller's (all-reduced) sums over Pn rows in all.
int afi_launch_bn_bwd_sums(const float* g, const float* x, const float* mean, const float* invstd, float* dgamma, float* dbeta, float* sums2C, long long P, int C,
                           float* scratch, hipStream_t st) {
    if (P <= 0 || C <= 0 || (C & 3) || !sums2C) return AFI_ERR_BAD_ARG;
    int chunks, rpc; afi_red_geometry(P, chunks, rpc);
    hipLaunchKernelGGL((afi_colred_partial_kernel<1>), dim3(afi_cdiv(C, 128), chunks), dim3(256), 0, st, x, g, mean, invstd, P, C, (long long)C, rpc, scratch,
                       (const float*)nullptr, (const float*)nullptr, 1.f);
    hipLaunchKernelGGL(afi_bn_bwd_finalize_kernel, dim3(afi_cdiv(C, AFI_FIN_CH)), dim3(256), 0, st, scratch, chunks, C, 1.f, dgamma, dbeta, sums2C);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
int afi_launch_bn_bwd_apply(const float* g, const float* x, float* dx, const float* mean, const float* invstd, const float* gamma, const float* sums2C, long long P,
                            long long P_total, int C, hipStream_t st) {
    if (P <= 0 || P_total < P || C <= 0 || (C & 3) || !sums2C) return AFI_ERR_BAD_ARG;
    hipLaunchKernelGGL((afi_bn_bwd_apply_kernel<false>), dim3(afi_ew_grid(P * C / 4)), dim3(256), 0, st, g, x, dx, mean, invstd, gamma, sums2C, P, C, (const float*)nullptr, 1.f,
                       (float*)nullptr, P_total);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
int afi_launch_colsum_accum(const float* g, long long P, int C, long long ld, float alpha, float* db, float* scratch, hipStream_t st) {
    if (P <= 0 || C <= 0 || (C & 3)) return AFI_ERR_BAD_ARG;
    int chunks, rpc; afi_red_geometry(P, chunks, rpc);
    hipLaunchKernelGGL((afi_colred_partial_kernel<2>), dim3(afi_cdiv(C, 128), chunks), dim3(256), 0, st, (const float*)nullptr, g,
                       (const float*)nullptr, (const float*)nullptr, P, C, ld, rpc, scratch, (const float*)nullptr, (const float*)nullptr, 1.f);
    hipLaunchKernelGGL(afi_colsum_finalize_kernel, dim3(afi_cdiv(C, AFI_FIN_CH)), dim3(256), 0, st, scratch, chunks, C, alpha, db);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// ---------------------------------------------------------------- last discriminator conv (1024 -> 1) as a stencil
// The 3x3 conv with ONE output channel is computed as  D9[q][t] = <X[q][:], W[t][:]>  (a 1x1 GEMM with 9 columns,
// run on the MFMA kernel so X is read exactly once) followed by  logit[p] = b + sum_t D9[p + tap_t][t].
__global__ void afi_stencil9_sum_kernel(const float* __restrict__ d9, int ld, const float* __restrict__ bias, float* __restrict__ out,
                                        int N, int H, int W) {
    const long long total = (long long)N * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int x = (int)(i % W); long long r = i / W; int y = (int)(r % H); int n = (int)(r / H);
        float s = bias ? bias[0] : 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) s += d9[(((long long)n * H + yy) * W + xx) * ld + t];
        }
        out[i] = s;
    }
}
// dD9[q][t] = dlogit[q - tap_t] (0 outside); columns 9..ld-1 are written as zeros
__global__ void afi_stencil9_scatter_kernel(const float* __restrict__ dlogit, float* __restrict__ dd9, int ld, int N, int H, int W) {
    const long long total = (long long)N * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int x = (int)(i % W); long long r = i / W; int y = (int)(r % H); int n = (int)(r / H);
        float* dst = dd9 + i * ld;
        for (int t = 0; t < ld; ++t) {
            float v = 0.f;
            if (t < 9) {
                int yy = y - (t / 3 - 1), xx = x - (t % 3 - 1);
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) v = dlogit[((long long)n * H + yy) * W + xx];
            }
            dst[t] = v;
        }
    }
}
int afi_launch_stencil9_sum(const float* d9, int ld, const float* bias, float* out, int N, int H, int W, hipStream_t st) {
    hipLaunchKernelGGL(afi_stencil9_sum_kernel, dim3(afi_ew_grid((long long)N * H * W)), dim3(256), 0, st, d9, ld, bias, out, N, H, W);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
int afi_launch_stencil9_scatter(const float* dlogit, float* dd9, int ld, int N, int H, int W, hipStream_t st) {
    hipLaunchKernelGGL(afi_stencil9_scatter_kernel, dim3(afi_ew_grid((long long)N * H * W)), dim3(256), 0, st, dlogit, dd9, ld, N, H, W);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// ---------------------------------------------------------------- the discriminator's tail without its two largest tensors
// Block 2's BatchNorm apply + LeakyReLU, the last conv (F3 -> 1) and their backward, with neither the activation y2 nor the gradient with
// respect to it ever written (AFI_OPT_D_FUSE_TAIL; at 2 x 200 x 336 x 1024 each is 550 MB):
//   forward   D9[q][t] = <lrelu(affine(c2[q][:])), w3[t][:]>      one read of c2 (the apply pass read it, wrote y2, and the GEMM read y2)
//   backward  g[q][c] = sum_t dD9[q][t] * w3[t][c] is nine multiply-adds per element from 64 bytes of dD9 per pixel, so both BatchNorm-backward
//             passes GENERATE it instead of reading it: the sums pass reads c2 only (and takes the last conv's weight gradient
//             dw3[t][c] = sum_q dD9[q][t] * y2[q][c] along, y2 being recomputed there for the mask anyway), the apply pass reads c2 and writes
//             d(conv output).  3.85 GB -> 1.65 GB of traffic per backward at that size, 1.65 -> 0.55 per forward.
// One wave owns one pixel row x 256 channels at a time (lane = four channels, 1 KB coalesced), so a row's nine dD9 values are wave-uniform
// (scalar loads); four waves of a block take the C / 256 channel groups of a row (C <= 1024, C % 16 == 0) or four rows.  The affine and the
// mask are afi_bn.h's (the decisions of the forward, bit for bit).
#define AFI_TAIL_MAX_C 1024
typedef float afi_f32x2 __attribute__((ext_vector_type(2)));
#define AFI_TAIL_MAX_CHUNKS 512                             /* backward sums: partial rows [chunks][11][C] */
extern "C" long long afi_disc_tail_scratch_floats(int C) { return (long long)AFI_TAIL_MAX_CHUNKS * 11 * C + 2 * C; }
// the gradient with respect to the activation, from the row's nine dD9 values: ONE definition for the two passes that generate it
__device__ __forceinline__ f32x4 afi_tail_g(const float d[9], const f32x4 w[9]) {
    f32x4 g = w[0] * d[0];
#pragma unroll
    for (int t = 1; t < 9; ++t) g = __builtin_elementwise_fma(w[t], f32x4{d[t], d[t], d[t], d[t]}, g);
    return g;
}
struct AfiTailLane { int grp, rl, RL, c; bool cok; };
__device__ __forceinline__ AfiTailLane afi_tail_lane(int C, int wave, int lane) {
    const int G = C > 512 ? 4 : (C > 256 ? 2 : 1);          // waves per pixel row
    AfiTailLane t;
    t.grp = wave % G; t.rl = wave / G; t.RL = 4 / G; t.c = t.grp * 256 + lane * 4; t.cok = t.c < C;
    return t;
}
// The forward product on the fp32 matrix cores (C % 16 == 0): D9^T[t][row] = sum_c W[t][c] * Y[c][row] as v_mfma_f32_16x16x4_f32 with the
// taps as the 16 rows of A (nine live), sixteen pixel rows as the columns of B and four channels per instruction -- the reduction over the
// channels happens inside the MFMA accumulators, so nothing crosses lanes and the VALU only evaluates the affine (a first form with one
// wave per pixel row and a DPP reduction of the nine dot products per row ran at 2.3 TB/s: 240 us at 2 x 200 x 336 x 1024).  Lane (j = lane & 15,
// k = lane >> 4) loads the 16 bytes of pixel row j at channels 16 s + 4 k .. + 3 (64 contiguous bytes per row and instruction) and feeds
// them to four MFMAs; its A operands (tap j, the same four channels) and its four parameter vectors come from LDS (9 x C weights +
// 4 x C parameters: 52 KB at C = 1024).  A wave owns sixteen rows, a block of eight waves 128, and walks its share of the 128-row tiles.
#ifndef AFI_TAIL_UNROLL
#define AFI_TAIL_UNROLL 4
#endif
#ifndef AFI_TAIL_ABLATE
#define AFI_TAIL_ABLATE 0                                   /* tools/micro/tail_probe.py builds only: 1 no MFMAs, 2 no affine, 4 no weight reads (wrong results) */
#endif
#ifndef AFI_TAIL_WAVES
#define AFI_TAIL_WAVES 4                                     /* waves per block: sixteen pixel rows each, one set of LDS images (eight: 201 against 158 us) */
#endif
// lrelu(z) as max(z, z * slope): the value of afi_bn_lrelu's select for every z (0 < slope < 1: z * slope < z exactly when z > 0; both forms
// return -0 for -0 and NaN for NaN) in half the instructions
__device__ __forceinline__ f32x4 afi_tail_act(f32x4 v, f32x4 mu, f32x4 is, f32x4 ga, f32x4 be, float slope) {
    const f32x4 z = afi_bn_affine(v, mu, is, ga, be);
    const f32x4 zs = z * slope;
    return f32x4{fmaxf(z[0], zs[0]), fmaxf(z[1], zs[1]), fmaxf(z[2], zs[2]), fmaxf(z[3], zs[3])};
}
template <bool BN>
__global__ __launch_bounds__(64 * AFI_TAIL_WAVES) void afi_disc_tail_fwd_mfma_kernel(const float* __restrict__ x, const AfiBnLoad bn, float slope, const float* __restrict__ w3,
                                                                                     float* __restrict__ d9, long long P, int C) {
    extern __shared__ __attribute__((aligned(16))) float tail_lds[];
    const int ldw = C + 4;                                  // (row stride of the weights: the sixteen taps of a quarter wave on distinct banks)
    float* Ws = tail_lds;                                   // [9][C + 4], then four zeros (what the lanes of taps 9 .. 15 read)
    float* Zs = tail_lds + 9 * ldw;
    float* Ps = Zs + 4;                                     // [4][C]: mean, invstd, gamma, beta
    for (int e = threadIdx.x; e < 9 * (C >> 2); e += 64 * AFI_TAIL_WAVES) {
        const int t = e / (C >> 2), c4 = e - t * (C >> 2);
        *(f32x4*)(Ws + t * ldw + 4 * c4) = *(const f32x4*)(w3 + (long long)t * C + 4 * c4);
    }
    if (threadIdx.x == 0) *(f32x4*)Zs = f32x4{0.f, 0.f, 0.f, 0.f};
    if (BN)
        for (int e = threadIdx.x; e < (C >> 2); e += 64 * AFI_TAIL_WAVES) {
            *(f32x4*)(Ps + 4 * e) = *(const f32x4*)(bn.mean + 4 * e); *(f32x4*)(Ps + C + 4 * e) = *(const f32x4*)(bn.invstd + 4 * e);
            *(f32x4*)(Ps + 2 * C + 4 * e) = *(const f32x4*)(bn.gamma + 4 * e); *(f32x4*)(Ps + 3 * C + 4 * e) = *(const f32x4*)(bn.beta + 4 * e);
        }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, k = lane >> 4;
    const bool live = j < 9;
    const float* wr = live ? Ws + j * ldw + 4 * k : Zs;     // this lane's A operands: + wstep per 16-channel step (0 for the zero lanes)
    const int wstep = live ? 16 : 0;
    const float* pr = Ps + 4 * k;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const int nS = C >> 4;
    constexpr int TR = 16 * AFI_TAIL_WAVES;                // pixel rows per tile
    const long long ntile = (P + TR - 1) / TR;
    for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {       // (the LDS images serve every tile of the block)
        const long long row = tile * TR + wave * 16 + j;
        const bool rok = row < P;
        const float* xr = x + (rok ? row : P - 1) * C + 4 * k;                 // (rows past the end re-read the last one and store nothing)
        f32x4 acc = zero;
        // a batch of AFI_TAIL_UNROLL 16-channel steps: this batch's LDS operands and the NEXT batch's rows are requested first, then the batch is
        // multiplied -- the MFMAs wait for neither (a step that read its weights just before its MFMAs ran at 157 us, without the MFMAs 113,
        // without the weight reads 121: the LDS round trip in front of every group of four MFMAs was what it paid)
        auto step = [&](f32x4 y, const f32x4 wv, const f32x4 p0, const f32x4 p1, const f32x4 p2, const f32x4 p3) {
            if (BN && !(AFI_TAIL_ABLATE & 2)) y = afi_tail_act(y, p0, p1, p2, p3, slope);
            if (AFI_TAIL_ABLATE & 1) { acc += wv * y; return; }
#pragma unroll
            for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[m], y[m], acc, 0, 0, 0);
        };
        auto lds_w = [&](int sidx) { return (AFI_TAIL_ABLATE & 4) ? f32x4{1.f, 1.f, 1.f, 1.f} : *(const f32x4*)(wr + sidx * wstep); };
        int s0 = 0;
        f32x4 v[AFI_TAIL_UNROLL];
        if (AFI_TAIL_UNROLL <= nS) {
#pragma unroll
            for (int u = 0; u < AFI_TAIL_UNROLL; ++u) v[u] = __builtin_nontemporal_load((const f32x4*)(xr + 16 * u));
        }
        for (; s0 + AFI_TAIL_UNROLL <= nS; s0 += AFI_TAIL_UNROLL) {
            f32x4 wv[AFI_TAIL_UNROLL], pp[AFI_TAIL_UNROLL][4], vn[AFI_TAIL_UNROLL];
#pragma unroll
            for (int u = 0; u < AFI_TAIL_UNROLL; ++u) {
                wv[u] = lds_w(s0 + u);
                if (BN) {
                    const float* q = pr + 16 * (s0 + u);
                    pp[u][0] = *(const f32x4*)q; pp[u][1] = *(const f32x4*)(q + C); pp[u][2] = *(const f32x4*)(q + 2 * C); pp[u][3] = *(const f32x4*)(q + 3 * C);
                }
            }
            const bool more = s0 + 2 * AFI_TAIL_UNROLL <= nS;                  // (uniform)
            if (more) {
#pragma unroll
                for (int u = 0; u < AFI_TAIL_UNROLL; ++u) vn[u] = __builtin_nontemporal_load((const f32x4*)(xr + 16 * (s0 + AFI_TAIL_UNROLL + u)));
            }
#pragma unroll
            for (int u = 0; u < AFI_TAIL_UNROLL; ++u) step(v[u], wv[u], pp[u][0], pp[u][1], pp[u][2], pp[u][3]);
            if (more) {
#pragma unroll
                for (int u = 0; u < AFI_TAIL_UNROLL; ++u) v[u] = vn[u];
            }
        }
        for (; s0 < nS; ++s0) {
            const float* q = pr + 16 * s0;
            const f32x4 z4 = zero;
            step(__builtin_nontemporal_load((const f32x4*)(xr + 16 * s0)), lds_w(s0), BN ? *(const f32x4*)q : z4, BN ? *(const f32x4*)(q + C) : z4,
                 BN ? *(const f32x4*)(q + 2 * C) : z4, BN ? *(const f32x4*)(q + 3 * C) : z4);
        }
        // acc[r] = D9[row j][tap 4 k + r]: one 16-byte store per lane (taps 9 .. 15: zeros)
        if (rok) *(f32x4*)(d9 + row * 16 + 4 * k) = acc;
    }
}
// partial[chunk][q][C]: q = 0 sum g m, 1 sum g m xhat, 2 + t sum dD9[.][t] y   (g m: the gradient through the LeakyReLU mask m)
template <bool WQ>
__global__ __launch_bounds__(256) void afi_disc_tail_bwd_sums_kernel(const float* __restrict__ x, const float* __restrict__ dd9, const AfiBnLoad bn, float slope,
                                                                     const float* __restrict__ w3, long long P, int C, int rows_per_chunk,
                                                                     float* __restrict__ partial) {
    __shared__ f32x4 red[3][4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.x * 256 + lane * 4;
    const bool cok = c < C;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 w[9], mu = zero, is = zero, ga = zero, be = zero, s0 = zero, s1 = zero, wq[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) { w[t] = cok ? *(const f32x4*)(w3 + (long long)t * C + c) : zero; wq[t] = zero; }
    if (cok) { mu = *(const f32x4*)(bn.mean + c); is = *(const f32x4*)(bn.invstd + c); ga = *(const f32x4*)(bn.gamma + c); be = *(const f32x4*)(bn.beta + c); }
    const long long r0 = (long long)blockIdx.y * rows_per_chunk;
    const long long r1 = (r0 + rows_per_chunk < P) ? r0 + rows_per_chunk : P;
    for (long long r = r0 + wave; r < r1; r += 8) {         // two rows in flight per wave
        const bool two = r + 4 < r1;                        // (uniform)
        const f32x4 xa = cok ? __builtin_nontemporal_load((const f32x4*)(x + r * C + c)) : zero;
        const f32x4 xb = (cok && two) ? __builtin_nontemporal_load((const f32x4*)(x + (r + 4) * C + c)) : zero;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u && !two) break;
            const f32x4 xv = u ? xb : xa;
            const float* dr = dd9 + (r + 4 * u) * 16;
            float d[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) d[t] = dr[t];
            f32x4 gv = afi_tail_g(d, w);
            const f32x4 z = afi_bn_affine(xv, mu, is, ga, be);
#pragma unroll
            for (int j = 0; j < 4; ++j) gv[j] = z[j] > 0.f ? gv[j] : gv[j] * slope;
            const f32x4 xh = (xv - mu) * is;
            s0 += gv; s1 += gv * xh;
            if (WQ) {
                f32x4 y = z;
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = z[j] > 0.f ? z[j] : z[j] * slope;
#pragma unroll
                for (int t = 0; t < 9; ++t) wq[t] = __builtin_elementwise_fma(y, f32x4{d[t], d[t], d[t], d[t]}, wq[t]);
            }
        }
    }
    // the four waves' sums meet in LDS, four quantities at a time, in a fixed order
    float* dst = partial + (long long)blockIdx.y * 11 * C;
    constexpr int NQ = WQ ? 11 : 2;
#pragma unroll
    for (int q0 = 0; q0 < NQ; q0 += 4) {
        if (q0) __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = q0 + k;
            if (q < NQ && wave > 0) red[wave - 1][k][lane] = q == 0 ? s0 : (q == 1 ? s1 : wq[q < 2 ? 0 : q - 2]);
        }
        __syncthreads();
        if (wave == 0 && cok) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int q = q0 + k;
                if (q >= NQ) continue;
                f32x4 v = q == 0 ? s0 : (q == 1 ? s1 : wq[q < 2 ? 0 : q - 2]);
                v += red[0][k][lane]; v += red[1][k][lane]; v += red[2][k][lane];
                *(f32x4*)(dst + (long long)q * C + c) = v;
            }
        }
    }
}
// grid (C / 32, nq): quantity q of 32 channels summed over the chunks (eight lanes per channel, fixed order)
__global__ __launch_bounds__(256) void afi_disc_tail_finalize_kernel(const float* __restrict__ partial, int chunks, int C, float* __restrict__ sums,
                                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dw3) {
    __shared__ float red[8][32];
    const int cl = threadIdx.x & 31, ln = threadIdx.x >> 5, q = blockIdx.y;
    const int c = blockIdx.x * 32 + cl;
    float a = 0.f;
    if (c < C)
        for (int i = ln; i < chunks; i += 8) a += partial[((long long)i * 11 + q) * C + c];
    red[ln][cl] = a;
    __syncthreads();
    if (ln != 0 || c >= C) return;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += red[j][cl];
    if (q == 0) { sums[c] = s; if (dbeta) dbeta[c] += s; }
    else if (q == 1) { sums[C + c] = s; if (dgamma) dgamma[c] += s; }
    else if (dw3) dw3[(long long)(q - 2) * C + c] += s;
}
// d(conv output) = gamma invstd (g m - sum_gm / P - xhat sum_gmx / P): afi_bn_bwd_apply_kernel<true> with g generated
template <bool AMAX>
__global__ __launch_bounds__(256) void afi_disc_tail_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dd9, const AfiBnLoad bn, float slope,
                                                                      const float* __restrict__ w3, const float* __restrict__ sums, float* __restrict__ dx,
                                                                      long long P, int C, float* amax) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const AfiTailLane L = afi_tail_lane(C, wave, lane);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 w[9], mu = zero, is = zero, ga = zero, be = zero, sg = zero, sgx = zero;
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = L.cok ? *(const f32x4*)(w3 + (long long)t * C + L.c) : zero;
    if (L.cok) {
        mu = *(const f32x4*)(bn.mean + L.c); is = *(const f32x4*)(bn.invstd + L.c); ga = *(const f32x4*)(bn.gamma + L.c); be = *(const f32x4*)(bn.beta + L.c);
        sg = *(const f32x4*)(sums + L.c); sgx = *(const f32x4*)(sums + C + L.c);
    }
    const float inv_n = 1.f / (float)P;
    float am = 0.f;
    const long long step = (long long)gridDim.x * L.RL;
    for (long long r = (long long)blockIdx.x * L.RL + L.rl; r < P; r += 2 * step) {     // two rows in flight per wave
        const bool two = r + step < P;                      // (uniform)
        const f32x4 xa = L.cok ? __builtin_nontemporal_load((const f32x4*)(x + r * C + L.c)) : zero;
        const f32x4 xb = (L.cok && two) ? __builtin_nontemporal_load((const f32x4*)(x + (r + step) * C + L.c)) : zero;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u && !two) break;
            const f32x4 xv = u ? xb : xa;
            const long long rr = r + u * step;
            const float* dr = dd9 + rr * 16;
            float d[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) d[t] = dr[t];
            f32x4 gv = afi_tail_g(d, w);
            const f32x4 z = afi_bn_affine(xv, mu, is, ga, be);
#pragma unroll
            for (int j = 0; j < 4; ++j) gv[j] = z[j] > 0.f ? gv[j] : gv[j] * slope;
            const f32x4 xh = (xv - mu) * is;
            const f32x4 o = ga * is * (gv - sg * inv_n - xh * (sgx * inv_n));
            if (AMAX) am = afi_ew_amax4(am, o);
            if (L.cok) *(f32x4*)(dx + rr * C + L.c) = o;
        }
    }
    if (AMAX) afi_ew_amax_publish(am, amax);
}
static bool afi_tail_ok(long long P, int C) { return P > 0 && C > 0 && !(C & 15) && C <= AFI_TAIL_MAX_C; }
// bn == nullptr: x is the activation itself (nothing applied on load)
int afi_launch_disc_tail_fwd(const float* x, const AfiBnLoad* bn, float slope, const float* w3, float* d9, long long P, int C, hipStream_t st) {
    if (!x || !w3 || !d9 || !afi_tail_ok(P, C)) return AFI_ERR_BAD_ARG;
    const AfiBnLoad off{nullptr, nullptr, nullptr, nullptr};
    const size_t lds = sizeof(float) * (9 * (size_t)(C + 4) + 4 + 4 * (size_t)C);
    const long long tiles = (P + 16 * AFI_TAIL_WAVES - 1) / (16 * AFI_TAIL_WAVES);
    const long long per = (tiles + 767) / 768;              // 256 CUs x (up to) 3 blocks of LDS images: every block takes `per` tiles (or one fewer)
    const unsigned grid = (unsigned)((tiles + per - 1) / per);
    if (bn && bn->mean) hipLaunchKernelGGL(afi_disc_tail_fwd_mfma_kernel<true>, dim3(grid), dim3(64 * AFI_TAIL_WAVES), lds, st, x, *bn, slope, w3, d9, P, C);
    else hipLaunchKernelGGL(afi_disc_tail_fwd_mfma_kernel<false>, dim3(grid), dim3(64 * AFI_TAIL_WAVES), lds, st, x, off, slope, w3, d9, P, C);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
// x = the block's saved conv output, dd9 = the scattered logit gradients [P][16]; dx = d(conv output) (amax: raised to its largest magnitude);
// dgamma / dbeta / dw3 (+=, each optional); scratch: afi_disc_tail_scratch_floats(C)
int afi_launch_disc_tail_bwd(const float* x, const float* dd9, const AfiBnLoad bn, float slope, const float* w3, float* dx, float* dgamma, float* dbeta, float* dw3,
                             long long P, int C, float* scratch, float* amax, hipStream_t st) {
    if (!x || !dd9 || !w3 || !dx || !scratch || !bn.mean || !afi_tail_ok(P, C)) return AFI_ERR_BAD_ARG;
    long long want = (P + 31) / 32;
    if (want > AFI_TAIL_MAX_CHUNKS) want = AFI_TAIL_MAX_CHUNKS;
    const int rpc = (int)((P + want - 1) / want), chunks = (int)((P + rpc - 1) / rpc);
    float* sums = scratch + (long long)AFI_TAIL_MAX_CHUNKS * 11 * C;
    const dim3 sgrid(afi_cdiv(C, 256), chunks);
    if (dw3) hipLaunchKernelGGL(afi_disc_tail_bwd_sums_kernel<true>, sgrid, dim3(256), 0, st, x, dd9, bn, slope, w3, P, C, rpc, scratch);
    else hipLaunchKernelGGL(afi_disc_tail_bwd_sums_kernel<false>, sgrid, dim3(256), 0, st, x, dd9, bn, slope, w3, P, C, rpc, scratch);
    hipLaunchKernelGGL(afi_disc_tail_finalize_kernel, dim3(afi_cdiv(C, 32), dw3 ? 11 : 2), dim3(256), 0, st, scratch, chunks, C, sums, dgamma, dbeta, dw3);
    const int G = C > 512 ? 4 : (C > 256 ? 2 : 1);
    long long agrid = (P * G + 3) / 4;                      // one wave per (row, channel group), grid-stride beyond 256 CUs x 8 blocks
    if (agrid > 2048) agrid = 2048;
    if (amax) hipLaunchKernelGGL(afi_disc_tail_bwd_apply_kernel<true>, dim3((unsigned)agrid), dim3(256), 0, st, x, dd9, bn, slope, w3, sums, dx, P, C, amax);
    else hipLaunchKernelGGL(afi_disc_tail_bwd_apply_kernel<false>, dim3((unsigned)agrid), dim3(256), 0, st, x, dd9, bn, slope, w3, sums, dx, P, C, amax);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// ---------------------------------------------------------------- losses
// BCE-with-logits (mean) against a constant target t:  loss += lscale * mean(max(z,0) - z t + log1p(exp(-|z|)))
// dz = gscale * (sigmoid(z) - t) / n   (written only if dz != null)
__global__ __launch_bounds__(256) void afi_bce_logits_kernel(const float* __restrict__ z, long long n, float target, float lscale,
                                                             float* __restrict__ loss, float gscale, float* __restrict__ dz) {
    __shared__ float red[4];
    float s = 0.f;
    const float inv_n = 1.f / (float)n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = z[i];
        const float e = expf(-fabsf(v));
        s += fmaxf(v, 0.f) - v * target + log1pf(e);
        if (dz) {
            const float sig = v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
            dz[i] = gscale * (sig - target) * inv_n;
        }
    }
    s = afi_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0 && loss) atomicAdd(loss, lscale * inv_n * (red[0] + red[1] + red[2] + red[3]));
}
int afi_launch_bce_logits(const float* z, long long n, float target, float lscale, float* loss, float gscale, float* dz, hipStream_t st) {
    if (n <= 0) return AFI_ERR_BAD_ARG;
    long long g = (n + 255) / 256; if (g > 256) g = 256;
    hipLaunchKernelGGL(afi_bce_logits_kernel, dim3((unsigned)g), dim3(256), 0, st, z, n, target, lscale, loss, gscale, dz);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// L1 (mean) between two pixel-major views over the common crop [N, h, w, C]:
//   loss += lscale * mean|a - b| ;  da (dense [N, Ha, Wa, C], the full extent of a) = gscale*sign(a-b)/n inside the crop, 0 outside
__global__ __launch_bounds__(256) void afi_l1_kernel(AfiView a, AfiView b, int N, int h, int w, int C, int Ha, int Wa, float lscale,
                                                     float* __restrict__ loss, float gscale, float* __restrict__ da) {
    __shared__ float red[4];
    const int C4 = C / 4;
    const long long total = (long long)N * Ha * Wa * C4;   // iterate over the FULL extent of a so da gets its zeros
    const float inv_n = 1.f / ((float)N * (float)h * (float)w * (float)C);
    const float gs = gscale * inv_n;
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4; long long r = i / C4;
        const int x = (int)(r % Wa); r /= Wa; const int y = (int)(r % Ha); const int n = (int)(r / Ha);
        f32x4 g = {0, 0, 0, 0};
        if (y < h && x < w) {
            const f32x4 av = *(const f32x4*)(a.p + n * a.sN + y * a.sH + x * a.sW + c);
            const f32x4 bv = *(const f32x4*)(b.p + n * b.sN + y * b.sH + x * b.sW + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = av[j] - bv[j];
                s += fabsf(d);
                g[j] = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
            }
        }
        if (da) *(f32x4*)(da + i * 4) = g;
    }
    s = afi_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0 && loss) atomicAdd(loss, lscale * inv_n * (red[0] + red[1] + red[2] + red[3]));
}
int afi_launch_l1(AfiView a, AfiView b, int N, int h, int w, int C, int Ha, int Wa, float lscale, float* loss, float gscale, float* da,
                  hipStream_t st) {
    if (N <= 0 || h <= 0 || w <= 0 || C <= 0 || (C & 3) || h > Ha || w > Wa) return AFI_ERR_BAD_ARG;
    hipLaunchKernelGGL(afi_l1_kernel, dim3(afi_ew_grid((long long)N * Ha * Wa * C / 4)), dim3(256), 0, st, a, b, N, h, w, C, Ha, Wa, lscale,
                       loss, gscale, da);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// ---------------------------------------------------------------- bilinear x2 (standalone forward / backward)
// (index map and the transpose's element: afi_bilinear.h)
// out[N,2H,2W,C] (dense) = beta*out + bilinear2x(x[N,H,W,C] view)
__global__ void afi_bilinear2x_fwd_kernel(AfiView x, int N, int H, int W, int C, float beta, float* __restrict__ out) {
    const int C4 = C / 4;
    const long long total = (long long)N * 4 * H * W * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4; long long r = i / C4;
        const int xo = (int)(r % (2 * W)); r /= 2 * W; const int yo = (int)(r % (2 * H)); const int n = (int)(r / (2 * H));
        int y0, y1, x0, x1; float ly, lx;
        afi_bil_idx2(yo, H, y0, y1, ly); afi_bil_idx2(xo, W, x0, x1, lx);
        const float* b = x.p + n * x.sN + c;
        const f32x4 v00 = *(const f32x4*)(b + y0 * x.sH + x0 * x.sW), v01 = *(const f32x4*)(b + y0 * x.sH + x1 * x.sW);
        const f32x4 v10 = *(const f32x4*)(b + y1 * x.sH + x0 * x.sW), v11 = *(const f32x4*)(b + y1 * x.sH + x1 * x.sW);
        f32x4 v = (v00 * (1.f - lx) + v01 * lx) * (1.f - ly) + (v10 * (1.f - lx) + v11 * lx) * ly;
        if (beta != 0.f) v += beta * *(const f32x4*)(out + i * 4);
        *(f32x4*)(out + i * 4) = v;
    }
}
// dx[N,H,W,C] (dense) = beta*dx + bilinear2x^T(dout[N,2H,2W,C] dense)
__global__ void afi_bilinear2x_bwd_kernel(const float* __restrict__ dout, int N, int H, int W, int C, float beta, float* __restrict__ dx) {
    const long long total = (long long)N * H * W * (C / 4);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        f32x4 acc = afi_bilinear2x_bwd_elem(dout, H, W, C, i);
        if (beta != 0.f) acc += beta * *(const f32x4*)(dx + i * 4);
        *(f32x4*)(dx + i * 4) = acc;
    }
}
int afi_launch_bilinear2x_fwd(AfiView x, int N, int H, int W, int C, float beta, float* out, hipStream_t st) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return AFI_ERR_BAD_ARG;
    hipLaunchKernelGGL(afi_bilinear2x_fwd_kernel, dim3(afi_ew_grid((long long)N * H * W * C)), dim3(256), 0, st, x, N, H, W, C, beta, out);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
int afi_launch_bilinear2x_bwd(const float* dout, int N, int H, int W, int C, float beta, float* dx, hipStream_t st) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return AFI_ERR_BAD_ARG;
    hipLaunchKernelGGL(afi_bilinear2x_bwd_kernel, dim3(afi_ew_grid((long long)N * H * W * C / 4)), dim3(256), 0, st, dout, N, H, W, C, beta, dx);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// ---------------------------------------------------------------- multi-tensor SGD with momentum
// torch.optim.SGD semantics (detectron2 build_optimizer): d = g*gscale + wd*p ; buf = mom*buf + d ; p -= lr*buf
struct AfiSgdDesc { float* p; const float* g; float* m; long long n; float wd; float pad; };
__global__ void afi_sgd_kernel(const AfiSgdDesc* __restrict__ descs, float lr, float mom, float gscale) {
    const AfiSgdDesc d = descs[blockIdx.y];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += (long long)gridDim.x * blockDim.x) {
        const float pv = d.p[i];
        const float dd = d.g[i] * gscale + d.wd * pv;
        const float b = mom * d.m[i] + dd;
        d.m[i] = b;
        d.p[i] = pv - lr * b;
    }
}
int afi_launch_sgd(const void* descs_dev, int ntensors, long long max_n, float lr, float mom, float gscale, hipStream_t st) {
    if (ntensors <= 0) return AFI_ERR_BAD_ARG;
    long long gx = (max_n + 255) / 256; if (gx > 512) gx = 512; if (gx < 1) gx = 1;
    hipLaunchKernelGGL(afi_sgd_kernel, dim3((unsigned)gx, ntensors), dim3(256), 0, st, (const AfiSgdDesc*)descs_dev, lr, mom, gscale);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
// scale a flat buffer (used to average all-reduced gradients when the collective sums)
__global__ void afi_scale_kernel(float* __restrict__ p, long long n, float s) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] *= s;
}
int afi_launch_scale(float* p, long long n, float s, hipStream_t st) {
    hipLaunchKernelGGL(afi_scale_kernel, dim3(afi_ew_grid(n)), dim3(256), 0, st, p, n, s);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// out = scale * g * (act > 0): gradient through an (in-place) ReLU whose OUTPUT was kept (pafpn_sr.py:178)
__global__ void afi_relu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ act, float* __restrict__ out, long long n4, float s) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 gv = ((const f32x4*)g)[i], av = ((const f32x4*)act)[i];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = av[j] > 0.f ? s * gv[j] : 0.f;
        ((f32x4*)out)[i] = o;
    }
}
int afi_launch_relu_bwd(const float* g, const float* act, float* out, long long n, float s, hipStream_t st) {
    if (n <= 0 || (n & 3)) return AFI_ERR_BAD_ARG;
    hipLaunchKernelGGL(afi_relu_bwd_kernel, dim3(afi_ew_grid(n >> 2)), dim3(256), 0, st, g, act, out, n >> 2, s);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// ---------------------------------------------------------------- BiFPN inference pieces (bifpn_sr.py:569-733, bifpn_layers/wrappers.py)
// depthwise 3x3, stride 1, zero pad 1 ("static_same" of SeparableConv2d.depthwise, no bias): one thread = one pixel x 4 channels.
// w: [9][C] (tap-major repack of torch's [C][1][3][3]).  HBM-bound: x is read once from HBM (the 9 taps hit L2), out written once.
__global__ void afi_dwconv3x3_kernel(const AfiView x, int N, int H, int W, int C, const float* __restrict__ w, float* __restrict__ out) {
    const int C4 = C >> 2;
    const long long total = (long long)N * H * W * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4; long long r = i / C4;
        const int xx = (int)(r % W); r /= W; const int yy = (int)(r % H); const int n = (int)(r / H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int y2 = yy + t / 3 - 1, x2 = xx + t % 3 - 1;
            if ((unsigned)y2 < (unsigned)H && (unsigned)x2 < (unsigned)W)
                acc += *(const f32x4*)(x.p + (long long)n * x.sN + (long long)y2 * x.sH + (long long)x2 * x.sW + c) * *(const f32x4*)(w + t * C + c);
        }
        *(f32x4*)(out + i * 4) = acc;
    }
}
int afi_launch_dwconv3x3(AfiView x, int N, int H, int W, int C, const float* w, float* out, hipStream_t st) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return AFI_ERR_BAD_ARG;
    if (C & 3) return AFI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(afi_dwconv3x3_kernel, dim3(afi_ew_grid((long long)N * H * W * (C >> 2))), dim3(256), 0, st, x, N, H, W, C, w, out);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
// MaxPool2d(3, 2, "static_same"): F.pad(0 left/top, 1 right/bottom, ZEROS) then max_pool2d(3, 2): out[oy][ox] = max over rows
// 2oy..2oy+2, cols 2ox..2ox+2 where positions == H / == W count as 0.0 (the zero pad takes part in the max), Ho = (H-2)/2 + 1.
__global__ void afi_maxpool3s2_same_kernel(const AfiView x, int N, int H, int W, int C, int Ho, int Wo, float* __restrict__ out) {
    const int C4 = C >> 2;
    const long long total = (long long)N * Ho * Wo * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4; long long r = i / C4;
        const int ox = (int)(r % Wo); r /= Wo; const int oy = (int)(r % Ho); const int n = (int)(r / Ho);
        f32x4 m = {-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int y2 = 2 * oy + t / 3, x2 = 2 * ox + t % 3;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};                  // the pad row / column
            if (y2 < H && x2 < W) v = *(const f32x4*)(x.p + (long long)n * x.sN + (long long)y2 * x.sH + (long long)x2 * x.sW + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], v[j]);
        }
        *(f32x4*)(out + i * 4) = m;
    }
}
int afi_launch_maxpool3s2_same(AfiView x, int N, int H, int W, int C, float* out, hipStream_t st) {
    if (N <= 0 || H < 2 || W < 2 || C <= 0) return AFI_ERR_BAD_ARG;
    if (C & 3) return AFI_ERR_UNSUPPORTED;
    const int Ho = (H - 2) / 2 + 1, Wo = (W - 2) / 2 + 1;
    hipLaunchKernelGGL(afi_maxpool3s2_same_kernel, dim3(afi_ew_grid((long long)N * Ho * Wo * (C >> 2))), dim3(256), 0, st, x, N, H, W, C, Ho, Wo, out);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
// out = swish(w[0]*a + w[1]*b (+ w[2]*c)),  swish(v) = v * sigmoid(v)  (BiFPN "_attention" with the RAW fusion weights, then
// MemoryEfficientSwish: bifpn_sr.py:535-563, activations.py).  w: DEVICE pointer to the 2 or 3 weights (no host sync).
__global__ void afi_fuse_swish_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                      const float* __restrict__ w, float* __restrict__ out, long long n4) {
    const float w0 = w[0], w1 = w[1], w2 = c ? w[2] : 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 v = w0 * ((const f32x4*)a)[i] + w1 * ((const f32x4*)b)[i];
        if (c) v += w2 * ((const f32x4*)c)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] / (1.f + __expf(-v[j]));
        ((f32x4*)out)[i] = v;
    }
}
int afi_launch_fuse_swish(const float* a, const float* b, const float* c, const float* w, float* out, long long n, hipStream_t st) {
    if (n <= 0 || (n & 3) || !a || !b || !w || !out) return AFI_ERR_BAD_ARG;
    hipLaunchKernelGGL(afi_fuse_swish_kernel, dim3(afi_ew_grid(n >> 2)), dim3(256), 0, st, a, b, c, w, out, n >> 2);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// ---------------------------------------------------------------- small helpers
// *out += alpha * sum(v[0..n))
__global__ __launch_bounds__(1024) void afi_sum_accum_kernel(const float* __restrict__ v, long long n, float alpha, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    const long long n4 = ((((uintptr_t)v) & 15) == 0) ? n >> 2 : 0;           // 16-byte loads where the vector allows them (it comes from an allocator: always)
    for (long long i = threadIdx.x; i < n4; i += blockDim.x) { const f32x4 q = *(const f32x4*)(v + 4 * i); s += (q[0] + q[1]) + (q[2] + q[3]); }
    for (long long i = 4 * n4 + threadIdx.x; i < n; i += blockDim.x) s += v[i];
    s = afi_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];           // fixed order
        atomicAdd(out, alpha * t);
    }
}
int afi_launch_sum_accum(const float* v, long long n, float alpha, float* out, hipStream_t st) {
    // ONE block: its threads' partial sums meet in a fixed order, so the result does not depend on which of several blocks' atomics lands
    // first.  The vector is one logit gradient per pixel (134 K floats at most); on the large levels this launch sits ON the main stream
    // (no side stream above kSideStreamMaxPixels), so the block is 1024 threads of 16-byte loads: 33 dependent loads per thread instead of
    // the 525 of a 256-thread scalar walk (58 us -> a few us per launch at P2)
    hipLaunchKernelGGL(afi_sum_accum_kernel, dim3(1), dim3(1024), 0, st, v, n, alpha, out);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
__global__ void afi_inc_i64_kernel(long long* p) { if (threadIdx.x == 0 && blockIdx.x == 0) *p += 1; }
int afi_launch_inc_i64(long long* p, hipStream_t st) {
    hipLaunchKernelGGL(afi_inc_i64_kernel, dim3(1), dim3(64), 0, st, p);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
// eval-mode BatchNorm: invstd[c] = rsqrt(running_var[c] + eps)
__global__ void afi_invstd_kernel(const float* __restrict__ var, float* __restrict__ invstd, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) invstd[c] = rsqrtf(var[c] + AFI_BN_EPS);
}
int afi_launch_invstd(const float* var, float* invstd, int C, hipStream_t st) {
    hipLaunchKernelGGL(afi_invstd_kernel, dim3(afi_cdiv(C, 256)), dim3(256), 0, st, var, invstd, C);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
