// Winograd F(2x2, 3x3) transforms around the batched 1x1 MFMA GEMM (large maps, many channels).
//
// A 3x3 / stride-1 / pad-1 correlation over a 2x2 output tile costs 16 multiplies per (cin, cout) pair in the transformed
// domain instead of 36:   Y = A^T [ (G g G^T) (.) (B^T d B) ] A   (Lavin & Gray, 2016).  Per layer:
//
//   U[a][n][k]  = G g G^T          weight transform      (afi_wino_weight_kernel; once per call, weights are small)
//   V[a][t][k]  = B^T d B          input transform       (afi_wino_input_kernel;  HBM-bound: reads X, writes 4x its volume)
//   M[a][t][n]  = sum_k V[a][t][k] * U[a][n][k]          16 independent GEMMs = ONE launch of the pixel GEMM (ntaps = 1,
//                                                         "images" = the 16 transform points, per-image weight stride)
//   Y           = A^T M A (+ bias, * LeakyReLU'(Z))      output transform (afi_wino_output_kernel; HBM-bound)
//
// a = 4*i + j indexes the 4x4 transform points, t the 2x2 output tiles (N * ceil(H/2) * ceil(W/2), padded to a multiple of
// 128 so a GEMM tile never straddles two transform points), k the input and n the output channels.  The matrix-core work
// drops 2.25x; the price is ~10 GB of transform traffic for the largest layer (2 x 200 x 336 x 1024 -> 1024), ~2.2 ms next
// to 8.3 ms of GEMM instead of 18.6 ms of direct convolution.  fp32 throughout; the transform constants are 0, +-1, +-1/2,
// so the result differs from the direct kernel by a few ulp of the accumulated sum (tests hold it to the same 1e-3 bar).
#include "afi_common.h"
#include "afi_epilogue.h"
#include "afi_bn.h"
#include "afi_f16_split.h"

// ---------------------------------------------------------------- largest magnitude of a transform's SOURCE tensor, as a by-product
// (the f16x3 arithmetic of the batched GEMMs, afi_gemm_f16.h: every plane of a transform is bounded by a constant times this value, and the
// GEMM derives its power-of-two operand scale from it).  A thread keeps the maximum of what it loads (after the BatchNorm affine, where
// the tensor is read through one); at the end of its grid-stride walk the block reduces and publishes with ONE conditional atomic max on
// the bit pattern (non-negative floats order like unsigned integers).  The slot is zero-filled before the launch.
__device__ __forceinline__ float afi_amax4(float m, f32x4 v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
}
__device__ __forceinline__ void afi_amax_publish(float m, float* slot) {
    __shared__ float red[16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const int tid = threadIdx.x + blockDim.x * threadIdx.y, nw = (blockDim.x * blockDim.y + 63) >> 6;
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < nw; ++w) m = fmaxf(m, red[w]);
        const unsigned bits = __float_as_uint(m);
        // (a relaxed device-scope read first: after the first round of blocks most blocks cannot raise the slot, and thousands of atomics
        //  on one word would serialise at the memory side)
        if (bits > __hip_atomic_load((const unsigned*)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) (void)atomicMax((unsigned*)slot, bits);
    }
}

// ---------------------------------------------------------------- planes written already split (f16x3, source maximum known beforehand)
// Where the producer of the source tensor has published its largest magnitude (the BatchNorm passes of the discriminator do, as a
// by-product), a transform knows every plane's scale s_a = 2^(14 - floor(log2(c_a amax))) before it starts and writes x s_a as two fp16
// pieces instead of one fp32: the row keeps its 4 K bytes -- [32-channel block][hi: 32 x fp16 | lo: 32 x fp16] -- and the GEMMs stage the
// pieces by LDS-DMA with no conversion instruction left in their loops.  A thread's float4 (channels c .. c + 3) becomes 8 bytes of hi
// and 8 bytes of lo; dst_row = the row's first float in the fp32 layout (same address arithmetic, same pitch).
__device__ __forceinline__ void afi_store_split4(float* dst_row, int c, f32x4 v, float s) {
    unsigned char* b = (unsigned char*)dst_row + (c >> 5) * 128 + (c & 31) * 2;
    unsigned h0, h1, l0, l1;
    afi_split2_f16_pair_cvt(v[0], v[1], s, h0, l0);
    afi_split2_f16_pair_cvt(v[2], v[3], s, h1, l1);
    *(u32x2*)b = u32x2{h0, h1};
    *(u32x2*)(b + 64) = u32x2{l0, l1};
}

static bool afi_epilogue_is_simple_host(const AfiPixGemm& p) {
    return p.o_up == 1 && p.beta == 0.f && !p.R1.p && !p.R2.p && !p.r2_post && p.oH >= p.H && p.oW >= p.W &&
           (!p.Z.p || (p.z_lo == 0 && p.z_hi >= p.Ncols));
}
static unsigned wino_grid(long long work_items) {
    long long g = (work_items + 255) / 256;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ---------------------------------------------------------------- weights: w [O][3][3][I] (memory order) -> U
// mode 0 (forward):   U[a][o][i] from g[ky][kx] = w[o][ky][kx][i]            GEMM columns = O, K = I
// mode 1 (data grad): U[a][i][o] from g[ky][kx] = w[o][2-ky][2-kx][i]        GEMM columns = I, K = O  (flipped taps, swapped roles)
// wmax (optional): raised to the largest magnitude of w (the f16x3 arithmetic bounds every plane of U by a constant times it; zero-filled by the caller)
__global__ void afi_wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int O, int I, int mode, float* wmax) {
    const long long total = (long long)O * I;
    float am = 0.f;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e % I), o = (int)(e / I);
        float g[3][3];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int tt = mode ? 8 - t : t;
            g[t / 3][t % 3] = w[((long long)o * 9 + tt) * I + i];
            am = fmaxf(am, fabsf(g[t / 3][t % 3]));
        }
        float a[4][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            a[0][j] = g[0][j];
            a[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
            a[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
            a[3][j] = g[2][j];
        }
        const long long plane = (long long)O * I;
        const long long off = mode ? (long long)i * O + o : (long long)o * I + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float u0 = a[r][0], u1 = 0.5f * (a[r][0] + a[r][1] + a[r][2]), u2 = 0.5f * (a[r][0] - a[r][1] + a[r][2]), u3 = a[r][2];
            U[(4 * r + 0) * plane + off] = u0;
            U[(4 * r + 1) * plane + off] = u1;
            U[(4 * r + 2) * plane + off] = u2;
            U[(4 * r + 3) * plane + off] = u3;
        }
    }
    if (wmax) afi_amax_publish(am, wmax);                    // (uniform)
}
int afi_launch_wino_weight(const float* w, float* U, int O, int I, int mode, hipStream_t st, float* wmax) {
    if (O <= 0 || I <= 0) return AFI_ERR_BAD_ARG;
    hipLaunchKernelGGL(afi_wino_weight_kernel, dim3(wino_grid((long long)O * I)), dim3(256), 0, st, w, U, O, I, mode, wmax);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// ---------------------------------------------------------------- input: X (view, [N][H][W][C]) -> V [16][Tpad][C]
// thread = (tile, channel quad); the 4x4 patch starts at (2*ty - 1, 2*tx - 1), zeros outside the image
// BN: x is read through a BatchNorm affine + LeakyReLU (AfiBnLoad, afi_bn.h) -- the input is a discriminator block's saved conv output
// AM: 0 fp32 planes; 1 fp32 planes + the source's largest magnitude raised into *amax; 2 planes split into fp16 pieces with the scales of `bnd`
template <bool BN, int AM = 0>
__global__ __launch_bounds__(256) void afi_wino_input_kernel(const AfiView x, int N, int H, int W, int C, int Th, int Tw, long long T,
                                                             long long Tpad, float* __restrict__ Vout, long long ldo, const AfiBnLoad bn, float* amax, const AfiF16Bound bnd) {
    constexpr bool AMAX = AM == 1;
    const float src_max = AM == 2 ? bnd.amax[0] : 0.f;
    const int C4 = C >> 2;
    const long long total = Tpad * C4;
    const long long plane = Tpad * ldo;                    // ldo: row pitch of the plane (>= C: this call may fill a channel slice)
    float am = 0.f;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C4) * 4;
        const long long t = e / C4;
        float* dst = Vout + t * ldo + c;
        float* drow = Vout + t * ldo;
        auto put = [&](int a, f32x4 v) {
            if (AM == 2) afi_store_split4(drow + a * plane, c, v, afi_f16_scale(src_max * bnd.cmul[a]));
            else *(f32x4*)(dst + a * plane) = v;
        };
        if (t >= T) {                                        // padding tiles: zeros (their GEMM rows are never read back)
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 16; ++a) *(f32x4*)(dst + a * plane) = z;       // (all-zero bits are zeros in either layout)
            continue;
        }
        const int tx = (int)(t % Tw); const long long r = t / Tw; const int ty = (int)(r % Th); const int n = (int)(r / Th);
        const float* base = x.p + (long long)n * x.sN + c;
        f32x4 mu, is, ga, be;
        if constexpr (BN) { mu = *(const f32x4*)(bn.mean + c); is = *(const f32x4*)(bn.invstd + c); ga = *(const f32x4*)(bn.gamma + c); be = *(const f32x4*)(bn.beta + c); }
        f32x4 d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = 2 * ty - 1 + i;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xx = 2 * tx - 1 + j;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
                    v = *(const f32x4*)(base + (long long)yy * x.sH + (long long)xx * x.sW);
                    if constexpr (BN) v = afi_bn_lrelu(v, mu, is, ga, be, AFI_LRELU_SLOPE);
                }
                d[i][j] = v;
            }
        }
        if constexpr (AMAX) {                                // (behind ALL the loads: a use inside a load's own branch makes every load wait for itself)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) am = afi_amax4(am, d[i][j]);
        }
        f32x4 s[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                        // B^T d
            s[0][j] = d[0][j] - d[2][j];
            s[1][j] = d[1][j] + d[2][j];
            s[2][j] = d[2][j] - d[1][j];
            s[3][j] = d[1][j] - d[3][j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                        // (B^T d) B
            put(4 * i + 0, s[i][0] - s[i][2]);
            put(4 * i + 1, s[i][1] + s[i][2]);
            put(4 * i + 2, s[i][2] - s[i][1]);
            put(4 * i + 3, s[i][1] - s[i][3]);
        }
    }
    if constexpr (AMAX) afi_amax_publish(am, amax);
}
static inline bool wino_bn_ok(const AfiBnLoad* bn) {
    return !bn || !bn->mean || (bn->invstd && bn->gamma && bn->beta && !((((uintptr_t)bn->mean) | ((uintptr_t)bn->invstd) | ((uintptr_t)bn->gamma) | ((uintptr_t)bn->beta)) & 15));
}
// amax (optional): raised to the largest magnitude of what the launch reads of x (see afi_amax_publish; zero-filled by the caller).
// pre (optional, instead): the planes are written split into fp16 pieces with the scales pre->amax[0] x pre->cmul[plane] (ldo, C multiples of 32)
static const AfiF16Bound kNoBound = {nullptr, 0, 0, {0}};
int afi_launch_wino_input(AfiView x, int N, int H, int W, int C, long long Tpad, float* V, hipStream_t st, long long ldo, const AfiBnLoad* bn, float* amax,
                          const AfiF16Bound* pre) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || !wino_bn_ok(bn)) return AFI_ERR_BAD_ARG;
    const int Th = (H + 1) / 2, Tw = (W + 1) / 2;
    const long long T = (long long)N * Th * Tw;
    if (Tpad < T) return AFI_ERR_BAD_ARG;
    const AfiBnLoad off{nullptr, nullptr, nullptr, nullptr};
    const dim3 grid(wino_grid(Tpad * (C >> 2))), blk(256);
    const long long ld = ldo > 0 ? ldo : (long long)C;
    const bool b = bn && bn->mean;
    if (pre) {
        if (!pre->amax || (ld & 31) || (C & 31) || amax) return AFI_ERR_BAD_ARG;
        if (b) hipLaunchKernelGGL((afi_wino_input_kernel<true, 2>), grid, blk, 0, st, x, N, H, W, C, Th, Tw, T, Tpad, V, ld, *bn, amax, *pre);
        else hipLaunchKernelGGL((afi_wino_input_kernel<false, 2>), grid, blk, 0, st, x, N, H, W, C, Th, Tw, T, Tpad, V, ld, off, amax, *pre);
    } else if (b && amax) hipLaunchKernelGGL((afi_wino_input_kernel<true, 1>), grid, blk, 0, st, x, N, H, W, C, Th, Tw, T, Tpad, V, ld, *bn, amax, kNoBound);
    else if (b) hipLaunchKernelGGL((afi_wino_input_kernel<true, 0>), grid, blk, 0, st, x, N, H, W, C, Th, Tw, T, Tpad, V, ld, *bn, amax, kNoBound);
    else if (amax) hipLaunchKernelGGL((afi_wino_input_kernel<false, 1>), grid, blk, 0, st, x, N, H, W, C, Th, Tw, T, Tpad, V, ld, off, amax, kNoBound);
    else hipLaunchKernelGGL((afi_wino_input_kernel<false, 0>), grid, blk, 0, st, x, N, H, W, C, Th, Tw, T, Tpad, V, ld, off, amax, kNoBound);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// ---------------------------------------------------------------- output: M [16][Tpad][C] -> Y (view), 2x2 pixels per tile
// y = alpha * (A^T m A) + bias[c];  then * (Z > 0 ? 1 : 0.2) when a mask tensor is given (the LeakyReLU' of the dgrad chain)
__global__ __launch_bounds__(256) void afi_wino_output_kernel(const float* __restrict__ Min, long long Tpad, int N, int H, int W, int C, int Th, int Tw,
                                                              long long T, const float* __restrict__ bias, float alpha, const AfiView out, const AfiView z) {
    const int C4 = C >> 2;
    const long long total = T * C4;
    const long long plane = Tpad * C;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C4) * 4;
        const long long t = e / C4;
        const int tx = (int)(t % Tw); const long long r = t / Tw; const int ty = (int)(r % Th); const int n = (int)(r / Th);
        const float* src = Min + t * C + c;
        f32x4 m[4][4];
#pragma unroll
        for (int a = 0; a < 16; ++a) m[a >> 2][a & 3] = *(const f32x4*)(src + a * plane);
        f32x4 s[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                        // A^T m
            s[0][j] = m[0][j] + m[1][j] + m[2][j];
            s[1][j] = m[1][j] - m[2][j] - m[3][j];
        }
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (bias) b = *(const f32x4*)(bias + c);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int yy = 2 * ty + i;
            if (yy >= H) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int xx = 2 * tx + j;
                if (xx >= W) continue;
                f32x4 v = (j == 0) ? s[i][0] + s[i][1] + s[i][2] : s[i][1] - s[i][2] - s[i][3];
                v = alpha * v + b;
                if (z.p) {
                    const f32x4 zz = *(const f32x4*)(z.p + (long long)n * z.sN + (long long)yy * z.sH + (long long)xx * z.sW + c);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] *= (zz[q] > 0.f ? 1.f : AFI_LRELU_SLOPE);
                }
                *(f32x4*)(out.p + (long long)n * out.sN + (long long)yy * out.sH + (long long)xx * out.sW + c) = v;
            }
        }
    }
}
int afi_launch_wino_output(const float* M, long long Tpad, int N, int H, int W, int C, const float* bias, float alpha, AfiView out, AfiView z,
                           hipStream_t st) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return AFI_ERR_BAD_ARG;
    const int Th = (H + 1) / 2, Tw = (W + 1) / 2;
    const long long T = (long long)N * Th * Tw;
    if (Tpad < T) return AFI_ERR_BAD_ARG;
    hipLaunchKernelGGL(afi_wino_output_kernel, dim3(wino_grid(T * (C >> 2))), dim3(256), 0, st, M, Tpad, N, H, W, C, Th, Tw, T, bias, alpha, out, z);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// ---------------------------------------------------------------- output with the pixel GEMM's full epilogue
// Same transform, but each output float4 goes through afi_epilogue_store with the conv's own descriptor: bias, alpha / beta,
// LeakyReLU, the two residual adds with channel ranges, the bilinear x2 skip, the pixel-shuffle store of the conv-transpose
// (columns = 4 phases x Cout) and the LeakyReLU' mask -- so any 3x3 / stride-1 conv of the interpolator can take this path.
// ---------------------------------------------------------------- BatchNorm statistics inside the output transforms
// STATS variants of the two output-transform kernels below: every thread keeps the sum and the sum of squares of the values it STORES, per
// channel of its float4, in fp64 (torch's CPU accumulation type for float; no shift needed at 53 bits), the block reduces them over the
// threads that share a channel quad (blockDim % (C / 4) == 0: a thread's quad never changes along its grid-stride walk) and writes ONE
// row of partials [2][C]; afi_launch_bn_stats_from_partials sums the rows in a fixed order.  Replaces a full pass over the map
// (afi_bn_stats_partial_kernel: 1.85 ms of a stage-1 step).
#ifndef AFI_STATS_MAX_ROWS
#define AFI_STATS_MAX_ROWS 1024
#endif
typedef double f64x4w __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void afi_stats_acc(f64x4w& s0, f64x4w& s1, f32x4& mn, f32x4& mx, f32x4 v) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { const double d = (double)v[j]; s0[j] += d; s1[j] += d * d; mn[j] = fminf(mn[j], v[j]); mx[j] = fmaxf(mx[j], v[j]); }
}
// the backward sums of one stored float4 (AfiPixGemm::bstats): o = d(loss)/d(activation) at (pixel row, channel quad c), cv the conv output there
__device__ __forceinline__ void afi_bstats_acc(f64x4w& s0, f64x4w& s1, f32x4 o, f32x4 cv, f32x4 mu, f32x4 is, f32x4 ga, f32x4 be, float slope) {
    const f32x4 z = afi_bn_affine(cv, mu, is, ga, be);
    const f32x4 xh = (cv - mu) * is;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double gm = (double)(z[j] > 0.f ? o[j] : o[j] * slope);
        s0[j] += gm; s1[j] += gm * (double)xh[j];
    }
}
__device__ __forceinline__ void afi_stats_block_write(const AfiPixGemm& p, f64x4w s0, f64x4w s1, f32x4 mn, f32x4 mx, double* dst_rows = nullptr) {
    __shared__ f64x4w red[2][256];
    __shared__ f32x4 redm[2][256];
    const int C4 = p.Ncols >> 2;
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
    redm[0][threadIdx.x] = mn; redm[1][threadIdx.x] = mx;
    __syncthreads();
    if ((int)threadIdx.x < C4) {
        for (int k = threadIdx.x + C4; k < 256; k += C4) {                                           // fixed order: bit-reproducible
            s0 += red[0][k]; s1 += red[1][k];
#pragma unroll
            for (int j = 0; j < 4; ++j) { mn[j] = fminf(mn[j], redm[0][k][j]); mx[j] = fmaxf(mx[j], redm[1][k][j]); }
        }
        double* row = (dst_rows ? dst_rows : p.stats) + (long long)blockIdx.x * 2 * p.Ncols;
        *(f64x4w*)(row + 4 * threadIdx.x) = s0;
        *(f64x4w*)(row + p.Ncols + 4 * threadIdx.x) = s1;
        if (p.stats_mm && !dst_rows) {                                                                            // (a thread that stored nothing leaves +-inf: neutral)
            float* mrow = p.stats_mm + (long long)blockIdx.x * 2 * p.Ncols;
            *(f32x4*)(mrow + 4 * threadIdx.x) = mn;
            *(f32x4*)(mrow + p.Ncols + 4 * threadIdx.x) = mx;
        }
    }
}
// rows of partials (= blocks) the STATS launch of an output transform over T tiles x C channels uses; 0 = this shape is not fused
int afi_wino_stats_rows(long long T, int C) {
    if (C != 256 && C != 512 && C != 1024) return 0;
    long long g = (T * (C >> 2) + 255) / 256;
    if (g > AFI_STATS_MAX_ROWS) g = AFI_STATS_MAX_ROWS;
    return (int)(g < 1 ? 1 : g);
}
static bool afi_stats_fusable(const AfiPixGemm& p) { return (p.stats || p.bstats) && (p.Ncols == 256 || p.Ncols == 512 || p.Ncols == 1024); }

// STATS: 0 none, 1 the forward's statistics of the stored output, 2 the BatchNorm-backward sums of it (AfiPixGemm::bstats)
template <bool SIMPLE, int STATS = 0>
__global__ __launch_bounds__(256) void afi_wino_output_epi_kernel(const float* __restrict__ Min, long long Tpad, int Th, int Tw, long long T,
                                                                  const AfiPixGemm p) {
    const int C = p.Ncols, C4 = C >> 2;
    const long long total = T * C4;
    const long long plane = Tpad * C;
    f64x4w st0 = {0, 0, 0, 0}, st1 = {0, 0, 0, 0};
    f32x4 smn = {INFINITY, INFINITY, INFINITY, INFINITY}, smx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    f32x4 bmu = {0, 0, 0, 0}, bis = bmu, bga = bmu, bbe = bmu;   // STATS == 2: this thread's channel quad never changes along its walk (blockDim % C4 == 0)
    if (STATS == 2) {
        const int cq = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) % C4) * 4;
        bmu = *(const f32x4*)(p.bstats_bn.mean + cq); bis = *(const f32x4*)(p.bstats_bn.invstd + cq);
        bga = *(const f32x4*)(p.bstats_bn.gamma + cq); bbe = *(const f32x4*)(p.bstats_bn.beta + cq);
    }
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C4) * 4;
        const long long t = e / C4;
        const int tx = (int)(t % Tw); const long long r = t / Tw; const int ty = (int)(r % Th); const int n = (int)(r / Th);
        const float* src = Min + t * C + c;
        f32x4 m[4][4];
#pragma unroll
        for (int a = 0; a < 16; ++a) m[a >> 2][a & 3] = __builtin_nontemporal_load((const f32x4*)(src + a * plane));   // M is read exactly once
        f32x4 s[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[0][j] = m[0][j] + m[1][j] + m[2][j];
            s[1][j] = m[1][j] - m[2][j] - m[3][j];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int yy = 2 * ty + i;
            if (yy >= p.H) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int xx = 2 * tx + j;
                if (xx >= p.W) continue;
                const f32x4 v = (j == 0) ? s[i][0] + s[i][1] + s[i][2] : s[i][1] - s[i][2] - s[i][3];
                if (SIMPLE) {
                    const f32x4 o = afi_epilogue_store_simple(p, n, yy, xx, c, v);
                    if (STATS == 1) afi_stats_acc(st0, st1, smn, smx, o);
                    if (STATS == 2) {
                        const f32x4 cv = __builtin_nontemporal_load((const f32x4*)(p.bstats_c + (((long long)n * p.H + yy) * p.W + xx) * C + c));
                        afi_bstats_acc(st0, st1, o, cv, bmu, bis, bga, bbe, p.bstats_slope);
                    }
                } else afi_epilogue_store(p, n, yy, xx, c, v);
            }
        }
    }
    if (STATS) afi_stats_block_write(p, st0, st1, smn, smx, STATS == 2 ? p.bstats : nullptr);
}
int afi_launch_wino_output_epi(const float* M, long long Tpad, const AfiPixGemm& p, hipStream_t st) {
    if (p.N <= 0 || p.H <= 0 || p.W <= 0 || p.Ncols <= 0 || (p.Ncols & 3)) return AFI_ERR_BAD_ARG;
    const int Th = (p.H + 1) / 2, Tw = (p.W + 1) / 2;
    const long long T = (long long)p.N * Th * Tw;
    if (Tpad < T) return AFI_ERR_BAD_ARG;
    if (p.stats || p.bstats) {                              // (the caller asked for fused statistics: afi_wino_stats_rows said this shape takes them)
        if (!afi_epilogue_is_simple_host(p) || !afi_stats_fusable(p) || (p.stats && p.bstats) || (p.bstats && (!p.bstats_c || !p.bstats_bn.mean))) return AFI_ERR_BAD_ARG;
        AfiPixGemm q = p;
        q.stats_rows = afi_wino_stats_rows(T, p.Ncols);
        if (p.stats) hipLaunchKernelGGL((afi_wino_output_epi_kernel<true, 1>), dim3(q.stats_rows), dim3(256), 0, st, M, Tpad, Th, Tw, T, q);
        else hipLaunchKernelGGL((afi_wino_output_epi_kernel<true, 2>), dim3(q.stats_rows), dim3(256), 0, st, M, Tpad, Th, Tw, T, q);
    } else if (afi_epilogue_is_simple_host(p)) hipLaunchKernelGGL((afi_wino_output_epi_kernel<true>), dim3(wino_grid(T * (p.Ncols >> 2))), dim3(256), 0, st, M, Tpad, Th, Tw, T, p);
    else hipLaunchKernelGGL((afi_wino_output_epi_kernel<false>), dim3(wino_grid(T * (p.Ncols >> 2))), dim3(256), 0, st, M, Tpad, Th, Tw, T, p);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// ================================================================ weight gradient: F(3x3, 2x2)
//   dW[ky][kx] = sum_tiles sum_{i,j<2} dY[2ty+i][2tx+j] * X[2ty-1+i+ky][2tx-1+j+kx]
// is a correlation of the SAME 4x4 input patch with the tile's 2x2 block of dY: with the interpolation points of F(2,3) it
// needs 16 products per (co, ci) and tile instead of 36, and B^T is the same matrix, so V = B^T d B is shared with the forward:
//   Q[a][t][co] = G' dy G'^T                     (afi_wino_dy_kernel;  G' = [[1,0],[1/2,1/2],[1/2,-1/2],[0,1]])
//   dU[a][co][ci] = sum_t Q[a][t][co] * V[a][t][ci]     16 GEMMs with K = tiles: ONE launch of the weight-gradient kernel
//   dW[co][ky][kx][ci] += A'^T dU A'             (afi_wino_dw_kernel;  A'^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,-1]])
template <int AM>
__global__ __launch_bounds__(256) void afi_wino_dy_kernel(const AfiView dy, int N, int H, int W, int C, int Th, int Tw, long long T, long long Tpad,
                                                          float* __restrict__ Q, long long ldo, float* amax, const AfiF16Bound bnd) {
    constexpr bool AMAX = AM == 1;
    const float src_max = AM == 2 ? bnd.amax[0] : 0.f;
    const int C4 = C >> 2;
    const long long total = Tpad * C4;
    const long long plane = Tpad * ldo;                    // ldo: row pitch of the plane (>= C: this call may fill a channel slice)
    float am = 0.f;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C4) * 4;
        const long long t = e / C4;
        float* dst = Q + t * ldo + c;
        float* drow = Q + t * ldo;
        auto put = [&](int a, f32x4 v) {
            if (AM == 2) afi_store_split4(drow + a * plane, c, v, afi_f16_scale(src_max * bnd.cmul[a]));
            else *(f32x4*)(dst + a * plane) = v;
        };
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if (t >= T) {
#pragma unroll
            for (int a = 0; a < 16; ++a) *(f32x4*)(dst + a * plane) = zero;
            continue;
        }
        const int tx = (int)(t % Tw); const long long r = t / Tw; const int ty = (int)(r % Th); const int n = (int)(r / Th);
        const float* base = dy.p + (long long)n * dy.sN + c;
        f32x4 d[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int yy = 2 * ty + i, xx = 2 * tx + j;
                d[i][j] = (yy < H && xx < W) ? *(const f32x4*)(base + (long long)yy * dy.sH + (long long)xx * dy.sW) : zero;
            }
        if constexpr (AMAX) am = afi_amax4(afi_amax4(afi_amax4(afi_amax4(am, d[0][0]), d[0][1]), d[1][0]), d[1][1]);
        f32x4 a[4][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            a[0][j] = d[0][j];
            a[1][j] = 0.5f * (d[0][j] + d[1][j]);
            a[2][j] = 0.5f * (d[0][j] - d[1][j]);
            a[3][j] = d[1][j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            put(4 * i + 0, a[i][0]);
            put(4 * i + 1, 0.5f * (a[i][0] + a[i][1]));
            put(4 * i + 2, 0.5f * (a[i][0] - a[i][1]));
            put(4 * i + 3, a[i][1]);
        }
    }
    if constexpr (AMAX) afi_amax_publish(am, amax);
}
int afi_launch_wino_dy(AfiView dy, int N, int H, int W, int C, long long Tpad, float* Q, hipStream_t st, long long ldo, float* amax, const AfiF16Bound* pre) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return AFI_ERR_BAD_ARG;
    const int Th = (H + 1) / 2, Tw = (W + 1) / 2;
    const long long T = (long long)N * Th * Tw;
    if (Tpad < T) return AFI_ERR_BAD_ARG;
    const long long ld = ldo > 0 ? ldo : (long long)C;
    if (pre) {
        if (!pre->amax || (ld & 31) || (C & 31) || amax) return AFI_ERR_BAD_ARG;
        hipLaunchKernelGGL(afi_wino_dy_kernel<2>, dim3(wino_grid(Tpad * (C >> 2))), dim3(256), 0, st, dy, N, H, W, C, Th, Tw, T, Tpad, Q, ld, amax, *pre);
    } else if (amax) hipLaunchKernelGGL(afi_wino_dy_kernel<1>, dim3(wino_grid(Tpad * (C >> 2))), dim3(256), 0, st, dy, N, H, W, C, Th, Tw, T, Tpad, Q, ld, amax, kNoBound);
    else hipLaunchKernelGGL(afi_wino_dy_kernel<0>, dim3(wino_grid(Tpad * (C >> 2))), dim3(256), 0, st, dy, N, H, W, C, Th, Tw, T, Tpad, Q, ld, amax, kNoBound);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// dW[o][ky][kx][i] += alpha * (A'^T dU A')[ky][kx]  with dU [16][O][I]
__global__ void afi_wino_dw_kernel(const float* __restrict__ dU, float* __restrict__ dW, int O, int I, float alpha) {
    const long long total = (long long)O * I;
    const long long plane = total;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e % I), o = (int)(e / I);
        float m[4][4];
#pragma unroll
        for (int a = 0; a < 16; ++a) m[a >> 2][a & 3] = dU[a * plane + e];
        float s[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[0][j] = m[0][j] + m[1][j] + m[2][j];
            s[1][j] = m[1][j] - m[2][j];
            s[2][j] = m[1][j] + m[2][j] - m[3][j];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float* dst = dW + ((long long)o * 9 + 3 * k) * I + i;
            dst[0] += alpha * (s[k][0] + s[k][1] + s[k][2]);
            dst[I] += alpha * (s[k][1] - s[k][2]);
            dst[2 * I] += alpha * (s[k][1] + s[k][2] - s[k][3]);
        }
    }
}
int afi_launch_wino_dw(const float* dU, float* dW, int O, int I, float alpha, hipStream_t st) {
    if (O <= 0 || I <= 0) return AFI_ERR_BAD_ARG;
    hipLaunchKernelGGL(afi_wino_dw_kernel, dim3(wino_grid((long long)O * I)), dim3(256), 0, st, dU, dW, O, I, alpha);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// ================================================================ F(4x4, 3x3) / F(3x3, 4x4): data and weight gradients, forwards that feed no
// backward, and (round 5) the forwards of the discriminator's blocks the caller selects (nets.hip: AFI_OPT_WINOGRAD_F4_FORWARD)
// 36 products per (cin, cout) pair and 4x4 output tile instead of 144 (4x fewer matrix-core FLOPs than direct, 1.78x fewer than
// F(2x2)) and 2.25x (not 4x) transform traffic.  Its fp32 result is less accurate than F(2x2)'s (1e-6 of the output scale): that is
// nothing for gradients, but a forward whose output decides LeakyReLU masks behind a BatchNorm shows it as flipped masks, which is why the
// blocks that take it are chosen by measurement and why the interpolation points below replaced the textbook ones.
// Textbook points {0, +-1, +-2, inf} (AFI_WINO4_POINTS = 0, kept for A/B):
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]           (weights, data gradient)
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]                                 (4x4 outputs)
//   G'  = [1/4 0 0 0; -1/6(1 1 1 1); -1/6(1 -1 1 -1); 1/24(1 2 4 8); 1/24(1 -2 4 -8); 0 0 0 1]     (4x4 block of dY, weight gradient)
//   A'^T= [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 1]                                                 (3x3 weight taps)
// AFI_WINO4_POINTS = 1 (default since round 5, when the discriminator's mask-deciding forwards moved to this form): interpolation points
// {0, 1, -1, 1/2, -2, inf} instead of the textbook {0, +-1, +-2, inf}: the largest row sums of B^T and A^T fall from 10 / 19 to 7 / 11.1 and the
// fp32 rounding of a conv with them (every constant is still exact in fp32 except the 1/3 and 1/15 multiples of G, G').
//   B^T = [1 -3/2 -2 3/2 1 0; 0 -1 1/2 5/2 1 0; 0 1 -5/2 1/2 1 0; 0 -2 -1 2 1 0; 0 1/2 -1 -1/2 1 0; 0 1 -3/2 -2 3/2 1]
//   G   = [1 0 0; 1/3(1 1 1); -1/3(1 -1 1); -16/15(1 1/2 1/4); 1/15(1 -2 4); 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 1/2 -2 0; 0 1 1 1/4 4 0; 0 1 -1 1/8 -8 1]
//   G'  = [1 0 0 0; 1/3(1 1 1 1); -1/3(1 -1 1 -1); -16/15(1 1/2 1/4 1/8); 1/15(1 -2 4 -8); 0 0 0 1]
//   A'^T= [1 1 1 1 1 0; 0 1 -1 1/2 -2 0; 0 1 1 1/4 4 1]
// (AFI_WINO4_POINTS is defined in afi_common.h: igemm.hip's plane bounds follow the same choice)
template <typename T>
__device__ __forceinline__ void wino4_bt(T& d0, T& d1, T& d2, T& d3, T& d4, T& d5) {      // in place: B^T d
#if AFI_WINO4_POINTS == 1
    const T t0 = d0 + 1.5f * (d3 - d1) - 2.f * d2 + d4;
    const T t1 = -d1 + 0.5f * d2 + 2.5f * d3 + d4;
    const T t2 = d1 - 2.5f * d2 + 0.5f * d3 + d4;
    const T t3 = 2.f * (d3 - d1) - d2 + d4;
    const T t4 = 0.5f * (d1 - d3) - d2 + d4;
    const T t5 = d1 + 1.5f * (d4 - d2) - 2.f * d3 + d5;
#else
    const T t0 = 4.f * d0 - 5.f * d2 + d4;
    const T t1 = -4.f * (d1 + d2) + d3 + d4;
    const T t2 = 4.f * (d1 - d2) - d3 + d4;
    const T t3 = -2.f * d1 - d2 + 2.f * d3 + d4;
    const T t4 = 2.f * d1 - d2 - 2.f * d3 + d4;
    const T t5 = 4.f * d1 - 5.f * d3 + d5;
#endif
    d0 = t0; d1 = t1; d2 = t2; d3 = t3; d4 = t4; d5 = t5;
}
// G applied to three taps, G' to four samples (one definition for the weight and the dY transforms)
template <typename T>
__device__ __forceinline__ void wino4_g3(const T g0, const T g1, const T g2, T (&a)[6]) {
#if AFI_WINO4_POINTS == 1
    a[0] = g0;
    a[1] = (1.f / 3.f) * (g0 + g1 + g2);
    a[2] = (-1.f / 3.f) * (g0 - g1 + g2);
    a[3] = (-16.f / 15.f) * g0 + (-8.f / 15.f) * g1 + (-4.f / 15.f) * g2;
    a[4] = (1.f / 15.f) * g0 + (-2.f / 15.f) * g1 + (4.f / 15.f) * g2;
    a[5] = g2;
#else
    a[0] = 0.25f * g0;
    a[1] = (-1.f / 6.f) * (g0 + g1 + g2);
    a[2] = (-1.f / 6.f) * (g0 - g1 + g2);
    a[3] = (1.f / 24.f) * g0 + (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
    a[4] = (1.f / 24.f) * g0 - (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
    a[5] = g2;
#endif
}
template <typename T>
__device__ __forceinline__ void wino4_g4(const T v0, const T v1, const T v2, const T v3, T (&a)[6]) {
#if AFI_WINO4_POINTS == 1
    a[0] = v0;
    a[1] = (1.f / 3.f) * (v0 + v1 + v2 + v3);
    a[2] = (-1.f / 3.f) * (v0 - v1 + v2 - v3);
    a[3] = (-16.f / 15.f) * v0 + (-8.f / 15.f) * v1 + (-4.f / 15.f) * v2 + (-2.f / 15.f) * v3;
    a[4] = (1.f / 15.f) * v0 + (-2.f / 15.f) * v1 + (4.f / 15.f) * v2 + (-8.f / 15.f) * v3;
    a[5] = v3;
#else
    a[0] = 0.25f * v0;
    a[1] = (-1.f / 6.f) * (v0 + v1 + v2 + v3);
    a[2] = (-1.f / 6.f) * (v0 - v1 + v2 - v3);
    a[3] = (1.f / 24.f) * v0 + (1.f / 12.f) * v1 + (1.f / 6.f) * v2 + (1.f / 3.f) * v3;
    a[4] = (1.f / 24.f) * v0 - (1.f / 12.f) * v1 + (1.f / 6.f) * v2 - (1.f / 3.f) * v3;
    a[5] = v3;
#endif
}
// A^T m (four outputs) and A'^T m (three weight taps) of six transform-domain values
template <typename T>
__device__ __forceinline__ void wino4_at(const T m0, const T m1, const T m2, const T m3, const T m4, const T m5, T& y0, T& y1, T& y2, T& y3) {
    const T p12 = m1 + m2, d12 = m1 - m2;
#if AFI_WINO4_POINTS == 1
    y0 = m0 + p12 + m3 + m4;
    y1 = d12 + 0.5f * m3 - 2.f * m4;
    y2 = p12 + 0.25f * m3 + 4.f * m4;
    y3 = d12 + 0.125f * m3 - 8.f * m4 + m5;
#else
    const T p34 = m3 + m4, d34 = m3 - m4;
    y0 = m0 + p12 + p34;
    y1 = d12 + 2.f * d34;
    y2 = p12 + 4.f * p34;
    y3 = d12 + 8.f * d34 + m5;
#endif
}
template <typename T>
__device__ __forceinline__ void wino4_at3(const T m0, const T m1, const T m2, const T m3, const T m4, const T m5, T& y0, T& y1, T& y2) {
#if AFI_WINO4_POINTS == 1
    y0 = m0 + m1 + m2 + m3 + m4;
    y1 = (m1 - m2) + 0.5f * m3 - 2.f * m4;
    y2 = (m1 + m2) + 0.25f * m3 + 4.f * m4 + m5;
#else
    y0 = m0 + m1 + m2 + m3 + m4;
    y1 = (m1 - m2) + 2.f * (m3 - m4);
    y2 = (m1 + m2) + 4.f * (m3 + m4) + m5;
#endif
}

// input: X (view) -> V [36][Tpad][C]; the 6x6 patch of tile (ty, tx) starts at (4*ty - 1, 4*tx - 1)
template <bool BN, int AM = 0>
__global__ __launch_bounds__(256) void afi_wino4_input_kernel(const AfiView x, int N, int H, int W, int C, int Th, int Tw, long long T, long long Tpad,
                                                              float* __restrict__ Vout, long long ldo, const AfiBnLoad bn, float* amax, const AfiF16Bound bnd) {
    constexpr bool AMAX = AM == 1;
    const float src_max = AM == 2 ? bnd.amax[0] : 0.f;
    const int C4 = C >> 2;
    const long long total = Tpad * C4;
    const long long plane = Tpad * ldo;                    // ldo: row pitch of the plane (>= C: this call may fill a channel slice)
    float am = 0.f;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C4) * 4;
        const long long t = e / C4;
        float* dst = Vout + t * ldo + c;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if (t >= T) {
#pragma unroll
            for (int a = 0; a < 36; ++a) *(f32x4*)(dst + a * plane) = zero;
            continue;
        }
        const int tx = (int)(t % Tw); const long long r = t / Tw; const int ty = (int)(r % Th); const int n = (int)(r / Th);
        const float* base = x.p + (long long)n * x.sN + c;
        f32x4 mu, is, ga, be;
        if constexpr (BN) { mu = *(const f32x4*)(bn.mean + c); is = *(const f32x4*)(bn.invstd + c); ga = *(const f32x4*)(bn.gamma + c); be = *(const f32x4*)(bn.beta + c); }
        f32x4 d[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int yy = 4 * ty - 1 + i;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int xx = 4 * tx - 1 + j;
                f32x4 v = zero;
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
                    v = *(const f32x4*)(base + (long long)yy * x.sH + (long long)xx * x.sW);
                    if constexpr (BN) v = afi_bn_lrelu(v, mu, is, ga, be, AFI_LRELU_SLOPE);
                }
                d[i][j] = v;
            }
        }
        if constexpr (AMAX) {                                // (behind ALL the loads: a use inside a load's own branch makes every load wait for itself)
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) am = afi_amax4(am, d[i][j]);
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) wino4_bt(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);      // columns
#pragma unroll
        for (int i = 0; i < 6; ++i) {                                                                   // rows, stored at once
            wino4_bt(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5]);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                if (AM == 2) afi_store_split4(Vout + t * ldo + (6 * i + j) * plane, c, d[i][j], afi_f16_scale(src_max * bnd.cmul[6 * i + j]));
                else *(f32x4*)(dst + (6 * i + j) * plane) = d[i][j];
            }
        }
    }
    if constexpr (AMAX) afi_amax_publish(am, amax);
}
int afi_launch_wino4_input(AfiView x, int N, int H, int W, int C, long long Tpad, float* V, hipStream_t st, long long ldo, const AfiBnLoad* bn, float* amax,
                           const AfiF16Bound* pre) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || !wino_bn_ok(bn)) return AFI_ERR_BAD_ARG;
    const int Th = (H + 3) / 4, Tw = (W + 3) / 4;
    const long long T = (long long)N * Th * Tw;
    if (Tpad < T) return AFI_ERR_BAD_ARG;
    const AfiBnLoad off{nullptr, nullptr, nullptr, nullptr};
    const dim3 grid(wino_grid(Tpad * (C >> 2))), blk(256);
    const long long ld = ldo > 0 ? ldo : (long long)C;
    const bool b = bn && bn->mean;
    if (pre) {
        if (!pre->amax || (ld & 31) || (C & 31) || amax) return AFI_ERR_BAD_ARG;
        if (b) hipLaunchKernelGGL((afi_wino4_input_kernel<true, 2>), grid, blk, 0, st, x, N, H, W, C, Th, Tw, T, Tpad, V, ld, *bn, amax, *pre);
        else hipLaunchKernelGGL((afi_wino4_input_kernel<false, 2>), grid, blk, 0, st, x, N, H, W, C, Th, Tw, T, Tpad, V, ld, off, amax, *pre);
    } else if (b && amax) hipLaunchKernelGGL((afi_wino4_input_kernel<true, 1>), grid, blk, 0, st, x, N, H, W, C, Th, Tw, T, Tpad, V, ld, *bn, amax, kNoBound);
    else if (b) hipLaunchKernelGGL((afi_wino4_input_kernel<true, 0>), grid, blk, 0, st, x, N, H, W, C, Th, Tw, T, Tpad, V, ld, *bn, amax, kNoBound);
    else if (amax) hipLaunchKernelGGL((afi_wino4_input_kernel<false, 1>), grid, blk, 0, st, x, N, H, W, C, Th, Tw, T, Tpad, V, ld, off, amax, kNoBound);
    else hipLaunchKernelGGL((afi_wino4_input_kernel<false, 0>), grid, blk, 0, st, x, N, H, W, C, Th, Tw, T, Tpad, V, ld, off, amax, kNoBound);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// weights for the data gradient: U[a][i][o] = G g' G^T with g'[ky][kx] = w[o][2-ky][2-kx][i]   (36 planes).
// A 32 x 32 (o, i) tile per 1024-thread block: w is read with i fastest (its memory order), each plane is transposed through
// LDS and written with o fastest (U's order) -- both sides coalesced (the direct form wrote with a stride of O floats).
__global__ __launch_bounds__(1024) void afi_wino4_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int O, int I, int mode, float* wmax) {
    __shared__ float tile[32][33];
    const int ti = threadIdx.x, to = threadIdx.y;
    const int i = blockIdx.x * 32 + ti, o = blockIdx.y * 32 + to;
    const bool ok = i < I && o < O;
    float g[3][3];
    float am = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) { g[t / 3][t % 3] = ok ? w[((long long)o * 9 + (mode ? 8 - t : t)) * I + i] : 0.f; am = fmaxf(am, fabsf(g[t / 3][t % 3])); }   // mode 0: forward (no flip)
    if (wmax) afi_amax_publish(am, wmax);                    // (uniform; its barrier comes before the tile's)
    float a[6][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float col[6];
        wino4_g3(g[0][j], g[1][j], g[2][j], col);
#pragma unroll
        for (int r = 0; r < 6; ++r) a[r][j] = col[r];
    }
    const long long plane = (long long)O * I;
    const int oi = blockIdx.x * 32 + to, oo = blockIdx.y * 32 + ti;      // transposed roles for the store: ti walks o
    const bool ok2 = oi < I && oo < O;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        float u[6];
        wino4_g3(a[r][0], a[r][1], a[r][2], u);
        if (!mode) {                                         // forward: U[a][o][i], the thread's own (o, i): already coalesced
#pragma unroll
            for (int c = 0; c < 6; ++c)
                if (ok) U[(6 * r + c) * plane + (long long)o * I + i] = u[c];
            continue;
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            __syncthreads();
            tile[to][ti] = u[c];                             // [o][i]
            __syncthreads();
            if (ok2) U[(6 * r + c) * plane + (long long)oi * O + oo] = tile[ti][to];
        }
    }
}
int afi_launch_wino4_weight(const float* w, float* U, int O, int I, int mode, hipStream_t st, float* wmax) {
    if (O <= 0 || I <= 0) return AFI_ERR_BAD_ARG;
    hipLaunchKernelGGL(afi_wino4_weight_kernel, dim3((I + 31) / 32, (O + 31) / 32), dim3(32, 32), 0, st, w, U, O, I, mode, wmax);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// output: M [36][Tpad][C] -> 4x4 pixels per tile through the descriptor's epilogue
template <bool SIMPLE, int STATS = 0>
__global__ __launch_bounds__(256) void afi_wino4_output_epi_kernel(const float* __restrict__ Min, long long Tpad, int Th, int Tw, long long T,
                                                                   const AfiPixGemm p) {
    const int C = p.Ncols, C4 = C >> 2;
    const long long total = T * C4;
    const long long plane = Tpad * C;
    f64x4w st0 = {0, 0, 0, 0}, st1 = {0, 0, 0, 0};
    f32x4 smn = {INFINITY, INFINITY, INFINITY, INFINITY}, smx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    f32x4 bmu = {0, 0, 0, 0}, bis = bmu, bga = bmu, bbe = bmu;   // STATS == 2: this thread's channel quad never changes along its walk (blockDim % C4 == 0)
    if (STATS == 2) {
        const int cq = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) % C4) * 4;
        bmu = *(const f32x4*)(p.bstats_bn.mean + cq); bis = *(const f32x4*)(p.bstats_bn.invstd + cq);
        bga = *(const f32x4*)(p.bstats_bn.gamma + cq); bbe = *(const f32x4*)(p.bstats_bn.beta + cq);
    }
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C4) * 4;
        const long long t = e / C4;
        const int tx = (int)(t % Tw); const long long r = t / Tw; const int ty = (int)(r % Th); const int n = (int)(r / Th);
        const float* src = Min + t * C + c;
        f32x4 s[4][6];                                       // A^T m, accumulated row by row of m
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            // (M is read exactly once: nontemporal loads, -4 % on this kernel)
            const f32x4 m0 = __builtin_nontemporal_load((const f32x4*)(src + (0 * 6 + j) * plane)), m1 = __builtin_nontemporal_load((const f32x4*)(src + (1 * 6 + j) * plane));
            const f32x4 m2 = __builtin_nontemporal_load((const f32x4*)(src + (2 * 6 + j) * plane)), m3 = __builtin_nontemporal_load((const f32x4*)(src + (3 * 6 + j) * plane));
            const f32x4 m4 = __builtin_nontemporal_load((const f32x4*)(src + (4 * 6 + j) * plane)), m5 = __builtin_nontemporal_load((const f32x4*)(src + (5 * 6 + j) * plane));
            wino4_at(m0, m1, m2, m3, m4, m5, s[0][j], s[1][j], s[2][j], s[3][j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = 4 * ty + i;
            if (yy >= p.H) continue;
            f32x4 y0, y1, y2, y3;
            wino4_at(s[i][0], s[i][1], s[i][2], s[i][3], s[i][4], s[i][5], y0, y1, y2, y3);
            const int xx = 4 * tx;
            auto put = [&](int xo, f32x4 v) {
                if (SIMPLE) {
                    const f32x4 o = afi_epilogue_store_simple(p, n, yy, xo, c, v);
                    if (STATS == 1) afi_stats_acc(st0, st1, smn, smx, o);
                    if (STATS == 2) {
                        const f32x4 cv = __builtin_nontemporal_load((const f32x4*)(p.bstats_c + (((long long)n * p.H + yy) * p.W + xo) * C + c));
                        afi_bstats_acc(st0, st1, o, cv, bmu, bis, bga, bbe, p.bstats_slope);
                    }
                } else afi_epilogue_store(p, n, yy, xo, c, v);
            };
            if (xx < p.W) put(xx, y0);
            if (xx + 1 < p.W) put(xx + 1, y1);
            if (xx + 2 < p.W) put(xx + 2, y2);
            if (xx + 3 < p.W) put(xx + 3, y3);
        }
    }
    if (STATS) afi_stats_block_write(p, st0, st1, smn, smx, STATS == 2 ? p.bstats : nullptr);
}
int afi_launch_wino4_output_epi(const float* M, long long Tpad, const AfiPixGemm& p, hipStream_t st) {
    if (p.N <= 0 || p.H <= 0 || p.W <= 0 || p.Ncols <= 0 || (p.Ncols & 3)) return AFI_ERR_BAD_ARG;
    const int Th = (p.H + 3) / 4, Tw = (p.W + 3) / 4;
    const long long T = (long long)p.N * Th * Tw;
    if (Tpad < T) return AFI_ERR_BAD_ARG;
    if (p.stats || p.bstats) {
        if (!afi_epilogue_is_simple_host(p) || !afi_stats_fusable(p) || (p.stats && p.bstats) || (p.bstats && (!p.bstats_c || !p.bstats_bn.mean))) return AFI_ERR_BAD_ARG;
        AfiPixGemm q = p;
        q.stats_rows = afi_wino_stats_rows(T, p.Ncols);
        if (p.stats) hipLaunchKernelGGL((afi_wino4_output_epi_kernel<true, 1>), dim3(q.stats_rows), dim3(256), 0, st, M, Tpad, Th, Tw, T, q);
        else hipLaunchKernelGGL((afi_wino4_output_epi_kernel<true, 2>), dim3(q.stats_rows), dim3(256), 0, st, M, Tpad, Th, Tw, T, q);
    } else if (afi_epilogue_is_simple_host(p)) hipLaunchKernelGGL((afi_wino4_output_epi_kernel<true>), dim3(wino_grid(T * (p.Ncols >> 2))), dim3(256), 0, st, M, Tpad, Th, Tw, T, p);
    else hipLaunchKernelGGL((afi_wino4_output_epi_kernel<false>), dim3(wino_grid(T * (p.Ncols >> 2))), dim3(256), 0, st, M, Tpad, Th, Tw, T, p);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// weight gradient: Q[a][t][co] = G' e G'^T for the 4x4 block e of dY of tile t
template <int AM>
__global__ __launch_bounds__(256) void afi_wino4_dy_kernel(const AfiView dy, int N, int H, int W, int C, int Th, int Tw, long long T, long long Tpad,
                                                           float* __restrict__ Q, long long ldo, float* amax, const AfiF16Bound bnd) {
    constexpr bool AMAX = AM == 1;
    const float src_max = AM == 2 ? bnd.amax[0] : 0.f;
    const int C4 = C >> 2;
    const long long total = Tpad * C4;
    const long long plane = Tpad * ldo;                    // ldo: row pitch of the plane (>= C: this call may fill a channel slice)
    float am = 0.f;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C4) * 4;
        const long long t = e / C4;
        float* dst = Q + t * ldo + c;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if (t >= T) {
#pragma unroll
            for (int a = 0; a < 36; ++a) *(f32x4*)(dst + a * plane) = zero;
            continue;
        }
        const int tx = (int)(t % Tw); const long long r = t / Tw; const int ty = (int)(r % Th); const int n = (int)(r / Th);
        const float* base = dy.p + (long long)n * dy.sN + c;
        f32x4 a[6][4];                                       // G' e (columns of e)
        const bool interior = 4 * ty + 3 < H && 4 * tx + 3 < W;    // whole 4x4 block inside the map: unconditional loads
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v[4];
            if (interior) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    v[i] = __builtin_nontemporal_load((const f32x4*)(base + (long long)(4 * ty + i) * dy.sH + (long long)(4 * tx + j) * dy.sW));
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int yy = 4 * ty + i, xx = 4 * tx + j;
                    v[i] = (yy < H && xx < W) ? __builtin_nontemporal_load((const f32x4*)(base + (long long)yy * dy.sH + (long long)xx * dy.sW)) : zero;
                }
            }
            if constexpr (AMAX) am = afi_amax4(afi_amax4(afi_amax4(afi_amax4(am, v[0]), v[1]), v[2]), v[3]);
            f32x4 col[6];
            wino4_g4(v[0], v[1], v[2], v[3], col);
#pragma unroll
            for (int r = 0; r < 6; ++r) a[r][j] = col[r];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const f32x4 v0 = a[i][0], v1 = a[i][1], v2 = a[i][2], v3 = a[i][3];
            auto put = [&](int pl, f32x4 v) {
                if (AM == 2) afi_store_split4(Q + t * ldo + pl * plane, c, v, afi_f16_scale(src_max * bnd.cmul[pl]));
                else *(f32x4*)(dst + pl * plane) = v;
            };
            f32x4 row[6];
            wino4_g4(v0, v1, v2, v3, row);
#pragma unroll
            for (int c2 = 0; c2 < 6; ++c2) put(6 * i + c2, row[c2]);
        }
    }
    if constexpr (AMAX) afi_amax_publish(am, amax);
}
int afi_launch_wino4_dy(AfiView dy, int N, int H, int W, int C, long long Tpad, float* Q, hipStream_t st, long long ldo, float* amax, const AfiF16Bound* pre) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return AFI_ERR_BAD_ARG;
    const int Th = (H + 3) / 4, Tw = (W + 3) / 4;
    const long long T = (long long)N * Th * Tw;
    if (Tpad < T) return AFI_ERR_BAD_ARG;
    const long long ld = ldo > 0 ? ldo : (long long)C;
    if (pre) {
        if (!pre->amax || (ld & 31) || (C & 31) || amax) return AFI_ERR_BAD_ARG;
        hipLaunchKernelGGL(afi_wino4_dy_kernel<2>, dim3(wino_grid(Tpad * (C >> 2))), dim3(256), 0, st, dy, N, H, W, C, Th, Tw, T, Tpad, Q, ld, amax, *pre);
    } else if (amax) hipLaunchKernelGGL(afi_wino4_dy_kernel<1>, dim3(wino_grid(Tpad * (C >> 2))), dim3(256), 0, st, dy, N, H, W, C, Th, Tw, T, Tpad, Q, ld, amax, kNoBound);
    else hipLaunchKernelGGL(afi_wino4_dy_kernel<0>, dim3(wino_grid(Tpad * (C >> 2))), dim3(256), 0, st, dy, N, H, W, C, Th, Tw, T, Tpad, Q, ld, amax, kNoBound);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// dW[o][ky][kx][i] += alpha * (A'^T dU A')[ky][kx]  with dU [36][O][I]
__global__ void afi_wino4_dw_kernel(const float* __restrict__ dU, float* __restrict__ dW, int O, int I, float alpha) {
    const long long total = (long long)O * I;
    const long long plane = total;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e % I), o = (int)(e / I);
        float s[3][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float m0 = dU[(0 * 6 + j) * plane + e], m1 = dU[(1 * 6 + j) * plane + e], m2 = dU[(2 * 6 + j) * plane + e];
            const float m3 = dU[(3 * 6 + j) * plane + e], m4 = dU[(4 * 6 + j) * plane + e], m5 = dU[(5 * 6 + j) * plane + e];
            wino4_at3(m0, m1, m2, m3, m4, m5, s[0][j], s[1][j], s[2][j]);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float* dst = dW + ((long long)o * 9 + 3 * k) * I + i;
            float w0, w1, w2;
            wino4_at3(s[k][0], s[k][1], s[k][2], s[k][3], s[k][4], s[k][5], w0, w1, w2);
            dst[0] += alpha * w0;
            dst[I] += alpha * w1;
            dst[2 * I] += alpha * w2;
        }
    }
}
int afi_launch_wino4_dw(const float* dU, float* dW, int O, int I, float alpha, hipStream_t st) {
    if (O <= 0 || I <= 0) return AFI_ERR_BAD_ARG;
    hipLaunchKernelGGL(afi_wino4_dw_kernel, dim3(wino_grid((long long)O * I)), dim3(256), 0, st, dU, dW, O, I, alpha);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
