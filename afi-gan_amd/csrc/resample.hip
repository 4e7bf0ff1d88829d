// Dual-scale data path on the device: the uint8 image resize under the reference's DatasetMapper
// (afigan/engine/dataset_mapper.py:69-193; sizes set by afigan/engine/transform_gen.py:171-217 and :514-559) and the
// normalise + pad of RCNN_FPN_only.forward (afigan/modeling/meta_arch/rcnn_only.py:36-39).
//
// The resize reproduces, bit for bit, what detectron2's ResizeTransform.apply_image runs for uint8 images:
// Pillow's ImagingResample with the BILINEAR filter -- a triangle filter whose support grows with the down-scale factor,
// coefficients computed in double precision and rounded to 22-bit fixed point, a horizontal pass rounded to uint8, then a
// vertical pass.  Byte work, HBM/launch bound: three small kernels (coefficients, horizontal, vertical + flip + CHW store).
// The coefficient kernel runs the same IEEE double operations, in the same order, as the C code it mirrors: contraction into
// FMAs is switched off for this file.
#pragma clang fp contract(off)
#include <math.h>
#include "../../include/afigan_hip.h"
#include "afi_common.h"

#define AFI_RS_PRECISION_BITS 22

__device__ __forceinline__ int afi_rs_clip8(int s) {
    const int v = s >> AFI_RS_PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// one thread per output column (first outW threads) or output row (next outH): bounds[i] = {first source index, tap count},
// kk[i][0..ks) = fixed-point weights (zero beyond the tap count)
__global__ void afi_resample_coeffs_kernel(int inW, int outW, int ksW, int* __restrict__ bW, int* __restrict__ kW,
                                           int inH, int outH, int ksH, int* __restrict__ bH, int* __restrict__ kH) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int in_size, out_size, ks, xx;
    int* b; int* kk;
    if (i < outW) { in_size = inW; out_size = outW; ks = ksW; xx = i; b = bW; kk = kW; }
    else if (i < outW + outH) { in_size = inH; out_size = outH; ks = ksH; xx = i - outW; b = bH; kk = kH; }
    else return;
    const double scale = (double)((float)in_size - 0.0f) / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const double ss = 1.0 / filterscale;
    const double center = 0.0 + (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        double a = (x + xmin - center + 0.5) * ss;
        if (a < 0.0) a = -a;
        ww += a < 1.0 ? 1.0 - a : 0.0;
    }
    int* k = kk + (long long)xx * ks;
    for (int x = 0; x < ks; ++x) {
        double v = 0.0;
        if (x < xmax) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            v = a < 1.0 ? 1.0 - a : 0.0;
            if (ww != 0.0) v = v / ww;
        }
        k[x] = v < 0 ? (int)(-0.5 + v * (double)(1 << AFI_RS_PRECISION_BITS)) : (int)(0.5 + v * (double)(1 << AFI_RS_PRECISION_BITS));
    }
    b[2 * xx] = xmin;
    b[2 * xx + 1] = xmax;
}

// horizontal pass: tmp[y][xx][c] = clip8(sum_x src[y][xmin + x][c] * k[x])
template <int C>
__global__ void afi_resample_h_kernel(const unsigned char* __restrict__ src, int H0, int W0, unsigned char* __restrict__ tmp, int W1,
                                      const int* __restrict__ bounds, const int* __restrict__ kk, int ks) {
    const long long total = (long long)H0 * W1;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(idx / W1), xx = (int)(idx - (long long)y * W1);
        const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
        const int* k = kk + (long long)xx * ks;
        const unsigned char* p = src + ((long long)y * W0 + xmin) * C;
        int s[C];
#pragma unroll
        for (int c = 0; c < C; ++c) s[c] = 1 << (AFI_RS_PRECISION_BITS - 1);
        for (int x = 0; x < xmax; ++x) {
            const int kx = k[x];
#pragma unroll
            for (int c = 0; c < C; ++c) s[c] += (int)p[x * C + c] * kx;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) tmp[idx * C + c] = (unsigned char)afi_rs_clip8(s[c]);
    }
}

// vertical pass + optional horizontal flip + HWC or CHW store
template <int C>
__global__ void afi_resample_v_kernel(const unsigned char* __restrict__ tmp, int W1, unsigned char* __restrict__ dst, int H1,
                                      const int* __restrict__ bounds, const int* __restrict__ kk, int ks, int hflip, int chw) {
    const long long total = (long long)H1 * W1;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int yy = (int)(idx / W1), x = (int)(idx - (long long)yy * W1);
        const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
        const int* k = kk + (long long)yy * ks;
        const unsigned char* p = tmp + ((long long)ymin * W1 + x) * C;
        const long long pitch = (long long)W1 * C;
        int s[C];
#pragma unroll
        for (int c = 0; c < C; ++c) s[c] = 1 << (AFI_RS_PRECISION_BITS - 1);
        for (int y = 0; y < ymax; ++y) {
            const int ky = k[y];
#pragma unroll
            for (int c = 0; c < C; ++c) s[c] += (int)p[y * pitch + c] * ky;
        }
        const int xo = hflip ? W1 - 1 - x : x;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const long long o = chw ? ((long long)c * H1 + yy) * W1 + xo : ((long long)yy * W1 + xo) * C + c;
            dst[o] = (unsigned char)afi_rs_clip8(s[c]);
        }
    }
}

// out[c][y][x] = (img[c][y][x] - mean[c]) / std[c] inside H x W, 0 in the padding up to Hp x Wp   (rcnn_only.py:36-39)
struct AfiNormPrm { float mean[4], std[4]; };
__global__ void afi_normalize_pad_kernel(const unsigned char* __restrict__ img, int C, int H, int W, AfiNormPrm prm,
                                         float* __restrict__ out, int Hp, int Wp) {
    const long long total = (long long)C * Hp * Wp;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % Wp);
        const long long r = idx / Wp;
        const int y = (int)(r % Hp), c = (int)(r / Hp);
        float v = 0.f;
        if (y < H && x < W) v = __fdiv_rn((float)img[((long long)c * H + y) * W + x] - prm.mean[c], prm.std[c]);
        out[idx] = v;
    }
}

static int afi_rs_ksize(int in_size, int out_size) {
    double scale = (double)((float)in_size - 0.0f) / out_size;
    if (scale < 1.0) scale = 1.0;
    return (int)ceil(scale) * 2 + 1;
}
static long long afi_rs_align(long long b) { return (b + 255) & ~255LL; }
struct AfiRsWs { long long o_bw, o_kw, o_bh, o_kh, o_tmp, total; int ksw, ksh; };
static AfiRsWs afi_rs_ws(int H0, int W0, int C, int H1, int W1) {
    AfiRsWs w;
    w.ksw = afi_rs_ksize(W0, W1); w.ksh = afi_rs_ksize(H0, H1);
    long long o = 0;
    w.o_bw = o; o += afi_rs_align(8LL * W1);
    w.o_kw = o; o += afi_rs_align(4LL * W1 * w.ksw);
    w.o_bh = o; o += afi_rs_align(8LL * H1);
    w.o_kh = o; o += afi_rs_align(4LL * H1 * w.ksh);
    w.o_tmp = o; o += afi_rs_align((long long)H0 * W1 * C);
    w.total = o;
    return w;
}
static unsigned afi_rs_grid(long long items) {
    long long g = (items + 255) / 256;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (unsigned)g;
}

long long afi_resize_bilinear_u8_ws_bytes(int H0, int W0, int C, int H1, int W1) {
    if (H0 <= 0 || W0 <= 0 || H1 <= 0 || W1 <= 0 || (C != 1 && C != 3)) return -1;
    return afi_rs_ws(H0, W0, C, H1, W1).total;
}

int afi_resize_bilinear_u8(const unsigned char* src, int H0, int W0, int C, unsigned char* dst, int H1, int W1, int hflip,
                                      int out_chw, void* ws, long long ws_bytes, void* stream) {
    if (!src || !dst || !ws || H0 <= 0 || W0 <= 0 || H1 <= 0 || W1 <= 0) return AFI_ERR_BAD_ARG;
    if (C != 1 && C != 3) return AFI_ERR_UNSUPPORTED;      // Pillow modes "L" and "RGB"; "LA"/"RGBA" resize premultiplied, not this path
    // Pillow keeps the accumulators in 32 bits too; its own limit on the tap count is far beyond any image size used here
    if ((long long)H0 * W0 >= (1LL << 31) || (long long)H1 * W1 >= (1LL << 31)) return AFI_ERR_UNSUPPORTED;
    const AfiRsWs l = afi_rs_ws(H0, W0, C, H1, W1);
    if (ws_bytes < l.total) return AFI_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)ws;
    int* bw = (int*)(base + l.o_bw); int* kw = (int*)(base + l.o_kw);
    int* bh = (int*)(base + l.o_bh); int* kh = (int*)(base + l.o_kh);
    unsigned char* tmp = (unsigned char*)(base + l.o_tmp);
    hipLaunchKernelGGL(afi_resample_coeffs_kernel, dim3((W1 + H1 + 255) / 256), dim3(256), 0, st, W0, W1, l.ksw, bw, kw, H0, H1, l.ksh, bh, kh);
    const unsigned gh = afi_rs_grid((long long)H0 * W1), gv = afi_rs_grid((long long)H1 * W1);
#define AFI_RS_LAUNCH(CC)                                                                                                          \
    hipLaunchKernelGGL((afi_resample_h_kernel<CC>), dim3(gh), dim3(256), 0, st, src, H0, W0, tmp, W1, (const int*)bw, (const int*)kw, l.ksw); \
    hipLaunchKernelGGL((afi_resample_v_kernel<CC>), dim3(gv), dim3(256), 0, st, (const unsigned char*)tmp, W1, dst, H1, (const int*)bh,       \
                       (const int*)kh, l.ksh, hflip, out_chw)
    if (C == 1) { AFI_RS_LAUNCH(1); } else { AFI_RS_LAUNCH(3); }
#undef AFI_RS_LAUNCH
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

int afi_normalize_pad_u8(const unsigned char* img_chw, int C, int H, int W, const float* mean, const float* std_,
                                    float* out, int Hp, int Wp, void* stream) {
    if (!img_chw || !out || !mean || !std_ || C < 1 || H <= 0 || W <= 0 || Hp < H || Wp < W) return AFI_ERR_BAD_ARG;
    if (C > 4) return AFI_ERR_UNSUPPORTED;
    AfiNormPrm prm;
    for (int c = 0; c < 4; ++c) { prm.mean[c] = c < C ? mean[c] : 0.f; prm.std[c] = c < C ? std_[c] : 1.f; }
    hipLaunchKernelGGL(afi_normalize_pad_kernel, dim3(afi_rs_grid((long long)C * Hp * Wp)), dim3(256), 0, (hipStream_t)stream, img_chw, C, H, W,
                       prm, out, Hp, Wp);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
