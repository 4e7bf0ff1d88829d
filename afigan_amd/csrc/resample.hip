// Dual-scale data path on the device: the uint8 image resize under the reference's DatasetMapper
// (afigan/engine/dataset_mapper.py:69-193; sizes set by afigan/engine/transform_gen.py:171-217 and :514-559) and the
// normalise + pad of RCNN_FPN_only.forward (afigan/modeling/meta_arch/rcnn_only.py:36-39).
//
// The resize reproduces, bit for bit, what detectron2's ResizeTransform.apply_image runs for uint8 images:
// Pillow's ImagingResample with the BILINEAR filter -- a triangle filter whose support grows with the down-scale factor,
// coefficients computed in double precision and rounded to 22-bit fixed point, a horizontal pass rounded to uint8, then a
// vertical pass.  Byte work, HBM/launch bound: two launches per SAMPLE -- the coefficient tables of all four axes, then one fused
// kernel that produces `image` and `image_x0.5` from the source (both passes, flip and CHW store; no intermediate image in memory).
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

// Coefficient tables of up to four axes in one launch.  One thread per output index of an axis:
// bounds[i] = {first source index, tap count}, kk[i][0..ks) = fixed-point weights (zero beyond the tap count).
struct AfiRsAxis { int in_size, out_size, ks; int* bounds; int* kk; };
struct AfiRsAxes { AfiRsAxis a[4]; int n; };
__global__ void afi_resample_coeffs_kernel(AfiRsAxes ax) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int which = 0;
    while (which < ax.n && i >= ax.a[which].out_size) { i -= ax.a[which].out_size; ++which; }
    if (which >= ax.n) return;
    const int in_size = ax.a[which].in_size, out_size = ax.a[which].out_size, ks = ax.a[which].ks, xx = i;
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
    int* k = ax.a[which].kk + (long long)xx * ks;
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
    ax.a[which].bounds[2 * xx] = xmin;
    ax.a[which].bounds[2 * xx + 1] = xmax;
}

// Both passes in one kernel, for up to two target sizes of one source image.  Per output pixel the few intermediate pixels its
// vertical taps need are recomputed from the source -- each rounded to uint8 exactly as Pillow's horizontal pass stores them --
// so the intermediate image never exists in memory: the source (L2-resident) is read, the result written once.
// Optional horizontal flip and HWC / CHW store are part of the same write.
struct AfiRsTarget { unsigned char* dst; int H1, W1, ksw, ksh, hflip; const int* bw; const int* kw; const int* bh; const int* kh; };
struct AfiRsTargets { AfiRsTarget t[2]; int n; };
template <int C>
__global__ void afi_resample_fused_kernel(const unsigned char* __restrict__ src, int W0, AfiRsTargets tg, int chw) {
    const long long n0 = (long long)tg.t[0].H1 * tg.t[0].W1;
    const long long total = n0 + (tg.n > 1 ? (long long)tg.t[1].H1 * tg.t[1].W1 : 0);
    for (long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x; gi < total; gi += (long long)gridDim.x * blockDim.x) {
        const int w = gi >= n0;
        const AfiRsTarget& t = tg.t[w];
        const long long idx = gi - (w ? n0 : 0);
        const int yy = (int)(idx / t.W1), xx = (int)(idx - (long long)yy * t.W1);
        const int xmin = t.bw[2 * xx], xmax = t.bw[2 * xx + 1];
        const int ymin = t.bh[2 * yy], ymax = t.bh[2 * yy + 1];
        const int* kx = t.kw + (long long)xx * t.ksw;
        const int* ky = t.kh + (long long)yy * t.ksh;
        int s[C];
#pragma unroll
        for (int c = 0; c < C; ++c) s[c] = 1 << (AFI_RS_PRECISION_BITS - 1);
        for (int y = 0; y < ymax; ++y) {
            const unsigned char* p = src + ((long long)(ymin + y) * W0 + xmin) * C;
            int h[C];
#pragma unroll
            for (int c = 0; c < C; ++c) h[c] = 1 << (AFI_RS_PRECISION_BITS - 1);
            for (int x = 0; x < xmax; ++x) {
                const int k = kx[x];
#pragma unroll
                for (int c = 0; c < C; ++c) h[c] += (int)p[x * C + c] * k;
            }
            const int k = ky[y];
#pragma unroll
            for (int c = 0; c < C; ++c) s[c] += afi_rs_clip8(h[c]) * k;
        }
        const int xo = t.hflip ? t.W1 - 1 - xx : xx;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const long long o = chw ? ((long long)c * t.H1 + yy) * t.W1 + xo : ((long long)yy * t.W1 + xo) * C + c;
            t.dst[o] = (unsigned char)afi_rs_clip8(s[c]);
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
// workspace of one target: [bounds W][kk W][bounds H][kk H]
struct AfiRsWs { long long o_bw, o_kw, o_bh, o_kh, total; int ksw, ksh; };
static AfiRsWs afi_rs_ws(int H0, int W0, int H1, int W1) {
    AfiRsWs w;
    w.ksw = afi_rs_ksize(W0, W1); w.ksh = afi_rs_ksize(H0, H1);
    long long o = 0;
    w.o_bw = o; o += afi_rs_align(8LL * W1);
    w.o_kw = o; o += afi_rs_align(4LL * W1 * w.ksw);
    w.o_bh = o; o += afi_rs_align(8LL * H1);
    w.o_kh = o; o += afi_rs_align(4LL * H1 * w.ksh);
    w.total = o;
    return w;
}
static unsigned afi_rs_grid(long long items) {
    long long g = (items + 255) / 256;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (unsigned)g;
}
static bool afi_rs_shape_ok(int H0, int W0, int H1, int W1) {
    // Pillow keeps its accumulators in 32 bits too; its own limit on the tap count is far beyond any image size used here
    return H0 > 0 && W0 > 0 && H1 > 0 && W1 > 0 && (long long)H0 * W0 < (1LL << 31) && (long long)H1 * W1 < (1LL << 31);
}
// n = 1 or 2 targets of one source
static int afi_rs_run(const unsigned char* src, int H0, int W0, int C, int n, unsigned char* const dst[2], const int H1[2], const int W1[2],
                      const int hflip[2], int out_chw, void* ws, long long ws_bytes, hipStream_t st) {
    if (!src || !ws) return AFI_ERR_BAD_ARG;
    if (C != 1 && C != 3) return AFI_ERR_UNSUPPORTED;      // Pillow modes "L" and "RGB"; "LA"/"RGBA" resize premultiplied, not this path
    AfiRsAxes ax; ax.n = 2 * n;
    AfiRsTargets tg; tg.n = n;
    long long off = 0, outs = 0, pix = 0;
    for (int i = 0; i < n; ++i) {
        if (!dst[i]) return AFI_ERR_BAD_ARG;
        if (!afi_rs_shape_ok(H0, W0, H1[i], W1[i])) return AFI_ERR_UNSUPPORTED;
        const AfiRsWs l = afi_rs_ws(H0, W0, H1[i], W1[i]);
        if (ws_bytes < off + l.total) return AFI_ERR_WORKSPACE;
        char* base = (char*)ws + off;
        int* bw = (int*)(base + l.o_bw); int* kw = (int*)(base + l.o_kw);
        int* bh = (int*)(base + l.o_bh); int* kh = (int*)(base + l.o_kh);
        ax.a[2 * i] = AfiRsAxis{W0, W1[i], l.ksw, bw, kw};
        ax.a[2 * i + 1] = AfiRsAxis{H0, H1[i], l.ksh, bh, kh};
        tg.t[i] = AfiRsTarget{dst[i], H1[i], W1[i], l.ksw, l.ksh, hflip[i], bw, kw, bh, kh};
        off += l.total; outs += W1[i] + H1[i]; pix += (long long)H1[i] * W1[i];
    }
    for (int i = n; i < 2; ++i) tg.t[i] = tg.t[0];
    for (int i = 2 * n; i < 4; ++i) ax.a[i] = ax.a[0];
    hipLaunchKernelGGL(afi_resample_coeffs_kernel, dim3((unsigned)((outs + 255) / 256)), dim3(256), 0, st, ax);
    if (C == 1) hipLaunchKernelGGL((afi_resample_fused_kernel<1>), dim3(afi_rs_grid(pix)), dim3(256), 0, st, src, W0, tg, out_chw);
    else        hipLaunchKernelGGL((afi_resample_fused_kernel<3>), dim3(afi_rs_grid(pix)), dim3(256), 0, st, src, W0, tg, out_chw);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

long long afi_resize_bilinear_u8_ws_bytes(int H0, int W0, int C, int H1, int W1) {
    if (!afi_rs_shape_ok(H0, W0, H1, W1) || (C != 1 && C != 3)) return -1;
    return afi_rs_ws(H0, W0, H1, W1).total;
}

int afi_resize_bilinear_u8(const unsigned char* src, int H0, int W0, int C, unsigned char* dst, int H1, int W1, int hflip,
                           int out_chw, void* ws, long long ws_bytes, void* stream) {
    unsigned char* const d[2] = {dst, nullptr};
    const int h[2] = {H1, 0}, w[2] = {W1, 0}, f[2] = {hflip, 0};
    return afi_rs_run(src, H0, W0, C, 1, d, h, w, f, out_chw, ws, ws_bytes, (hipStream_t)stream);
}

long long afi_dual_scale_u8_ws_bytes(int H0, int W0, int C, int H1, int W1, int H2, int W2) {
    const long long a = afi_resize_bilinear_u8_ws_bytes(H0, W0, C, H1, W1), b = afi_resize_bilinear_u8_ws_bytes(H0, W0, C, H2, W2);
    return a < 0 || b < 0 ? -1 : a + b;
}

int afi_dual_scale_u8(const unsigned char* src, int H0, int W0, int C, unsigned char* image, int H1, int W1, int hflip,
                      unsigned char* image_r, int H2, int W2, int hflip_r, int out_chw, void* ws, long long ws_bytes, void* stream) {
    unsigned char* const d[2] = {image, image_r};
    const int h[2] = {H1, H2}, w[2] = {W1, W2}, f[2] = {hflip, hflip_r};
    return afi_rs_run(src, H0, W0, C, 2, d, h, w, f, out_chw, ws, ws_bytes, (hipStream_t)stream);
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
