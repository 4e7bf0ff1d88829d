// Internal shared definitions for the AFI-GAN gfx950 kernels (not part of the public C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define AFI_LRELU_SLOPE 0.2f

// status codes returned across the C-ABI (no exceptions cross it)
#define AFI_OK 0
#define AFI_ERR_BAD_ARG 1
#define AFI_ERR_UNSUPPORTED 2
#define AFI_ERR_LAUNCH 3
#ifndef AFI_TRY
#define AFI_TRY(expr) do { int _s = (expr); if (_s != AFI_OK) return _s; } while (0)
#endif

// A pixel-major ("NHWC") tensor view: element (n, y, x, c) lives at p[n*sN + y*sH + x*sW + c].
// Strides are in elements; the channel stride is always 1.  Cropped views (stage-1 _reshape_stage1)
// and channel slices of wider buffers (RDB dense buffer) are expressed through p and the strides.
struct AfiView {
    float* p;
    long long sN, sH, sW;
};

// A tensor that is read THROUGH a BatchNorm affine and LeakyReLU(0.2): the reader sees lrelu(((x - mean) * invstd) * gamma + beta) per channel
// (csrc/afi_bn.h: the exact arithmetic of the stand-alone apply pass), zero padding stays zero.  mean = null: off.  Taken by the Winograd
// input transforms: a discriminator block's activation is then never written -- its consumers read the saved conv output.
struct AfiBnLoad { const float* mean; const float* invstd; const float* gamma; const float* beta; };

// Parameters of the pixel-M implicit GEMM (forward conv, conv-transpose forward, and both dgrads).
//   C[m][n] = sum_k A[m][k] * B[k][n],  m <-> (img, y, x) on the GEMM pixel grid N x H x W,
//   k <-> (tap, kphase, c),  n <-> output column.
struct AfiPixGemm {
    // GEMM pixel grid (the low-resolution grid for the conv-transpose variants)
    int N, H, W;
    int ntaps;          // 9 (3x3) or 1 (1x1)
    int Ck;             // channels per (tap, kphase) on the K side
    int nKphase;        // 1, or 4 when the A side is a pixel-shuffled hi-res tensor (convT dgrad)
    int Ncols;          // GEMM N
    // A gather: pixel (y + a_sgn*dy, x + a_sgn*dx) on the grid, scaled by a_up (+ phase) in the tensor
    AfiView A;
    int a_sgn;          // +1 forward, -1 dgrad
    int a_up;           // 1, or 2 for convT dgrad (A is [N, 2H, 2W])
    // generic-tap form (gtap = 1; stride-2 conv forward and the four parity phases of its dgrad): tap t reads grid position
    // (y*a_stride + tap_dy[t], x*a_stride + tap_dx[t]), valid inside [0, aH) x [0, aW), with weight tap tap_w[t]
    int gtap, a_stride, aH, aW;
    signed char tap_dy[12], tap_dx[12], tap_w[12];
    // B (weights).  KC form (b_rc = 0): row n at  B + n*b_sRow + tap*b_sTap + c           (k contiguous)
    //               RC form (b_rc = 1): row k at  B + (kphase*Ck + c)*b_sRow + tap*b_sTap + n  (n contiguous)
    const float* B;
    long long b_sRow, b_sTap;
    int n_fastest;      // tile walk: 0 = M fastest per N tile (weight panel stationary in L2), 1 = N fastest per M tile (A tile stationary)
    long long b_sImg;   // weight stride per GEMM "image" (Winograd: one weight matrix per transform point); 0 = shared weights
    // output: column col -> phase = col / CoutPhase, channel = col % CoutPhase;
    // pixel (y*o_up + (phase>>1), x*o_up + (phase&1))
    AfiView O;
    int o_up;           // 1, or 2 for convT forward (O is [N, 2H, 2W])
    int oH, oW;         // rows whose output pixel falls outside [0, oH) x [0, oW) are not stored (odd sizes in the stride-2 dgrad)
    int CoutPhase;      // == Ncols when o_up == 1
    // epilogue: v = alpha*acc + bias[ch] + beta*O_old + r1s*R1 + r2s*R2 ; lrelu ; * lrelu'(Z)
    //   with r2_post: v = post_scale * act(alpha*acc + bias + ...) [-> O2 if set] + r2s*R2   (PAFPN: inter + relu(conv))
    float alpha, beta;
    const float* bias;  // indexed by channel (col % CoutPhase); may be null
    AfiView R1; float r1s; int r1_lo, r1_hi;   // applied for channels in [r1_lo, r1_hi); null p = off
    AfiView R2; float r2s; int r2_lo, r2_hi;
    int r2_post; float post_scale; int pad0_; AfiView O2;
    int r1_bilinear;    // R1 is a low-res [N, H/2, W/2] tensor, added as its bilinear x2 up-sampling
    int lrelu;          // activation on v: 0 none, 1 LeakyReLU(0.2), 2 ReLU
    AfiView Z; int z_lo, z_hi;                 // multiply by (Z > 0 ? 1 : 0.2) for channels in [z_lo, z_hi)
    // split-K scratch (optional): [splitK][M][roundup4(Ncols)] partial slabs; the launcher picks splitK and fills it in
    float* partial; long long partial_floats; int splitK; int kper;   // kper: K stages per split (set by the launcher)
    // BatchNorm batch statistics of the STORED output, accumulated by the Winograd output transform itself (a separate pass over the
    // map otherwise): fp64 partial sums [stats_rows][2][Ncols] (sum, sum of squares per channel), one row per block of that launch; the
    // launcher fills stats_rows.  Taken only by the plain-store epilogue on 256 / 512 / 1024-channel outputs; null = off.
    double* stats; int stats_rows;
    // ... and, beside them, the per-channel MINIMUM and MAXIMUM of the stored output as fp32 rows [stats_rows][2][Ncols] (min, max), one row per
    // block: from them the statistics finalizer derives the largest magnitude of the block's ACTIVATION lrelu(affine(c)) -- the affine is
    // monotonic per channel, so it is attained at one of the two -- before any kernel has evaluated it (AFI_OPT_D_FOLD_BN_APPLY).  Null = off.
    float* stats_mm;
    // The BatchNorm BACKWARD sums of the stored output, for an output that is the gradient with respect to a discriminator block's ACTIVATION (the
    // data gradient of the next block): fp64 rows [stats_rows][2][Ncols] of sum g m and sum g m xhat per channel, m the LeakyReLU' factor and
    // xhat the normalised value, both recomputed from the block's saved conv output bstats_c (dense [pixels][Ncols], the output's geometry)
    // through bstats_bn (csrc/afi_bn.h) -- the pass afi_colred_partial_kernel<3> makes over both tensors otherwise.  Same shapes as `stats`; null = off.
    double* bstats; const float* bstats_c; AfiBnLoad bstats_bn; float bstats_slope;
    int no_wcache;                             // Winograd form: B is a per-call scratch (its pointer says nothing about its contents): never cache its transform
    // Small-map bf16x6 form (csrc/smallmap.hip, afi_pix_gemm_wk6): the weights pre-split into bf16 MFMA-fragment images,
    // [N tile of 32][K stage of 32][n half][hi | mid | lo][lane] x 16 B, K stages in the kernel's own order (channel chunk, K phase, tap);
    // null = the fp32-MFMA kernel reads B itself.  A problem on input channels [c_lo, c_lo + Ck) of a wider weight starts at stage
    // bimg_stage0 = c_lo / 32 * ntaps of that weight's image; bimg_nstages = stages per N tile of the whole image.
    const unsigned char* Bimg; int bimg_stage0, bimg_nstages;
    AfiBnLoad a_bn;                                        // Winograd form only: A is read through this affine + LeakyReLU (every other form refuses it)
    // Winograd form under the f16x3 arithmetic only: the largest magnitude of A (device memory).  a_amax_known = 1: its producer has published it
    // (the input transform then writes the planes split into fp16 pieces); 0: a zero-filled slot the input transform raises; null: a slot of the call's pool
    float* a_amax; int a_amax_known;
    // Winograd form: caller-owned home for the input planes (at least 36 * Tpad * Ck floats) instead of the call's scratch, honoured only when
    // the planes are F(4x4) and written split into fp16 pieces -- the form the weight-gradient GEMM of the same conv reads, so a backward
    // pass can take them instead of transforming the input again (nets.hip: disc_v_shared decides on both sides)
    float* v_keep;
    int nt_local_sums;                                        // Winograd form under f16x3: the NT GEMM sums each k-step in a fresh fragment (AFI_OPT_F16_LOCAL_SUMS)
};
#define AFI_WK6_STAGE_BYTES 6144
// one weight (or weight view) to turn into such an image: the B addressing of AfiPixGemm (b_rc = 0: row n at B + n*b_sRow + tap*b_sTap + c;
// b_rc = 1: row (kphase*Ck + c) at B + ...*b_sRow + tap*b_sTap + n), image bytes = ceil(Ncols/32) * ceil(Ck/32)*nKphase*ntaps * 6144
// stage_off / nstages_img (0 = this job's own stage count): the job fills stages [stage_off, stage_off + its stages) of an image whose N tiles
// are nstages_img stages apart -- several weights side by side along K in ONE image (the dense block's four growth convs as one data gradient)
struct AfiWk6ImgJob { const float* B; long long b_sRow, b_sTap; int Ncols, Ck, ntaps, nKphase, b_rc, stage_off; unsigned char* dst; int nstages_img, pad1; };
// Work of a small-map backward pass that depends on nothing the pass computes and rides in its image launch as extra blocks (a launch of its
// own costs each of them more than the work): zero_p[0 .. 4 zero_n4) = 0 (the packed gradient buffers the stream-K weight gradients add into),
// and bl_dx[N,H,W,C] = bilinear2x^T(bl_dout[N,2H,2W,C]) (the skip path's gradient, generator_rdb.py:125).  Null pointers: off.
struct AfiWk6Side { float* zero_p; long long zero_n4; const float* bl_dout; float* bl_dx; int bl_N, bl_H, bl_W, bl_C; };
// The conv-transpose weight W [Cin][Cout][6][6] (generator_rdb.py:101-105) STRAIGHT into a bf16x6 image, with no packed fp32 copy in between
// (that copy was a launch of its own in front of the image launch): mode 0 the forward's image -- columns (phase, co), K = ci: the image of
// Wp [(phase*Cout + co)][tap][ci] --, mode 1 the data gradient's -- columns ci, K = (32-channel chunk of co, phase, tap).  Through an LDS
// tile (32 ci x 4 co, or 16 ci x 8 co, x 36 taps) so that reads and stores stay contiguous.  pack_dst (optional): the packed fp32 form Wp is
// written as well, by further blocks of the same launch (for the callers that still read it).  Cin, Cout multiples of 32.
struct AfiWk6ConvT { const float* W; unsigned char* dst; float* pack_dst; int Cin, Cout, mode, pad_; };

// Fused growth-conv chain of one dense block on a small map (csrc/smallmap.hip: afi_rdb_chain6_kernel): three dependent 3x3 convs with
// 32-channel outputs -- y2, y3, y4 of ResidualDenseBlock.forward, or the data gradients g3, g2, g1 of its backward -- in ONE launch.  A block
// owns an 8 x 8 pixel tile, keeps region 0 (the chain's 32-channel input: y1 / g4) with a 3-pixel halo in LDS and recomputes each link on
// a halo that shrinks by one (12 x 12, 10 x 10, 8 x 8): phase p multiplies regions 0..p (K chunks of 32 channels, 9 taps each, weights
// from bf16x6 images) into the next region, adds `partial`, applies LeakyReLU (mode 0) or the LeakyReLU' factor of Z (mode 1), zeroes what
// lies outside the map (the convs' zero padding), and the owner stores its own 8 x 8 pixels to `out`.
struct AfiChain6Phase {
    const unsigned char* img[3]; int stage0[3]; int pad_;    // K chunk ci (region ci): weight-image stages [stage0, stage0 + 9) of an N tile of 32
    AfiView partial, Z, out;                                 // 32-channel views on the map (p points at the slice's first channel)
};
struct AfiChain6 {
    int N, H, W, a_sgn, mode, tiles_y, tiles_x, pad_;
    AfiView src0, copy0;                                     // region 0's source; optional copy of its own pixels (null p: off)
    AfiChain6Phase ph[3];
};

// Parameters of the weight-gradient GEMM:  dW[co'][tap][ci] += alpha * sum_pix dY[pix][co'] * X[pix+tap][ci]
struct AfiWgradGemm {
    int N, H, W;        // pixel grid (low-res for convT)
    int ntaps;
    int Mrows;          // co' count (4*Cout for convT)
    int Ncols;          // ci count
    AfiView DY; int dy_up; int CoutPhase;      // dY gather (pixel-shuffled when dy_up == 2)
    AfiView X;                                  // input activations, gathered at (y*x_stride+dy, x*x_stride+dx) inside [0,xH) x [0,xW)
    int x_stride, xH, xW;
    float* DW; long long dw_sRow, dw_sTap;      // dW + co'*dw_sRow + tap*dw_sTap + ci
    long long dy_sTap, x_sTap;                  // ntaps == 16 (Winograd transform points): operand planes, DY + tap*dy_sTap, X + tap*x_sTap
    float alpha;
    int splitK;
};

// f16x3 arithmetic of the batched Winograd GEMMs (afi_gemm_f16.h): where an operand's power-of-two scale comes from.  amax[plane * stride]
// (device memory, written by the kernel that produced the operand) times cmul[plane] bounds the plane's largest magnitude.  stride = 0: one
// value for all planes (the largest magnitude of the tensor the planes are a transform of).
// interpolation points of the F(4x4, 3x3) / F(3x3, 4x4) transforms (csrc/winograd.hip): 1 = {0, 1, -1, 1/2, -2, inf} (the product), 0 = the
// textbook {0, +-1, +-2, inf} (A/B builds).  One definition: the transforms AND the plane bounds of afi_f16_bound (igemm.hip) follow it.
#ifndef AFI_WINO4_POINTS
#define AFI_WINO4_POINTS 1
#endif
struct AfiF16Bound { const float* amax; int stride; int pad_; float cmul[36]; };

// one bias-gradient problem of afi_launch_colsum_group: db[c] += alpha * sum_rows g[row*ld + c], c < C
struct AfiColsumProb { const float* g; float* db; long long P, ld; int C; float alpha; };

static inline int afi_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
