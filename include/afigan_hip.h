/*
 * afigan_hip.h -- C-ABI of the MI355X-native AFI-GAN hot path (libafigan_hip.so, gfx950).
 *
 * The reference (inhavl-shlee/AFI-GAN) is pure Python: its "plugin API" for this path is two nn.Module
 * classes,  Generator (afigan/modeling/feat_interpol/generator_rdb.py:73-130)  and
 * Discriminator (afigan/modeling/feat_interpol/feature_patch_discriminator.py:16-55),  plus the loss /
 * optimizer calls of the stage-1 loop (afigan/engine/stage1_trainer.py:305-443).  It has no FFI of its own,
 * so this header defines the boundary a binding for that path needs: plain pointers, sizes and a HIP stream;
 * no torch types; every function returns an int status (AFI_OK == 0) and never throws or allocates.
 * The Python binding a maintainer would add (ctypes) is shown in INTEGRATION.md and shipped in
 * afigan_amd/_lib.py.
 *
 * Conventions
 *   - Activations are PIXEL-MAJOR ("NHWC"): element (n, y, x, c) of a view lives at
 *     p[n*sN + y*sH + x*sW + c]; strides are in floats, the channel stride is 1.  A torch tensor in
 *     torch.channels_last memory format is exactly this (zero copy); cropped views
 *     (stage1_trainer.py:437-443) and channel slices of wider buffers are expressed through p / strides.
 *   - 3x3 weights are  [Cout][3][3][Cin]  (the physical layout of a [Cout,Cin,3,3] tensor in channels_last
 *     format, so the reference's state_dict shapes are unchanged); the ConvTranspose2d weight stays in torch's
 *     [Cin][Cout][6][6] layout.  All data is fp32 in memory.  Matrix math: the Winograd-domain GEMMs of the big 3x3 convs in the context's arithmetic
 *     (afi_ctx_set_compute_dtype: by default fp32 products formed exactly on the bf16 matrix cores), every other GEMM on v_mfma_f32_32x32x2_f32 (exact f32).
 *   - Channel counts must be multiples of 4 (float4 granularity); spatial sizes are arbitrary.
 *   - "accumulate" outputs (all gradients w.r.t. parameters) are += targets: zero them first, like
 *     optimizer.zero_grad() at stage1_trainer.py:374/426.
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).
 *   - `ctx` (first argument of the convolution and whole-network entry points) is the caller's afi_ctx_t or NULL, see below.
 */
#ifndef AFIGAN_HIP_H
#define AFIGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AFI_OK 0
#define AFI_ERR_BAD_ARG 1
#define AFI_ERR_UNSUPPORTED 2
#define AFI_ERR_LAUNCH 3
#define AFI_ERR_WORKSPACE 4

#define AFI_MAX_RDB 8

typedef struct afi_view {
    float* p;
    long long sN, sH, sW;
} afi_view_t;

int afi_abi_version(void);
const char* afi_status_string(int status);

/* ------------------------------------------------------------------ caller-owned context
 * Every piece of library state that outlives one call lives in an afi_ctx_t the caller creates, passes to the entry points that can
 * use it, and destroys: the split-K scratch of the per-op calls, the cache of transformed / packed weights, the transform-domain
 * weight-gradient accumulator, and the side stream the backward passes fork their weight gradients onto.  NOTHING is process-global
 * (the reference's modules keep their state per nn.Module instance; SURVEY 8b: "thread-safe per stream; no global mutable state").
 *   - One context serves ONE stream at a time on the device that was current when it was created (calls on another device return
 *     AFI_ERR_BAD_ARG); two engines / threads use two contexts and share nothing.
 *   - ctx may be NULL in every entry point: no scratch, no caches, no side stream (everything on `stream`, every transform per call).
 *   - The library never allocates device memory: the three buffers are the caller's and must stay alive and unchanged in meaning while
 *     registered (floats == 0 unregisters).  The side stream and its events are created on first use and released by afi_ctx_destroy. */
typedef struct afi_ctx afi_ctx_t;
/* sha256 (hex) of the sources the binary was compiled from -- every *.hip / *.h under afigan_amd/csrc and this header, in sorted order of
 * their names -- written into the build by __graft_entry__.build(); "unknown" for a build that bypassed it.  afigan_amd/_lib.py recomputes it
 * from the tree and refuses a library that was built from other sources: "which binary ran" has one answer. */
const char* afi_build_id(void);
int afi_ctx_create(afi_ctx_t** out);
int afi_ctx_destroy(afi_ctx_t* ctx);                       /* refused (AFI_ERR_BAD_ARG) while weight-gradient sums are pending */
/* Scratch for the PER-OP convolution entry points (afi_conv3x3_*, afi_conv1x1_*, afi_conv3x3s2_*, afi_convT6s2_*): with it, mid-size maps
 * (< 6 tiles of 128x128 per CU) run split-K with a deterministic second pass.  256 MB covers every shape that is split.  (The whole-net
 * entry points carve their split-K scratch out of the workspace they are given.) */
int afi_ctx_set_op_scratch(afi_ctx_t* ctx, float* scratch, long long floats);
/* Arithmetic of the Winograd-domain GEMMs issued under this context -- every 3x3 / stride-1 convolution with >= 128 channels on both
 * sides and >= 1024 pixels, forward, data gradient and weight gradient, i.e. >= 95 % of a stage-1 step's FLOPs.  Tensors, transforms,
 * epilogues, accumulators and every other kernel are fp32 under every setting; the settings differ in how a product of two fp32
 * operands is formed on the matrix cores (gfx950 runs fp32 MFMAs at 1/16 of its bf16 MFMA rate, and has no xf32).  Measured max-norm
 * error of the batched GEMM against fp64 on the same operands (tools/gemm_nt_dtype.py; tolerances asserted in tests/test_gpu_bf16.py):
 *   AFI_DTYPE_BF16X6  (default) each operand is split EXACTLY into three bf16, x = hi + mid + lo (3 x 8 = all 24 mantissa bits; the
 *                     residuals x - hi and x - hi - mid are exact in fp32), and the six partial products of order >= 2^-16
 *                     (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi) are accumulated smallest first in fp32 by the bf16 MFMA (16x16x32 / 32x32x16);
 *                     a bf16 x bf16 product is exact in fp32, and what is dropped (mid*lo, lo*mid, lo*lo <= 2^-23 of a product) is the
 *                     size of fp32's own rounding of that product.  Error 0.5e-6 .. 1.1e-6 -- at or below the fp32 MFMA's on every shape
 *                     measured -- at 1.5x its speed.  fp32-grade: every parity test of this repo runs on it at the fp32 tolerances.
 *   AFI_DTYPE_F16X3   each operand is scaled by a power of two (per operand and Winograd plane, exact) and split into TWO fp16 pieces,
 *                     x s = hi + lo + r with |r| <= 2^-23 |x s| (round to nearest; 11 + 11 significant bits, the residual x s - hi exact in
 *                     fp32); the three products hi*hi, hi*lo, lo*hi are accumulated smallest first in fp32 by the f16 MFMA (same rate as
 *                     the bf16 one), the dropped lo*lo is <= 2^-22 of a product with random sign, and the accumulator is scaled back by
 *                     the exact inverse in the epilogue.  Half the matrix-core work of BF16X6.  The scale needs the operand's largest
 *                     magnitude before the GEMM starts: the Winograd transforms that write the planes compute the largest magnitude of
 *                     the tensor they read as a by-product, and every plane is bounded by a constant times it (csrc/afi_gemm_f16.h);
 *                     elements more than ~2^18 below the plane's bound lose RELATIVE precision (their absolute error stays <= 2^-40 of
 *                     the bound), which a max-norm comparison with fp32 does not see and a dot product does not feel.  fp32-grade on
 *                     the operands of tests/test_gpu_bf16.py::test_f16x3_is_fp32_grade_on_hard_operands.
 *   AFI_DTYPE_F32     fp32 MFMA (v_mfma_f32_32x32x2_f32, bit-identical to an fmaf chain).  Error 1.0e-6 .. 1.3e-6.
 *   AFI_DTYPE_BF16X3  x = hi + lo (16 mantissa bits), three MFMAs (hi*hi + hi*lo + lo*hi).  Error 4.5e-6; conv outputs within 2e-4,
 *                     network outputs / input gradients within 1e-3, weight gradients within 1e-2 (L2; LeakyReLU-mask flips).  Opt-in.
 *   AFI_DTYPE_BF16    operands rounded to bf16 (8 bits), one MFMA; every convolution on F(2x2,3x3) tiles (the F(4x4) transforms do not
 *                     survive 8-bit operands: 3 % error).  Error 2.5e-3; conv outputs within 2e-2, networks 5e-2 / 1e-1.  Opt-in:
 *                     the reference has no reduced-precision mode (SOLVER.AMP is never read, defaults.py:82).
 * What follows the setting: the three batched GEMMs of a Winograd convolution -- forward (NT), data gradient (NT on the flipped weights)
 * and weight gradient (TN) -- so a backward pass runs in the arithmetic its forward ran in (the autograd wrappers carry the forward's context
 * into backward).  What never does: the direct implicit-GEMM kernels (1x1 and lateral convs, stride-2 and transposed convs, ragged / small
 * maps, the small-map grouped kernels) and their weight-gradient GEMMs multiply on v_mfma_f32_32x32x2_f32 under every setting.
 * Refused while weight-gradient sums are pending.  No environment variable changes the default. */
#define AFI_DTYPE_F32 0
#define AFI_DTYPE_BF16 1
#define AFI_DTYPE_F16X3 2
#define AFI_DTYPE_BF16X3 3
#define AFI_DTYPE_BF16X6 6
#define AFI_DTYPE_DEFAULT AFI_DTYPE_F16X3
int afi_ctx_set_compute_dtype(afi_ctx_t* ctx, int dtype);
int afi_ctx_get_compute_dtype(const afi_ctx_t* ctx);
/* Algorithm options of a context.  The library reads NO environment variable: every choice that changes numerics or scheduling is made
 * here, per context, and may be changed between calls (refused while weight-gradient sums are pending).  ctx == NULL reads the default. */
#define AFI_OPT_WINOGRAD 0                    /* 1 (default): 3x3 / stride-1 convs with >= 128 channels on both sides run in Winograd form from the pixel
                                                 thresholds below; 0: direct implicit GEMM on the fp32 MFMA everywhere */
#define AFI_OPT_WINOGRAD_F4_BACKWARD 1        /* 1 (default): F(4x4,3x3) / F(3x3,4x4) for data and weight gradients and for forwards no backward follows,
                                                 from 8192 pixels; 0: F(2x2,3x3) everywhere */
#define AFI_OPT_WINOGRAD_F4_FORWARD 2         /* the discriminator forwards a backward follows (their conv outputs decide LeakyReLU masks), per block:
                                                 bit n + 1 puts block n on F(4x4) (2: block 0, 4: block 1, 8: block 2; sums combine); 1: every block;
                                                 0: F(2x2) everywhere.  Default 12 (blocks 1 and 2, with AFI_OPT_F16_LOCAL_SUMS = 12): the largest block set whose gradient
                                                 deviation from fp64 stays below torch's own fp32 ops on the same inputs at P2 and at P3
                                                 (profiles/r06/dflip_p2_*_6seeds_local_sums.txt, dflip_p3_*: D fwd+bwd, relative L2 of dx / worst parameter gradient, means
                                                 over six seeds; P3 | P2): torch fp32 1.15e-3 / 1.54e-3 | 1.21e-3 / 1.54e-3; = 0: 0.79e-3 / 1.06e-3 | 0.88e-3 / 1.18e-3;
                                                 = 8: 0.97e-3 / 1.31e-3 | 1.14e-3 / 1.51e-3; = 12 in the plain summation order (round 5's default): 1.27e-3 / 1.73e-3 |
                                                 1.45e-3 / 1.91e-3; = 12 with the local sums: 0.88e-3 / 1.26e-3 | 1.04e-3 / 1.33e-3; = 1 with them: 1.32e-3 / 1.79e-3 |
                                                 1.40e-3 / 1.76e-3.  tests/test_gpu_d_parity.py holds the default to that bar.
                                                 Bit 16: the interpolator's own forwards too (off: 15x the deviation on its worst parameter gradient for 1 ms) */
#define AFI_OPT_BN_STATS_FP64 3               /* 1 (default): BatchNorm batch statistics accumulated in fp64 (torch's CPU accumulation type) */
#define AFI_OPT_D_WINOGRAD_MIN_PIXELS 4       /* 1024: discriminator calls of fewer pixels stay direct (values below 1024 act as 1024) */
#define AFI_OPT_G_WINOGRAD_MIN_PIXELS 5       /* 2048: the same for a convolution of the interpolator */
#define AFI_OPT_G_SMALLMAP_MAX_PIXELS 6       /* 2048: below, the dense blocks run in column-batched form (5 grouped launches per block); 0: never */
#define AFI_OPT_G_GROUPED_WGRAD_MAX_PIXELS 7  /* 3000: up to here all weight / bias gradients of a backward pass run as three grouped launches; 0: per layer */
#define AFI_OPT_G_BATCH_GROWTH_GRADS 8         /* 1 (default): above G_GROUPED_WGRAD_MAX_PIXELS the four growth convs of a dense block are batched where they share
                                               * an operand: what they take from the block input is ONE conv C -> 4G on packed weights [4G][3][3][C] (forward) and ONE
                                               * data gradient 4G -> C (backward), their weight gradients ONE 4G-row GEMM (packed, then unpacked) -- all Winograd-
                                               * eligible at the reference's widths; 0: four G-column / G-row GEMMs per block, as generator_rdb.py:64-71 reads */
#define AFI_OPT_G_SMALLMAP6_MAX_PIXELS 9      /* 4096: under AFI_DTYPE_BF16X6 (and channel counts that are multiples of 32) interpolator calls of up to this many
                                               * low-res pixels run the small-map schedule -- column-batched dense blocks, grouped weight gradients, every conv on the
                                               * bf16x6 small-map kernels with pre-split weight images (csrc/smallmap.hip) -- whatever options 5..7 say; 0: options 5..7 alone */
#define AFI_OPT_G_RDB_CHAIN 10                /* 0 (default): one launch per link of a dense block's chain of 32-channel convs.  1: under the small-map bf16x6
                                               * schedule the chain (y1 -> y2 -> y3 -> y4 forward, g4 -> g3 -> g2 -> g1 backward; growth rate 32) is ONE launch that
                                               * recomputes tile halos (afi_rdb_chain6_kernel): three launches per block and direction instead of five, 35 instead of 47
                                               * per config-1 forward + backward -- measured at break-even (19 us per chain launch against four links of 9 us minus the
                                               * 11 us GEMM it adds), so off by default; 2 / 3: backward / forward only (A/B) */
#define AFI_OPT_D_FOLD_BN_APPLY 11            /* 0 (default): every block's BatchNorm apply + LeakyReLU pass writes its activation.  1: where the discriminator's 3x3 convs
                                               * run in Winograd form the pass of blocks 0 and 1 is folded into the readers of the activation (the next block's input
                                               * transform, the backward's weight-gradient input transform read the saved conv output through the affine) and the
                                               * activation is never written (afi_discriminator_saved_activations) -- bit-identical results, one full read + write of
                                               * the activation less per block, and measured SLOWER: the transforms re-apply the affine on every overlapping tile read
                                               * (stage-1 step 103.7 against 103.5 ms, 113.6 against 113.2 with every kernel alone on the chip) */
#define AFI_OPT_DETERMINISTIC 12              /* 0 (default): weight-gradient GEMMs whose pixel range is split over blocks add their partial tiles by fp32 atomics
                                               * (summation order varies run to run: results agree to ~1e-6, not bit for bit).  1: no weight gradient is split
                                               * over blocks -- the Winograd TN GEMM and the direct weight-gradient kernel run one block per tile over the whole
                                               * pixel range, the interpolator's small-map backward takes the per-layer launches instead of the grouped
                                               * stream-K ones, bias sums take the two-stage fixed-order reduction -- so that two runs from the same state produce
                                               * the same bits (a resumed run continues bit for bit: tests/test_gpu_stage1.py).  Slower: for reproducing, not for speed. */
#define AFI_OPT_F16_PRESPLIT 13               /* 1 (default): under AFI_DTYPE_F16X3, a Winograd transform whose source tensor's largest magnitude is known before it
                                               * runs (the discriminator's activations and gradients: published by the BatchNorm passes that write them) writes
                                               * its planes already split into the two fp16 pieces and the GEMM stages them by DMA alone.  0: every plane is
                                               * written in fp32 and split when the GEMM reads its fragments (same results; stage-1 step 86.6 against 84.4 ms) */
#define AFI_OPT_F16_NT256_MIN_TILES 14        /* 512: the f16x3 NT GEMM takes its 256 x 256 tile (sixteen waves per block, half the operand bytes per product)
                                               * from this many tiles on (and 256-column multiples); 0: never (the 128 x 128 tile everywhere) */
#define AFI_OPT_F16_LOCAL_SUMS 15             /* under AFI_DTYPE_F16X3, which of the discriminator's forward convs that a backward follows (training == 1: their rounding
                                               * decides LeakyReLU masks) sum the three products of every k-step in a fresh fragment and add that fragment to the
                                               * accumulator with ONE fp32 addition, instead of three accumulating MFMAs: bit n + 1 = block n, 1 = every block, 0 = none.
                                               * Same products, fewer roundings of the large accumulator (an fp32 accumulate rounds at the accumulator's magnitude,
                                               * whatever the addend).  Pre-split planes and 256-column multiples only (blocks 1 and 2 at the reference's widths; other
                                               * shapes keep the plain order).  Default and measurements: DESIGN.md 0 / 4b */
#define AFI_OPT_D_FUSE_TAIL 16                /* 1 (default): the discriminator's last block and last conv (F3 -> 1, F3 <= 1024, F3 % 16 == 0) run fused -- the last conv reads the
                                               * block's saved conv output through its BatchNorm affine + LeakyReLU (the activation y[2] is never written:
                                               * afi_discriminator_saved_activations), and the block's BatchNorm backward GENERATES the gradient with respect to that
                                               * activation from the nine logit gradients of each pixel instead of reading it, taking the last conv's weight gradient
                                               * along: at 2 x 200 x 336 x 1024, 1.1 GB less traffic per forward and 2.2 GB less per backward.  Same decisions of
                                               * the LeakyReLU masks (the pinned affine), sums in another order.  0: the separate passes of rounds 1-5 */
#define AFI_OPT_D_FUSE_BWD_SUMS 17            /* 0 (default).  1: where the discriminator's data gradients run in Winograd form (and the call is not a paired one), the output
                                               * transform that writes the gradient with respect to a block's activation also takes that block's two BatchNorm-backward
                                               * sums (the LeakyReLU' mask and the normalised value recomputed from the saved conv output, fp64 partial rows), so the
                                               * separate sums pass -- one read of the gradient and one of the conv output -- becomes one read of the conv output.
                                               * Same gradients to fp32 rounding; measured: the transform with the sums takes 213 us more per call where the pass it
                                               * replaces took 185 (2 x 200 x 336, four calls per step), 69.5-69.7 against 69.1-69.5 ms per step -- off */
#define AFI_OPT_COUNT 18
int afi_ctx_set_option(afi_ctx_t* ctx, int option, long long value);
long long afi_ctx_get_option(const afi_ctx_t* ctx, int option);
/* The batched "NT" GEMM those convolutions run on, for tests and micro-benchmarks:  C[g][m][n] = sum_k A[g][m][k] * B[g][n][k] over
 * `planes` groups; rows_per_plane % 128 == 0, N % 128 == 0, K % 32 == 0 (else AFI_ERR_UNSUPPORTED); fp32 in memory for every dtype.
 * Under the bf16 settings the B operand (inside the library: the transformed weights, shared by every row tile of a plane) is first split
 * into bf16 parts in the order the kernel's LDS-DMA stages it; `scratch` (afi_gemm_nt_scratch_bytes; 0 for fp32) receives that image.
 * The library never allocates: too little scratch is AFI_ERR_WORKSPACE. */
long long afi_gemm_nt_scratch_bytes(int planes, int N, int K, int dtype);
int afi_gemm_nt(const float* A, const float* B, float* C, int planes, long long rows_per_plane, int N, int K, int dtype, void* scratch,
                long long scratch_bytes, void* stream);
/* ... and the weight-gradient form  dU[g][m][n] += sum_k Q[g][k][m] * V[g][k][n]  (both operands k-slow; rows_per_plane = K per plane,
 * % 32 == 0; M % 128 == 0, N % 128 == 0).  Split-K with fp32 atomics: the summation order varies run to run.  `scratch`
 * (afi_gemm_tn_scratch_bytes: 512 under AFI_DTYPE_F16X3 -- the per-plane maxima of both operands --, 0 otherwise; may be NULL then). */
long long afi_gemm_tn_scratch_bytes(int planes, int dtype);
int afi_gemm_tn(const float* Q, const float* V, float* dU, int planes, long long rows_per_plane, int M, int N, int dtype, void* scratch,
                long long scratch_bytes, void* stream);
/* Cache of the Winograd-transformed (and packed conv-transpose) weights.  Within one phase of a training step the same weights serve up
 * to ten calls (stage1_trainer.py:336-433: five levels x real / fake); each (weight pointer, tiling, direction) is transformed once and
 * re-used until afi_ctx_wino_weight_cache_invalidate() -- which the caller MUST issue whenever weight VALUES change (optimizer step,
 * load, broadcast); the weight MEMORY must stay allocated while entries exist (a freed-and-reused address would alias another weight).
 * Registering also invalidates.  140 M floats hold every transform of the reference's G and D. */
int afi_ctx_set_wino_weight_cache(afi_ctx_t* ctx, float* buf, long long floats);
int afi_ctx_wino_weight_cache_invalidate(afi_ctx_t* ctx);
/* Accumulator for the Winograd weight gradients.  While one is registered, every Winograd weight-gradient call adds its transform-domain
 * sum dU into a slot keyed by its dW target instead of zero-filling and transforming per call; the caller MUST call
 * afi_ctx_wino_wgrad_flush(ctx, stream) -- dW += alpha * A'^T dU A' for every slot, then the slots are released -- before it reads the
 * gradients (all-reduce, optimizer step), on a stream ordered after the calls that accumulated; after a failed phase
 * afi_ctx_wino_wgrad_discard drops the pending sums instead.  (Un)registering with sums pending is refused.  100 M floats hold every
 * slot of the reference's G and D. */
int afi_ctx_set_wino_wgrad_accum(afi_ctx_t* ctx, float* buf, long long floats);
int afi_ctx_wino_wgrad_flush(afi_ctx_t* ctx, void* stream);
int afi_ctx_wino_wgrad_discard(afi_ctx_t* ctx);

/* ------------------------------------------------------------------ whole-network entry points */

/* Parameters of Generator.Generators[0]  (generator_rdb.py:87-108; state_dict names in SURVEY.md 8b). */
typedef struct afi_gen_params {
    int C;                  /* in_channels (256) */
    int G;                  /* growth_rate (32) */
    int n_rdb;              /* n_residual_dense_blocks (3 in every caller) */
    float residual_scale;   /* 0.2 */
    float* w0; float* b0;                    /* Generators.0.0.0.{weight,bias}                       */
    float* rdb_w[AFI_MAX_RDB][5];            /* Generators.0.1.RDBs.r.conv{1..4}.0.weight, conv5.weight */
    float* w7; float* b7;                    /* Generators.0.2.0.{weight,bias}                       */
    float* wT; float* bT;                    /* Generators.0.3.0.{weight [Cin][Cout][6][6], bias}    */
    float* w9; float* b9;                    /* Generators.0.4.0.{weight,bias}                       */
} afi_gen_params_t;

/* floats of workspace needed by afi_generator_fwd (saved activations + packed conv-transpose weight) */
long long afi_generator_fwd_ws_floats(int C, int G, int n_rdb, int N, int H, int W);
/* floats of scratch needed by afi_generator_bwd */
long long afi_generator_bwd_ws_floats(int C, int G, int n_rdb, int N, int H, int W);

/* out[N,2H,2W,C] = bilinear_x2(x) + Generators[0](x)      (Generator.forward, generator_rdb.py:123-130) */
int afi_generator_fwd(afi_ctx_t* ctx, const afi_gen_params_t* prm, afi_view_t x, int N, int H, int W, afi_view_t out,
                      float* ws, long long ws_floats, void* stream);
/* Backward of the above.  `ws` is the forward workspace (unchanged since the forward), `dout` a DENSE
 * [N,2H,2W,C] gradient, `grads` the += targets laid out like the params (any pointer may be NULL to skip),
 * `dx` a DENSE [N,H,W,C] buffer or NULL when the input needs no gradient (stage 1: lr features are detached). */
int afi_generator_bwd(afi_ctx_t* ctx, const afi_gen_params_t* prm, const afi_gen_params_t* grads, afi_view_t x, int N, int H, int W,
                      const float* ws, const float* dout, float* dx, float* scratch, long long scratch_floats, void* stream);

/* Parameters of Discriminator.Discriminators[0] (feature_patch_discriminator.py:32-41). */
typedef struct afi_disc_params {
    int F[4];               /* channels: in (256), 512, 1024, 1024 */
    float* w[3]; float* b[3];                /* Discriminators.0.n.0.{weight,bias}          n = 0..2 */
    float* gamma[3]; float* beta[3];         /* Discriminators.0.n.0.norm.{weight,bias}              */
    float* running_mean[3]; float* running_var[3];
    long long* num_batches_tracked[3];
    float* w3; float* b3;                    /* Discriminators.0.3.0.{weight [1][3][3][F3], bias[1]} */
} afi_disc_params_t;

long long afi_discriminator_fwd_ws_floats(const int F[4], int N, int H, int W);
/* ABI v7.  The same for ONE context and ONE kind of call (`training` as afi_discriminator_fwd takes it): what that call really writes.  The
 * context-free query above is an upper bound for every context and mode -- it reserves the F(4x4) input planes a training forward of blocks 1 and
 * 2 may keep for its backward's weight gradient (36 x tiles x F[n] floats each: 0.6 + 1.25 GB at 2x200x336) -- this one reserves them only
 * where the context's arithmetic, AFI_OPT_WINOGRAD_F4_FORWARD / F16_PRESPLIT / D_FOLD_BN_APPLY and training == 1 make the forward keep them
 * (the default context: block 2 only).  Likewise the last block's activation y[2] (P x F[3] floats, the largest tensor of the network) is reserved
 * only where it is written: not under AFI_OPT_D_FUSE_TAIL (ABI v8: 0.55 GB per workspace at 2 x 200 x 336 x 1024).  Every offset that
 * afi_discriminator_ws_layout reports is the same either way (y[2] sits behind the fixed regions, the kept planes behind it), and a workspace
 * sized by either query serves afi_discriminator_fwd / _bwd under that context. */
long long afi_discriminator_fwd_ws_floats_ex(const afi_ctx_t* ctx, const int F[4], int N, int H, int W, int training);
long long afi_discriminator_bwd_ws_floats(const int F[4], int N, int H, int W);
/* Where afi_discriminator_fwd (training != 0) keeps what afi_discriminator_bwd reads, as offsets in floats into the forward
 * workspace: off12 = { c[0..2] conv outputs [P][F(n+1)], y[0..2] activations, mean[0..2], invstd[0..2] }.  For parity tooling
 * (tests feed the reference's saved activations to the backward; feature_patch_discriminator.py:35-38) and activation checkpoints. */
int afi_discriminator_ws_layout(const int F[4], int N, int H, int W, long long* off12);
/* Which activations the forward really writes, as a bit mask (bit n: y[n]).  Under AFI_OPT_D_FOLD_BN_APPLY = 1, where the 3x3 convs run in
 * Winograd form (maps of at least AFI_OPT_D_WINOGRAD_MIN_PIXELS pixels, AFI_OPT_WINOGRAD on), the activations of blocks 0 and 1 are NEVER
 * written: the next block's input
 * transform -- and the backward's weight-gradient input transforms -- read the saved conv output c[n] through the block's BatchNorm affine
 * and LeakyReLU, y = lrelu_0.2(((c - mean) * invstd) * gamma + beta) with every operation rounded to fp32 on its own (the arithmetic the
 * apply pass has; a reader of the workspace reproduces y[n] bit for bit from c[n], mean[n], invstd[n] and the block's gamma / beta that
 * way: tests/d_parity_util.py).  Bits 0 and 1 are clear there and set everywhere else (the default).  Bit 2 (y[2], which only the last conv
 * reads) is clear under AFI_OPT_D_FUSE_TAIL (the default: the mask is 3; 7 with that option off).
 * ctx may be NULL (defaults).
 * afi_discriminator_bwd must run under the same options as its forward. */
int afi_discriminator_saved_activations(const afi_ctx_t* ctx, const int F[4], int N, int H, int W);

/* logits[N,H,W] (dense) = Discriminators[0](x).  training != 0: batch statistics, running stats advance once,
 * num_batches_tracked += 1 (torch BatchNorm2d train mode);  training == 0: running statistics.
 * training == 1: afi_discriminator_bwd may follow on this workspace (the forward convs then keep the Winograd tiling whose
 * rounding does not disturb LeakyReLU mask decisions);  training == 2: train-mode statistics and logits only, no backward
 * will follow (stage-1 G phase, stage-2 generator-side terms): the convs may take the cheaper F(4x4,3x3) tiling, as in eval.
 * training == 3: as 2, but only the BatchNorm side effects are wanted (stage1_trainer.py:401-403: the G phase's D(real) call, whose
 * logits nothing reads): the running statistics and num_batches_tracked advance exactly as in mode 2; the last block's activation, the
 * last conv and the logits are not computed and `logits` may be NULL. */
int afi_discriminator_fwd(afi_ctx_t* ctx, const afi_disc_params_t* prm, afi_view_t x, int N, int H, int W, float* logits, int training,
                          float* ws, long long ws_floats, void* stream);
/* Backward (training-mode forward only).  grads: += targets (w, b, gamma, beta, w3, b3; other fields ignored).
 * dx: DENSE [N,H,W,F0] or NULL (both reference loops feed detached inputs).
 * PRECONDITION: `prm` holds the values the forward saw -- in particular gamma[n] / beta[n] must be BIT-IDENTICAL to the forward's.
 * The LeakyReLU' masks are recomputed here from the saved conv outputs c[n] (workspace) through the same affine
 * ((c - mean) * invstd) * gamma + beta, evaluated in the same order, instead of being read from the saved activations; a caller that
 * steps the BatchNorm affine between the two calls gets masks of ANOTHER affine and no error (the weights w[n] may not move either:
 * the data gradients read them).  Both reference loops run backward before any optimizer step (stage1_trainer.py:374-381,
 * stage2_trainer.py:335-342).  tests/test_gpu_d_parity.py::test_masks_recomputed_in_backward_match_the_forward counts the
 * disagreeing masks on un-nudged inputs.
 * PRECONDITION: the backward runs under the OPTIONS and the ARITHMETIC (compute dtype) its forward ran under -- on another context they must be set
 * alike.  Both passes derive from them what the workspace holds: under f16x3 the forward of blocks 1 and 2 (AFI_OPT_WINOGRAD_F4_FORWARD) keeps its
 * F(4x4) input planes, split into fp16 pieces, for the weight gradient of the same conv, and the backward reads them instead of transforming the
 * activation again; a backward that expects planes a differently-configured forward never wrote is refused where the library can see it
 * (AFI_ERR_BAD_ARG) and undefined where it cannot. */
int afi_discriminator_bwd(afi_ctx_t* ctx, const afi_disc_params_t* prm, const afi_disc_params_t* grads, afi_view_t x, int N, int H, int W,
                          const float* ws, const float* dlogits, float* dx, float* scratch, long long scratch_floats, void* stream);

/* TWO consecutive calls of the reference as one: x holds N images (N even), images [0, N/2) are the first call's batch and [N/2, N) the
 * second's -- D(real) then D(fake) in the D phase (stage1_trainer.py:349-359), D(fake) then D(real) in the G phase (:399-403).  Every
 * convolution runs once over all N images (half the launches, twice the rows per GEMM: what the small levels lack); every BatchNorm takes
 * its batch statistics over each half alone and moves the running statistics / num_batches_tracked twice, first half first, so the
 * parameters, buffers and logits come out as from the two calls (to fp32 rounding: under f16x3 the two halves share one operand scale per
 * Winograd plane -- a half whose largest magnitude lies 2^k below the other's keeps 22 - k significand bits in the FIRST conv; behind it every
 * BatchNorm has normalised the halves separately.  Pair tensors of like scale, as D(real) / D(fake) under an L1 term are).
 * Workspace sizes are those of the N-image call; `logits` / `dlogits` are [N, H, W]; the parameter gradients of both halves add up, as
 * two backward calls would leave them.  The second half's batch statistics sit behind the layout afi_discriminator_ws_layout reports.
 * Not available under AFI_OPT_D_FOLD_BN_APPLY (AFI_ERR_UNSUPPORTED). */
int afi_discriminator_fwd_paired(afi_ctx_t* ctx, const afi_disc_params_t* prm, afi_view_t x, int N, int H, int W, float* logits, int training,
                                 float* ws, long long ws_floats, void* stream);
int afi_discriminator_bwd_paired(afi_ctx_t* ctx, const afi_disc_params_t* prm, const afi_disc_params_t* grads, afi_view_t x, int N, int H, int W,
                                 const float* ws, const float* dlogits, float* dx, float* scratch, long long scratch_floats, void* stream);

/* ------------------------------------------------------------------ per-op entry points (also used by the tests) */

/* out[.., c_out] = act(alpha*conv3x3(x, w) + bias + beta*out);  w [Cout][3][3][Cin]  (generator_rdb.py:39-55,91-99,107)
 * lrelu: 0 = no activation, 1 = LeakyReLU(0.2), 2 = ReLU (used by the bench harness's guide network only) */
int afi_conv3x3_fwd(afi_ctx_t* ctx, afi_view_t x, int N, int H, int W, int Cin, const float* w, const float* bias, int Cout,
                    afi_view_t out, float alpha, float beta, int lrelu, void* stream);
/* dx = alpha*conv3x3^T(dy, w) + beta*dx, optionally times lrelu'(z) (z = the activation that produced x) */
int afi_conv3x3_dgrad(afi_ctx_t* ctx, afi_view_t dy, int N, int H, int W, int Cout, const float* w, int Cin, afi_view_t dx,
                      float alpha, float beta, afi_view_t z_or_null, void* stream);
/* dw[Cout][3][3][Cin] += alpha * sum_pix dy (x) x */
int afi_conv3x3_wgrad(afi_ctx_t* ctx, afi_view_t dy, afi_view_t x, int N, int H, int W, int Cout, int Cin, float* dw, float alpha, void* stream);

/* 1x1 convs of the AFI FPN lateral merge (fpn_sr.py:79-81,152-153), w [Cout][Cin]:
 * out = act(alpha*conv1x1(x, w) + bias + beta*out + r1_scale*r1)   -- r1 = the up-sampled top-down feature (or NULL) */
int afi_conv1x1_fwd(afi_ctx_t* ctx, afi_view_t x, int N, int H, int W, int Cin, const float* w, const float* bias, int Cout, afi_view_t out,
                    float alpha, float beta, afi_view_t r1_or_null, float r1_scale, int lrelu, void* stream);
int afi_conv1x1_dgrad(afi_ctx_t* ctx, afi_view_t dy, int N, int H, int W, int Cout, const float* w, int Cin, afi_view_t dx, float alpha, float beta,
                      void* stream);
int afi_conv1x1_wgrad(afi_ctx_t* ctx, afi_view_t dy, afi_view_t x, int N, int H, int W, int Cout, int Cin, float* dw, float alpha, void* stream);

/* The same 3x3 / stride-1 / pad-1 conv in Winograd F(2x2,3x3) form (2.25x fewer matrix-core FLOPs; large maps with many
 * channels): weight + input transforms, one batched 1x1 GEMM over the 16 transform points, output transform with the
 * epilogue.  fwd: out = conv(x, w) + bias.  dgrad: dx = conv^T(dy) * lrelu'(z) (z NULL = no mask).  ws from
 * afi_conv3x3_wino_ws_floats(N, H, W, Cin, Cout) (the same size serves both directions). */
long long afi_conv3x3_wino_ws_floats(int N, int H, int W, int Cin, int Cout);
int afi_conv3x3_wino_fwd(afi_ctx_t* ctx, afi_view_t x, int N, int H, int W, int Cin, const float* w_ohwi, const float* bias_or_null, int Cout,
                         afi_view_t out, float* ws, long long ws_floats, void* stream);
/* inference form: out = act(conv + bias), act 0 none / 1 LeakyReLU(0.2) / 2 ReLU; no backward will follow, so maps of >= 8192 pixels
 * take the F(4x4,3x3) tiling (4x fewer multiplies, ~3e-5 relative rounding) */
int afi_conv3x3_wino_infer(afi_ctx_t* ctx, afi_view_t x, int N, int H, int W, int Cin, const float* w_ohwi, const float* bias_or_null, int Cout,
                           afi_view_t out, int act, float* ws, long long ws_floats, void* stream);
int afi_conv3x3_wino_dgrad(afi_ctx_t* ctx, afi_view_t dy, int N, int H, int W, int Cout, const float* w_ohwi, int Cin, afi_view_t dx,
                           afi_view_t z_or_null, float* ws, long long ws_floats, void* stream);

/* weight gradient of the same conv in Winograd F(3x3,2x2) form (shares the input transform with the forward):
 * dw[Cout][3][3][Cin] += alpha * sum_pix dy (x) x */
int afi_conv3x3_wino_wgrad(afi_ctx_t* ctx, afi_view_t dy, afi_view_t x, int N, int H, int W, int Cout, int Cin, float* dw, float alpha, float* ws,
                           long long ws_floats, void* stream);

/* Conv2d(k=3, stride=2, padding=1) on [N,Hi,Wi,Cin] -> [N,Ho,Wo,Cout], Ho = ceil(Hi/2): the PAFPN bottom-up downsample conv
 * with its fused merge (pafpn_sr.py:105-117,177-183):
 *   a = act(conv(x,w) + bias)            -> act_out (or NULL; kept for the ReLU backward)
 *   out = post_scale*a + r_scale*r       (r NULL = no residual)
 * dgrad: dx[N,Hi,Wi,Cin] = alpha*conv^T(dy) + beta*dx, run as four parity-phase GEMMs (9 taps in total).
 * wgrad: dw[Cout][3][3][Cin] += alpha * sum_pix dy (x) x. */
int afi_conv3x3s2_fwd(afi_ctx_t* ctx, afi_view_t x, int N, int Hi, int Wi, int Cin, const float* w_ohwi, const float* bias_or_null, int Cout,
                      afi_view_t out, int act, afi_view_t act_out_or_null, float post_scale, afi_view_t r_or_null, float r_scale,
                      void* stream);
int afi_conv3x3s2_dgrad(afi_ctx_t* ctx, afi_view_t dy /*[N,Ho,Wo,Cout]*/, int N, int Hi, int Wi, int Cout, const float* w_ohwi, int Cin,
                        afi_view_t dx, float alpha, float beta, void* stream);
int afi_conv3x3s2_wgrad(afi_ctx_t* ctx, afi_view_t dy, afi_view_t x /*[N,Hi,Wi,Cin]*/, int N, int Hi, int Wi, int Cout, int Cin, float* dw,
                        float alpha, void* stream);
/* out[i] = scale * g[i] * (act[i] > 0): gradient through the ReLU of pafpn_sr.py:178 from its kept output (n % 4 == 0) */
int afi_relu_bwd(const float* g, const float* act, float* out, long long n, float scale, void* stream);

/* BiFPN_AFIGAN inference pieces around the interpolator (bifpn_sr.py:531-733, bifpn_layers/wrappers.py:172-252, activations.py):
 *   afi_dwconv3x3_fwd        SeparableConv2d.depthwise: 3x3, stride 1, zero pad 1, no bias; w9c = [9][C] (tap-major)
 *   afi_maxpool3s2_same_fwd  MaxPool2d(3, 2, "static_same"): zero pad right/bottom by 1 (the zeros take part in the max),
 *                            out dense [N, (H-2)/2+1, (W-2)/2+1, C]
 *   afi_fuse_swish_fwd       out = swish(w[0]*a + w[1]*b (+ w[2]*c)); a, b, c, out dense and equal-sized, w a DEVICE pointer
 * The pointwise 1x1 conv (+ folded eval-mode norm) is afi_conv1x1_fwd. */
int afi_dwconv3x3_fwd(afi_view_t x, int N, int H, int W, int C, const float* w9c, float* out, void* stream);
int afi_maxpool3s2_same_fwd(afi_view_t x, int N, int H, int W, int C, float* out, void* stream);
int afi_fuse_swish_fwd(const float* a, const float* b, const float* c_or_null, const float* w_dev, float* out, long long n,
                       void* stream);

/* BiFPN_AFIGAN under autograd (the same node, training mode): the backward of the three pieces above.  No atomics: every reduction is
 * two-stage in a fixed order.  The depthwise INPUT gradient is afi_dwconv3x3_fwd on dy with the taps reversed (w9c rows 8..0); the
 * pointwise conv is afi_conv1x1_{dgrad,wgrad}; the train-mode norm is afi_bn_stats_ex + afi_bn_apply_fwd + afi_bn_bwd.
 *   afi_fuse_swish_bwd           ds = dout * swish'(w.{a,b,c});  d{a,b,c} = w_k * ds (each optional), dw[2 or 3] = sum ds * {a,b,c}
 *                                (MemoryEfficientSwish.backward, bifpn_layers/activations.py); scratch: afi_fuse_swish_bwd_scratch_floats()
 *   afi_dwconv3x3_wgrad          dw9c[t][c] = sum_pixels dy[p][c] * x[p + tap t][c]; dy, x dense [N,H,W,C];
 *                                scratch: afi_dwconv3x3_wgrad_scratch_floats(C)
 *   afi_maxpool3s2_same_fwd_idx  the forward, also keeping idx[N,Ho,Wo,C] (one byte per element, 4-byte aligned): tap 3*dy+dx of the
 *                                first maximum in scan order, which is where torch's max_pool2d sends the gradient
 *   afi_maxpool3s2_same_bwd      dx dense [N,H,W,C] = gather of dout over the windows whose argmax is the element (a window won by the
 *                                zero pad sends its gradient nowhere, as F.pad's backward drops it) */
long long afi_fuse_swish_bwd_scratch_floats(void);
int afi_fuse_swish_bwd(const float* a, const float* b, const float* c_or_null, const float* w_dev, const float* dout, float* da_or_null,
                       float* db_or_null, float* dc_or_null, float* dw_or_null, long long n, float* scratch, void* stream);
long long afi_dwconv3x3_wgrad_scratch_floats(int C);
int afi_dwconv3x3_wgrad(const float* dy, const float* x, int N, int H, int W, int C, float* dw9c, float* scratch, void* stream);
int afi_maxpool3s2_same_fwd_idx(afi_view_t x, int N, int H, int W, int C, float* out, unsigned char* idx, void* stream);
int afi_maxpool3s2_same_bwd(const float* dout, const unsigned char* idx, int N, int H, int W, int C, float* dx, void* stream);

/* ConvTranspose2d(k=6,s=2,p=2) (generator_rdb.py:101-105) on the packed weight wp[4*Cout][3][3][Cin] */
int afi_convT6s2_pack_weight(const float* w_iohw, float* wp, int Cin, int Cout, void* stream);
int afi_convT6s2_unpack_wgrad(const float* dwp, float* dw_iohw, int Cin, int Cout, void* stream);   /* dw += */
int afi_convT6s2_fwd(afi_ctx_t* ctx, afi_view_t x, int N, int H, int W, int Cin, const float* wp, const float* bias, int Cout,
                     afi_view_t out /*[N,2H,2W,Cout]*/, int lrelu, void* stream);
int afi_convT6s2_dgrad(afi_ctx_t* ctx, afi_view_t dy /*[N,2H,2W,Cout]*/, int N, int H, int W, int Cout, const float* wp, int Cin,
                       afi_view_t dx, afi_view_t z_or_null, void* stream);
int afi_convT6s2_wgrad(afi_ctx_t* ctx, afi_view_t dy, afi_view_t x, int N, int H, int W, int Cout, int Cin, float* dwp, float alpha, void* stream);

/* out (dense [N,2H,2W,C]) = beta*out + bilinear_x2(x), align_corners=False (generator_rdb.py:125) */
int afi_bilinear2x_add_fwd(afi_view_t x, int N, int H, int W, int C, float beta, float* out, void* stream);
int afi_bilinear2x_add_bwd(const float* dout, int N, int H, int W, int C, float beta, float* dx, void* stream);

/* train-mode BatchNorm2d over a dense [P][C] matrix + LeakyReLU(0.2)  (feature_patch_discriminator.py:35-38) */
long long afi_reduce_scratch_floats(int C);
int afi_bn_stats(const float* x, long long P, int C, float* mean, float* invstd, float* var_biased_or_null,
                 float* running_mean_or_null, float* running_var_or_null, float* scratch, void* stream);
int afi_bn_apply_lrelu_fwd(const float* x, float* y, const float* mean, const float* invstd, const float* gamma,
                           const float* beta, long long P, int C, void* stream);
/* The same two passes for a norm with its own eps / momentum and no activation behind it (BiFPN: eps 1e-3, momentum 0.01,
 * bifpn_sr.py:279-280): statistics (+ running-stat update and num_batches_tracked += 1 when given), then
 * y = lrelu_slope((x - mean) * invstd * gamma + beta) with slope 1 = the plain affine.  Backward: afi_bn_bwd. */
int afi_bn_stats_ex(const float* x, long long P, int C, float eps, float momentum, float* mean, float* invstd, float* var_biased_or_null,
                    float* running_mean_or_null, float* running_var_or_null, long long* num_batches_tracked_or_null, float* scratch,
                    void* stream);
int afi_bn_apply_fwd(const float* x, float* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                     long long P, int C, float slope, void* stream);
int afi_bn_bwd(const float* g, const float* x, float* dx, const float* mean, const float* invstd, const float* gamma,
               float* dgamma, float* dbeta, long long P, int C, float* scratch, void* stream);
/* afi_bn_bwd in two halves, for a norm whose batch statistics span several ranks (norm = "SyncBN", bifpn_sr.py:210,279-280; torch.nn.SyncBatchNorm):
 * the sums of this rank's rows -- sums2C[0..C) = sum g, sums2C[C..2C) = sum g * xhat, xhat from the GLOBAL mean / invstd; dbeta / dgamma (optional) +=
 * them, per rank as torch keeps them --, which the caller all-reduces, then dx = gamma invstd (g - sums[0] / P_total - xhat sums[1] / P_total) over
 * this rank's P rows.  scratch: afi_reduce_scratch_floats(C). */
int afi_bn_bwd_sums(const float* g, const float* x, const float* mean, const float* invstd, float* dgamma_or_null, float* dbeta_or_null, float* sums2C,
                    long long P, int C, float* scratch, void* stream);
int afi_bn_bwd_apply(const float* g, const float* x, float* dx, const float* mean, const float* invstd, const float* gamma, const float* sums2C,
                     long long P, long long P_total, int C, void* stream);
/* db[C] += alpha * column sums of the [P][C] matrix g with row stride ld */
int afi_colsum_accum(const float* g, long long P, int C, long long ld, float alpha, float* db, float* scratch, void* stream);

/* nn.BCEWithLogitsLoss() mean vs a constant target (stage1_trainer.py:358-359,408):  *loss += lscale*bce;
 * dz (or NULL) = gscale * d bce / dz */
int afi_bce_logits_fwd_bwd(const float* z, long long n, float target, float lscale, float* loss, float gscale, float* dz, void* stream);
/* F.l1_loss mean over the common crop [N,h,w,C] of a and b (stage1_trainer.py:410,437-443): *loss += lscale*l1;
 * da (or NULL): DENSE [N,Ha,Wa,C] gradient w.r.t. a (zeros outside the crop) */
int afi_l1_fwd_bwd(afi_view_t a, afi_view_t b, int N, int h, int w, int C, int Ha, int Wa, float lscale, float* loss,
                   float gscale, float* da, void* stream);

/* multi-tensor SGD with momentum (torch.optim.SGD as built by detectron2 build_optimizer, stage1_trainer.py:110-114):
 *   d = g*gscale + wd*p ; buf = momentum*buf + d ; p -= lr*buf.    descs: DEVICE array of afi_sgd_desc_t */
typedef struct afi_sgd_desc { float* p; const float* g; float* m; long long n; float wd; float pad_; } afi_sgd_desc_t;
int afi_sgd_momentum_step(const afi_sgd_desc_t* descs_dev, int ntensors, long long max_n, float lr, float momentum,
                          float gscale, void* stream);
int afi_scale_inplace(float* p, long long n, float s, void* stream);

/* layout changes at the detectron2 boundary: [N][C][P] <-> [N][P][C] */
int afi_nchw_to_nhwc(const float* in, float* out, int N, int C, int P, void* stream);
int afi_nhwc_to_nchw(const float* in, float* out, int N, int C, int P, void* stream);

/* ------------------------------------------------------------------ dual-scale data path
 * afi_resize_bilinear_u8: what ResizeTransform.apply_image does to a uint8 image under the reference's DatasetMapper
 * (afigan/engine/dataset_mapper.py:85-107; target sizes from afigan/engine/transform_gen.py:198-217 for `image` and
 * :542-543 for `image_x0.5`): Pillow's antialiased BILINEAR resample, bit-exact (22-bit fixed-point coefficients, horizontal
 * pass rounded to uint8, then vertical pass), followed by the shared HFlipTransform when `hflip` != 0.
 *   src [H0][W0][C] uint8 (C = 1 or 3: Pillow modes L / RGB), dst [H1][W1][C] (out_chw == 0) or [C][H1][W1] (out_chw != 0, the mapper's tensor layout),
 *   ws  >= afi_resize_bilinear_u8_ws_bytes(...) bytes of device scratch (the coefficient tables).
 * afi_normalize_pad_u8: RCNN_FPN_only.forward's per-image normaliser and ImageList.from_tensors padding
 * (afigan/modeling/meta_arch/rcnn_only.py:36-39): out[c][y][x] = (img[c][y][x] - mean[c]) / std[c] in fp32 for y < H, x < W and
 * 0 up to Hp x Wp; img is [C][H][W] uint8 on the device, mean/std are HOST arrays of C floats, out is one [C][Hp][Wp] slot of the batch. */
/* afi_dual_scale_u8: both images of one sample in two launches -- `image` [H1,W1] and `image_x0.5` [H2,W2], each resized from
 * the same ORIGINAL src (dataset_mapper.py:103-105) with its own flip flag (equal when the flip is shared, transform_gen.py:546-554). */
long long afi_resize_bilinear_u8_ws_bytes(int H0, int W0, int C, int H1, int W1);
int afi_resize_bilinear_u8(const unsigned char* src, int H0, int W0, int C, unsigned char* dst, int H1, int W1, int hflip,
                           int out_chw, void* ws, long long ws_bytes, void* stream);
long long afi_dual_scale_u8_ws_bytes(int H0, int W0, int C, int H1, int W1, int H2, int W2);
int afi_dual_scale_u8(const unsigned char* src, int H0, int W0, int C, unsigned char* image, int H1, int W1, int hflip,
                      unsigned char* image_r, int H2, int W2, int hflip_r, int out_chw, void* ws, long long ws_bytes, void* stream);
int afi_normalize_pad_u8(const unsigned char* img_chw, int C, int H, int W, const float* mean, const float* std_,
                         float* out, int Hp, int Wp, void* stream);

/* ------------------------------------------------------------------ measurement support (bench.py)
 * When enabled, every MFMA GEMM launch is bracketed by two hipEvents recorded on the launch stream.
 * afi_profile_get(kind, out): out[0] launches, out[1] total ms, out[2] total algorithmic FLOP of that kernel since
 * the last afi_profile_enable(1).  One switch for the process (the launchers see streams, not contexts); its records are kept under a mutex, so
 * launches of several threads may be bracketed at once; the readers are meant for one benchmarking thread after a device synchronisation.
 * Off unless enabled: bench.py enables it for a separate pass of the same steps behind its timed region, not inside it (bench.py --profile-timed does). */
int afi_profile_enable(int on);
int afi_profile_num_kinds(void);
const char* afi_profile_kind_name(int kind);
int afi_profile_get(int kind, double* out3);
/* one CSV line per recorded launch (kind, GEMM rows, columns, K, split-K factor, ms, TFLOP/s) */
int afi_profile_dump(const char* path);

/* ------------------------------------------------------------------ diagnostics (host-only, no GPU needed)
 * The ownership the stream-K cut of the grouped small-map weight gradients (afi_generator_bwd on maps of <= 3000 pixels) gives every dW
 * tile: problem i has tiles[i] tiles of ceil(pixels[i] / 32) pixel stages; bpc = resident blocks per CU the runs are cut for (3).
 * Per tile (numbered problem-major): stored = runs that store it whole, added = runs that add to it by atomics, stages = stages covered.
 * A valid plan has (stored, added) = (1, 0) or (0, >= 2) and full stage coverage for every tile; tests/test_cabi.py checks exactly that. */
int afi_debug_wgrad_sk_plan(const long long* pixels, const int* tiles, int nprob, int bpc, int* stored, int* added, int* stages);
/* (GPU) The bf16x6 weight image of a conv-transpose weight W [Cin][Cout][6][6] built the two ways the library knows, for a byte comparison
 * (tests/test_gpu_ops.py): `direct` straight from W's own layout by the LDS-tiled blocks that ride in the small-map image launch, together
 * with the packed fp32 form `pack_ride` [4 Cout][9][Cin] written by further blocks of that launch; `via_pack` from the packed form
 * `pack_ref` made by the stand-alone pack kernel, through the generic image job.  mode 0: the forward's image (4 Cout columns, K = Cin),
 * mode 1: the data gradient's (Cin columns, K = (Cout chunk, phase, tap)).  Returns the image size in bytes through *bytes (buffers may be
 * null to query it).  Cin, Cout multiples of 32. */
int afi_debug_wk6_convT_images(const float* W, int Cin, int Cout, int mode, void* direct, float* pack_ride, void* via_pack, float* pack_ref,
                               long long* bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AFIGAN_HIP_H */
