// fp32-MFMA implicit-GEMM kernels for the AFI-GAN hot path on gfx950 (CDNA4).
//
// Every heavy op of the AF interpolator and of the feature-patch discriminator is a 3x3 correlation
// (SURVEY.md 2.1): forward convs, the ConvTranspose2d(k6,s2,p2) (= 4 phase-wise 3x3 convs + pixel
// shuffle), their data gradients and their weight gradients.  All of them run through two kernels:
//
//   afi_pix_gemm_kernel    C[pixel][col]  = sum_{tap,c} A[pixel (+/-) tap][c] * B[(tap,c)][col]
//   afi_wgrad_gemm_kernel  dW[row][tap][col] += sum_pixel dY[pixel][row] * X[pixel + tap][col]
//
// Design (MI355X-first, not a translated CUDA tiling):
//   * activations are pixel-major (NHWC): the K dimension (channels of one tap) is contiguous, so a
//     128-B cache line is one pixel's 32-channel chunk and the tap windows are re-read from L2, not HBM;
//   * exact-f32 matrix cores: v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD, bit-identical to an fmaf chain);
//     each wave owns MI x NI 32x32 accumulator blocks (64x64 per wave in the 128x128 tile);
//   * operands are staged through LDS in their NATURAL orientation, so there is never a transpose:
//       "KC" (k contiguous in memory)  -> LDS [row][BK+4], one ds_read_b128 feeds 4 MFMA k-steps
//                                         (16-B row pad => conflict-free b128 reads),
//       "RC" (row contiguous in memory) -> LDS [k][rows], ds_read_b32 (32 consecutive rows per half-wave);
//     the K order inside a stage is permuted identically for A and B (lane half h owns k = 8s+4h+j);
//   * register-staged prefetch (global -> VGPR while the MFMAs of the current stage run, VGPR -> LDS after
//     the barrier), 36 KB of LDS per 256-thread block so 4 blocks (16 waves) share a CU;
//   * XCD-aware block->tile map: each of the 8 XCDs gets one contiguous run of tiles (n fastest), so the
//     blocks that share an A tile hit the same private L2;
//   * fused epilogues: bias, LeakyReLU, alpha/beta, two residual adds, bilinear-x2 skip add, pixel-shuffle
//     store (conv-transpose), LeakyReLU-derivative mask (dgrad), channel-slice in/out of wider buffers
//     (the RDB dense buffer replaces every torch.cat of generator_rdb.py:66-68).
#include "afi_common.h"

#define AFI_BK 32
#define AFI_LDK (AFI_BK + 4)


// 16 bytes of zeros: masked lanes of the branch-free gathers read these instead of selecting after the load, so no
// instruction depends on a global load until the registers are written to LDS a stage later
__device__ __attribute__((aligned(16))) float afi_zeros[4] = {0.f, 0.f, 0.f, 0.f};   // non-const: stays in the global address space, so the select below keeps global_load (a const array is addrspace(4) and turns every gather into a flat_load)

#include "afi_epilogue.h"

// Second pass of the split-K form (small maps: too few tiles to fill 256 CUs): sum the per-split partial slabs in a FIXED
// order (bit-reproducible, no atomics, no memset) and apply the fused epilogue.  partial: [splitK][M][ldp].
template <bool POST>
__global__ __launch_bounds__(256) void afi_pix_splitk_epilogue_kernel(const AfiPixGemm p) {
    const int HW = p.H * p.W;
    const long long M = (long long)p.N * HW;
    const int ldp = (p.Ncols + 3) & ~3;
    const int C_F4 = ldp >> 2;
    const long long total = M * C_F4;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
        const long long m = it / C_F4;
        const int col = (int)(it - m * C_F4) * 4;
        // fixed summation order ks = 0, 1, 2, ... (bit-reproducible); the loads of four slabs are issued together so the
        // (latency-bound) pass does not serialise one L2 round trip per slab
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const float* src = p.partial + m * ldp + col;
        const long long slab = M * ldp;
        int ks = 0;
        for (; ks + 4 <= p.splitK; ks += 4) {
            const f32x4 t0 = *(const f32x4*)(src + (ks + 0) * slab), t1 = *(const f32x4*)(src + (ks + 1) * slab);
            const f32x4 t2 = *(const f32x4*)(src + (ks + 2) * slab), t3 = *(const f32x4*)(src + (ks + 3) * slab);
            v += t0; v += t1; v += t2; v += t3;
        }
        for (; ks < p.splitK; ++ks) v += *(const f32x4*)(src + ks * slab);
        const int img = (int)(m / HW);
        const int rem = (int)(m - (long long)img * HW);
        const int y = rem / p.W, x = rem - y * p.W;
        afi_epilogue_store<POST>(p, img, y, x, col, v);
    }
}

// HALO variant (3x3, stride-1 gathers on large maps): the M tile is an 8x16 pixel patch of one image and the A operand is
// staged ONCE per 32-channel chunk as the patch's 10x18 halo; the nine taps of the chunk are nine stages that read their
// fragments from that halo at shifted positions.  Per 9 stages a thread issues 6 A loads instead of 36 (global loads in
// the MFMA stream cost ~64 issue cycles each on this chip, measured), and the same for the LDS writes.
#define AFI_HALO_TY 8
#define AFI_HALO_TX 16
#define AFI_HALO_PIX ((AFI_HALO_TY + 2) * (AFI_HALO_TX + 2))
template <int BM, int BN, int WM, int WN, bool B_RC, int BK, bool HALO = false, bool GTAP = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4 ? 3 : 4)) void afi_pix_gemm_kernel(const AfiPixGemm p, int ntile_n, int ntiles, int chunk) {
    constexpr int LDK = BK + 4;                           // K-contiguous LDS rows: +16 B pad -> conflict-free ds_read_b128
    static_assert(!HALO || (BM == AFI_HALO_TY * AFI_HALO_TX && BK == 32), "halo variant: 8x16 patch, 32-channel chunks");
    static_assert(!(HALO && GTAP), "the halo variant is the plain stride-1 3x3 gather");
    constexpr int NT = 64 * WM * WN;                      // 256 threads (4 waves) or 512 (8 waves)
    constexpr int H_LOADS = (AFI_HALO_PIX * (BK / 4) + NT - 1) / NT;   // float4 loads per thread per halo (6)
    constexpr int HW_ = AFI_HALO_TX + 2;                  // halo row pitch in pixels
    constexpr int MI = BM / (32 * WM), NI = BN / (32 * WN);
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves per block");
    static_assert(MI >= 1 && NI >= 1, "tile too small for the wave layout");
    constexpr int K_F4 = BK / 4;                          // KC: float4 per row of a stage
    constexpr int K_RPP = NT / K_F4;                      // KC: rows covered per load pass
    constexpr int A_LOADS = BM / K_RPP;                   // float4 loads per thread per stage (A, KC)
    constexpr int B_F4 = BN / 4;                          // RC: float4 per k-row
    constexpr int B_LOADS = B_RC ? (BK * B_F4) / NT : BN / K_RPP;
    static_assert(A_LOADS >= 1 && B_LOADS >= 1, "tile too small");
    constexpr int B_ROWS_PER_PASS = NT / B_F4;            // RC

    constexpr int A_SLOTS = HALO ? H_LOADS : A_LOADS;     // A-side load slots (registers) per thread
    constexpr int A_TILE = (HALO ? AFI_HALO_PIX : BM) * LDK;   // floats per A stage buffer
    constexpr int B_TILE = B_RC ? BK * BN : BN * LDK;     // floats per B stage buffer
    constexpr int STAGE = A_TILE + B_TILE;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // one stage buffer: A at smem, B right behind it (double-buffering with one barrier per stage was measured slower: it
    // costs a third of the resident blocks)
    int* rowtab = (int*)(smem + STAGE);                   // [3][BM]: img, y, x of each tile row

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;

    // XCD-aware tile map: blocks b and b+8 share an XCD (its private L2).  Each XCD owns `chunk` consecutive M tiles and
    // walks them M-FASTEST for one N tile at a time, so the ~96 blocks resident on an XCD stream the SAME weight tile
    // (L2 hits) while each reads its own activation patch once per channel chunk.  (N-fastest order made every block pull
    // its 4.7 MB weight panel through the fabric: FETCH_SIZE 54 GB per step for this kernel vs ~6 GB algorithmic.)
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    // n_fastest (short-K GEMMs with small weight panels, i.e. the batched Winograd GEMM: a 128-column panel is K*512 B): walk the
    // N tiles of one M tile back to back instead, so the A tile is read from HBM once and re-used from L2 by its N tiles
    const int tile_n = p.n_fastest ? jb % ntile_n : jb / chunk;
    const int tile_m = xcd * chunk + (p.n_fastest ? jb / ntile_n : jb - tile_n * chunk);
    if (tile_n >= ntile_n || tile_m * ntile_n >= ntiles) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int HW = p.H * p.W;
    const long long M = (long long)p.N * HW;
    int patch_img = 0, patch_y0 = 0, patch_x0 = 0;        // HALO: image and origin of this block's 8x16 patch
    const long long b_img = (HALO || p.b_sImg == 0) ? 0 : (long long)(m0 / HW) * p.b_sImg;   // per-image weights: the launcher guarantees a tile stays inside one image

    // decode the tile's rows once (shared by the A gather and the epilogue)
    if constexpr (HALO) {
        const int tiles_x = (p.W + AFI_HALO_TX - 1) / AFI_HALO_TX, tiles_y = (p.H + AFI_HALO_TY - 1) / AFI_HALO_TY;
        patch_img = tile_m / (tiles_x * tiles_y);
        const int r = tile_m - patch_img * (tiles_x * tiles_y);
        patch_y0 = (r / tiles_x) * AFI_HALO_TY;
        patch_x0 = (r - (r / tiles_x) * tiles_x) * AFI_HALO_TX;
        if (tid < BM) {
            const int y = patch_y0 + tid / AFI_HALO_TX, x = patch_x0 + tid % AFI_HALO_TX;
            const bool ok = y < p.H && x < p.W;
            rowtab[tid] = ok ? patch_img : -1; rowtab[BM + tid] = y; rowtab[2 * BM + tid] = x;
        }
    } else if (tid < BM) {
        long long m = (long long)m0 + tid;
        int img = -1, y = -(1 << 20), x = -(1 << 20);
        if (m < M) {
            img = (int)(m / HW);
            int rem = (int)(m - (long long)img * HW);
            y = rem / p.W;
            x = rem - y * p.W;
        }
        rowtab[tid] = img; rowtab[BM + tid] = y; rowtab[2 * BM + tid] = x;
    }
    __syncthreads();

    // ---- loader state ----
    // Per tile row: the 64-bit offset of its centre pixel in A and a 9-bit mask of the taps that stay inside the image,
    // so a stage's gather is one scalar delta (tap / phase / channel chunk) + one add and one bit test per load.
    const int aq = tid % K_F4, ar = tid / K_F4;           // A (KC): float4 column, first row
    constexpr int A_STATE = HALO ? 1 : A_LOADS;           // the halo variant recomputes its (rare) addresses instead of keeping them
    long long a_off[A_STATE]; unsigned a_mask[A_STATE];
    if constexpr (HALO) {
        a_off[0] = 0; a_mask[0] = 0;
    } else
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int r = ar + K_RPP * i;
        const int img = rowtab[r], y = rowtab[BM + r], x = rowtab[2 * BM + r];
        unsigned m = 0;
        if (img >= 0) {
            if constexpr (GTAP) {
                for (int t = 0; t < p.ntaps; ++t) {
                    const int yy = y * p.a_stride + p.tap_dy[t], xx = x * p.a_stride + p.tap_dx[t];
                    if ((unsigned)yy < (unsigned)p.aH && (unsigned)xx < (unsigned)p.aW) m |= 1u << t;
                }
            } else if (p.ntaps == 9) {
#pragma unroll
                for (int t9 = 0; t9 < 9; ++t9) {
                    const int yy = y + p.a_sgn * (t9 / 3 - 1), xx = x + p.a_sgn * (t9 % 3 - 1);
                    if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) m |= 1u << t9;
                }
            } else {
                m = 1u;
            }
        }
        a_mask[i] = m;
        const int cm = GTAP ? p.a_up * p.a_stride : p.a_up;   // coordinate multiplier
        a_off[i] = (long long)(img < 0 ? 0 : img) * p.A.sN + (long long)(y * cm) * p.A.sH + (long long)(x * cm) * p.A.sW + 4 * aq;
    }
    const int Ck4 = (p.Ck + 3) & ~3;
    const int cchunks = (p.Ck + BK - 1) / BK;
    const int nK_total = p.ntaps * p.nKphase * cchunks;
    // split-K (blockIdx.y): this block multiplies stages [kc0, kc0 + nK) and leaves the raw partial tile in its slab
    const int kper = p.kper;
    const int kc0 = blockIdx.y * kper;
    const int nK = min(kper, nK_total - kc0);
    if (nK <= 0) return;                                  // (uniform) cannot happen with the launcher's splitK choice
    const int b_cq = tid % B_F4, b_kr = tid / B_F4;       // B (RC): float4 column, first k-row

    f32x4 a_reg[A_SLOTS], b_reg[B_LOADS];

    // K order: channel chunk outermost, then phase, tap innermost -- the 9 taps of one 32-channel chunk re-read (almost)
    // the same pixels in consecutive stages, so they hit L2 instead of going back to the fabric 9 times.
    // The gather of the next stage is issued ONE LOAD AT A TIME between the MFMA groups of the current stage (load_one):
    // a wave issues in order, so a monolithic block of address arithmetic would leave the matrix pipe idle behind it,
    // whereas ~10 VALU instructions + 1 load fit in the 256-cycle shadow of each group of four 32x32x2 MFMAs.
    int k_tap = kc0 % p.ntaps, k_kph = (kc0 / p.ntaps) % p.nKphase, k_c0 = (kc0 / (p.ntaps * p.nKphase)) * BK;   // NEXT stage to gather
    long long k_delta = 0; bool k_cok = false; bool k_more = true; int k_wtap = k_tap;
    auto stage_setup = [&](bool more) {                  // scalar per-stage part of the addresses
        int dy = 0, dx = 0;
        if constexpr (GTAP) { dy = p.tap_dy[k_tap]; dx = p.tap_dx[k_tap]; k_wtap = p.tap_w[k_tap]; }   // table offsets are final (the launcher requires a_sgn == +1)
        else { if (p.ntaps == 9) { dy = k_tap / 3 - 1; dx = k_tap - (k_tap / 3) * 3 - 1; } k_wtap = k_tap; }
        k_delta = (long long)(dy * p.a_sgn * p.a_up + (k_kph >> 1)) * p.A.sH + (long long)(dx * p.a_sgn * p.a_up + (k_kph & 1)) * p.A.sW + k_c0;
        k_cok = more && (k_c0 + 4 * aq) < Ck4;
        k_more = more;
    };
    auto stage_advance = [&]() {
        if (++k_tap == p.ntaps) {
            k_tap = 0;
            if (++k_kph == p.nKphase) { k_kph = 0; k_c0 += BK; }
        }
    };
    auto load_one = [&](int slot) {                       // slot is a compile-time constant after unrolling
        if (slot < A_SLOTS) {
            const int i = slot;
            if constexpr (HALO) {
                if (k_tap == 0) {                          // (uniform) a new channel chunk starts: fetch its halo once
                    // halo pixel hp = (tid + 256 i) / 8 of the 10x18 halo, float4 column aq (256 % 8 == 0)
                    const int hp = (tid + NT * i) / K_F4;
                    const int y = patch_y0 - 1 + hp / HW_, x = patch_x0 - 1 + hp % HW_;
                    const bool ok = k_cok && hp < AFI_HALO_PIX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                    const float* src = ok ? p.A.p + ((long long)patch_img * p.A.sN + (long long)y * p.A.sH + (long long)x * p.A.sW + 4 * aq + k_c0)
                                          : afi_zeros;
                    a_reg[i] = *(const f32x4*)src;
                }
            } else {
                const bool ok = k_cok && ((a_mask[i] >> k_tap) & 1u);
                const float* src = ok ? p.A.p + (a_off[i] + k_delta) : afi_zeros;   // branch-free: masked lanes read zeros
                a_reg[i] = *(const f32x4*)src;
            }
        } else if (slot < A_SLOTS + B_LOADS) {
            const int i = slot - A_SLOTS;
            if constexpr (!B_RC) {
                const int n = n0 + ar + K_RPP * i;
                const bool ok = k_cok && n < p.Ncols;
                const float* src = ok ? p.B + (b_img + (long long)n * p.b_sRow + (long long)k_wtap * p.b_sTap + k_c0 + 4 * aq) : afi_zeros;
                b_reg[i] = *(const f32x4*)src;
            } else {
                const int c = k_c0 + b_kr + B_ROWS_PER_PASS * i;
                const int n = n0 + 4 * b_cq;
                const bool ok = k_more && c < p.Ck && n < p.Ncols;
                const float* src = ok ? p.B + (b_img + (long long)(k_kph * p.Ck + c) * p.b_sRow + (long long)k_wtap * p.b_sTap + n) : afi_zeros;
                b_reg[i] = *(const f32x4*)src;
            }
        }
    };
    auto stage_store = [&](int tap_of_regs) {
        float* As = smem;
        float* Bs = As + A_TILE;
        if constexpr (HALO) {
            if (tap_of_regs == 0) {                        // (uniform) the registers hold a fresh halo
#pragma unroll
                for (int i = 0; i < H_LOADS; ++i)
                    if (tid + NT * i < AFI_HALO_PIX * K_F4) *(f32x4*)(As + ((tid + NT * i) / K_F4) * LDK + 4 * aq) = a_reg[i];
            }
        } else
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) *(f32x4*)(As + (ar + K_RPP * i) * LDK + 4 * aq) = a_reg[i];
        if constexpr (!B_RC) {
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) *(f32x4*)(Bs + (ar + K_RPP * i) * LDK + 4 * aq) = b_reg[i];
        } else {
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) *(f32x4*)(Bs + (b_kr + B_ROWS_PER_PASS * i) * BN + 4 * b_cq) = b_reg[i];
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    auto read_frags = [&](int s, int tap, f32x4 (&a)[MI], f32x4 (&b)[NI]) {
        const float* As = smem;
        const float* Bs = As + A_TILE;
        if constexpr (HALO) {
            // tile row m = 32*blk + lr is patch pixel (m / 16, m % 16); tap (dy, dx) reads halo pixel (+1 + sgn*dy, +1 + sgn*dx)
            const int shift = p.a_sgn * ((tap / 3 - 1) * HW_ + (tap - (tap / 3) * 3 - 1));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int hp = (((wm * MI + mi) * 32 + lr) / AFI_HALO_TX + 1) * HW_ + (lr % AFI_HALO_TX) + 1 + shift;
                a[mi] = *(const f32x4*)(As + hp * LDK + s * 8 + lh * 4);
            }
        } else
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
            a[mi] = *(const f32x4*)(As + ((wm * MI + mi) * 32 + lr) * LDK + s * 8 + lh * 4);
        if constexpr (!B_RC) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                b[ni] = *(const f32x4*)(Bs + ((wn * NI + ni) * 32 + lr) * LDK + s * 8 + lh * 4);
        } else {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int j = 0; j < 4; ++j) b[ni][j] = Bs[(s * 8 + lh * 4 + j) * BN + (wn * NI + ni) * 32 + lr];
        }
    };
    auto mfma_group = [&](const f32x4 (&a)[MI], const f32x4 (&b)[NI], int j) {
        // raised priority around the MFMA cluster: +1 % on the K-contiguous (forward) kernel, -2.4 % on the dgrad (measured)
        if constexpr (!B_RC) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
        if constexpr (!B_RC) __builtin_amdgcn_s_setprio(0);
    };
    // One stage out of LDS; the next stage's gather (scalar setup + all its loads) is issued between the 2nd and 3rd k-step,
    // i.e. in the middle of the stage's 64 MFMAs per wave.  (Pinning one load behind every MFMA group with sched_barrier
    // shortens a lone wave's bubble but was slower at 3 blocks per CU; moving the issue point changes nothing: DESIGN.md.)
    auto compute_stage = [&](bool more, int tap) {
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
            f32x4 fa[MI], fb[NI];
            read_frags(s, tap, fa, fb);
#pragma unroll
            for (int j = 0; j < 4; ++j) mfma_group(fa, fb, j);
            if (s == 1) {
                stage_setup(more);
#pragma unroll
                for (int slot = 0; slot < A_SLOTS + B_LOADS; ++slot) load_one(slot);
            }
        }
    };
    auto prefetch = [&]() {                               // prologue form: all loads of the next stage at once
        stage_setup(true);
#pragma unroll
        for (int slot = 0; slot < A_SLOTS + B_LOADS; ++slot) load_one(slot);
        stage_advance();
    };

    // registers -> LDS, barrier, the stage's MFMAs with the next stage's gather issued in their middle, barrier
    int c_tap = kc0 % p.ntaps;                             // tap of the stage about to be multiplied (== of the registers' data)
    prefetch();
    for (int kc = 0; kc < nK; ++kc) {
        stage_store(c_tap);
        __syncthreads();
        compute_stage(kc + 1 < nK, c_tap);                 // no branch: past the last stage every lane reads afi_zeros
        stage_advance();
        if (++c_tap == p.ntaps) c_tap = 0;
        __syncthreads();
    }

    // ---- epilogue: accumulators -> LDS (32 tile rows per wave row at a time) -> float4 rows, so every global access of
    //      the epilogue (store, residual / mask / bilinear reads) is a contiguous 16 B per lane, 512 B per 32 lanes ----
    constexpr int LDC = BN + 4;
    constexpr int C_F4 = BN / 4;
    static_assert(WM * 32 * LDC <= STAGE, "C staging tile must fit in the operand tiles");
    float* Cs = smem;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Cs[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + (wn * NI + ni) * 32 + lr] = acc[mi][ni][r];
        __syncthreads();
        for (int item = tid; item < WM * 32 * C_F4; item += NT) {
            const int rloc = item / C_F4, c4 = item - rloc * C_F4;
            const int rl = ((rloc >> 5) * MI + mi) * 32 + (rloc & 31);
            const int img = rowtab[rl];
            const int col = n0 + 4 * c4;
            if (img < 0 || col >= p.Ncols) continue;
            const f32x4 accv = *(const f32x4*)(Cs + rloc * LDC + 4 * c4);
            if (p.splitK > 1) {
                const int ldp = (p.Ncols + 3) & ~3;
                // slab row = the pixel's linear index (the halo variant's tile rows are patch-major, not linear)
                const long long mrow = HALO ? ((long long)img * p.H + rowtab[BM + rl]) * p.W + rowtab[2 * BM + rl] : (long long)(m0 + rl);
                *(f32x4*)(p.partial + ((long long)blockIdx.y * M + mrow) * ldp + col) = accv;
            } else {
                afi_epilogue_store<GTAP>(p, img, rowtab[BM + rl], rowtab[2 * BM + rl], col, accv);
            }
        }
        if (mi + 1 < MI) __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Plain batched "NT" GEMM for the Winograd planes:  C[g][m][n] = sum_k A[g][m][k] * B[g][n][k]   (both operands K-contiguous),
// every dimension tile-aligned (rows per plane % 128 == 0, N % 128 == 0, K % 32 == 0): no row table, no tap masks, no bounds
// checks, no zero page, one 64-bit base per operand: a leaner instruction stream than the general kernel, which leaves room for a second fragment register set
// (152 registers)
// (+4-5 % on the Winograd GEMMs).
// Same tile, LDS layout, fragment order and MFMA loop as afi_pix_gemm_kernel<128,128,2,2,KC>.
// ------------------------------------------------------------------------------------------------
struct AfiGemmNT {
    const float* A; const float* B; float* C;
    long long rows_per_plane;                              // rows of one plane (multiple of 128)
    int planes, N, K;                                      // B plane stride = N*K, C row pitch = N
};
template <int PF>
__global__ __launch_bounds__(256, 3) void afi_gemm_nt_kernel(const AfiGemmNT p, int ntile_n, int ntile_m, int chunk) {
    constexpr int BM = 128, BN = 128, BK = AFI_BK, LDK = BK + 4, WN = 2, MI = 2, NI = 2;
    constexpr int A_TILE = BM * LDK, B_TILE = BN * LDK;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    // N tiles of one M tile back to back (the A tile stays in L2 for them); each XCD owns `chunk` consecutive M tiles
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int tile_n = jb % ntile_n, tile_m = xcd * chunk + jb / ntile_n;
    if (tile_m >= ntile_m) return;
    const long long m0 = (long long)tile_m * BM;
    const int n0 = tile_n * BN;
    const int plane = (int)(m0 / p.rows_per_plane);
    const int aq = tid & 7, ar = tid >> 3;                 // float4 column, first row (32 rows per pass, 4 passes)
    const float* a_base = p.A + (m0 + ar) * p.K + 4 * aq;
    const float* b_base = p.B + ((long long)plane * p.N + n0 + ar) * p.K + 4 * aq;
    const long long pass = 32LL * p.K;
    const int nK = p.K / BK;

    // A (the HBM stream) is kept PF stages ahead in PF register sets; B (weights, L2-resident) one stage ahead in one set
    f32x4 a_reg[PF][4], b_reg[4];
    auto issue_a = [&](int set, int kc) {                  // past the end: re-read the last stage, never used
        const int k0 = (kc < nK ? kc : nK - 1) * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) a_reg[set][i] = *(const f32x4*)(a_base + i * pass + k0);
    };
    auto issue_b = [&](int kc) {
        const int k0 = (kc < nK ? kc : nK - 1) * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) b_reg[i] = *(const f32x4*)(b_base + i * pass + k0);
    };
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    float* As = smem;
    float* Bs = smem + A_TILE;
#pragma unroll
    for (int d = 0; d < PF; ++d) issue_a(d, d);
    issue_b(0);
    for (int kc = 0; kc < nK; kc += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            if (PF > 1 && kc + d >= nK) break;             // (uniform)
#pragma unroll
            for (int i = 0; i < 4; ++i) *(f32x4*)(As + (ar + 32 * i) * LDK + 4 * aq) = a_reg[d][i];
#pragma unroll
            for (int i = 0; i < 4; ++i) *(f32x4*)(Bs + (ar + 32 * i) * LDK + 4 * aq) = b_reg[i];
            __syncthreads();
            // fragments of slice s+1 are read from LDS while the 16 MFMAs of slice s issue (two register sets: 152 registers, +1.2 %)
            f32x4 fa[2][MI], fb[2][NI];
            auto frag = [&](int set, int s) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) fa[set][mi] = *(const f32x4*)(As + ((wm * MI + mi) * 32 + lr) * LDK + s * 8 + lh * 4);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) fb[set][ni] = *(const f32x4*)(Bs + ((wn * NI + ni) * 32 + lr) * LDK + s * 8 + lh * 4);
            };
            frag(0, 0);
#pragma unroll
            for (int s = 0; s < BK / 8; ++s) {
                if (s + 1 < BK / 8) frag((s + 1) & 1, s + 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][mi][j], fb[s & 1][ni][j], acc[mi][ni], 0, 0, 0);
                    __builtin_amdgcn_s_setprio(0);
                }
                if (s == 1) { issue_a(d, kc + d + PF); issue_b(kc + d + 1); }   // refill in the middle of the stage's MFMAs
            }
            __syncthreads();
        }
    }
    // epilogue: accumulators -> LDS -> float4 rows of C
    constexpr int LDC = BN + 4, C_F4 = BN / 4;
    float* Cs = smem;
    float* c_base = p.C + m0 * p.N + n0;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Cs[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + (wn * NI + ni) * 32 + lr] = acc[mi][ni][r];
        __syncthreads();
        for (int item = tid; item < 2 * 32 * C_F4; item += 256) {
            const int rloc = item / C_F4, c4 = item - rloc * C_F4;
            const int rl = ((rloc >> 5) * MI + mi) * 32 + (rloc & 31);
            *(f32x4*)(c_base + (long long)rl * p.N + 4 * c4) = *(const f32x4*)(Cs + rloc * LDC + 4 * c4);
        }
        if (mi + 1 < MI) __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradient: both operands are pixel-major, i.e. "RC" for a GEMM whose K runs over pixels
// ------------------------------------------------------------------------------------------------
#include "afi_wgrad_body.h"
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void afi_wgrad_gemm_kernel(const AfiWgradGemm p, int ntile_m, int ntile_n, int kper) {
    // XCD-aware order (bijective remap): blocks b, b+8, ... share an XCD, so give each XCD a contiguous run of logical ids
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    afi_wgrad_gemm_body<BM, BN, WM, WN>(p, ntile_m, ntile_n, kper, t, (int)blockIdx.y, gridDim.y > 1);
}

// ------------------------------------------------------------------------------------------------
// Plain batched "TN" GEMM for the Winograd weight gradient:  dU[g][m][n] += sum_k Q[g][k][m] * V[g][k][n]   (both operands
// row-major over k, i.e. the layout of afi_wgrad_gemm_kernel's LDS tiles), every dimension tile-aligned.  Same tile (128x128,
// four waves side by side), interleaved vector fragment reads and MFMA loop as afi_wgrad_gemm_kernel<128,128,1,4>, without its
// per-load pixel bookkeeping (y / x tracking, row and image wraps, tap bounds, zero page).
// ------------------------------------------------------------------------------------------------
struct AfiGemmTN {
    const float* Q; const float* V; float* dU;
    long long rows_per_plane;                              // K of one plane (multiple of 32)
    int planes, M, N;
};
__global__ __launch_bounds__(256) void afi_gemm_tn_kernel(const AfiGemmTN p, int ntile_m, int ntile_n, int kper) {
    constexpr int BM = 128, BN = 128, BK = AFI_BK, MI = 4, NI = 1;
    typedef float fragA __attribute__((ext_vector_type(MI)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [BK][BM]
    float* Bs = smem + BK * BM;       // [BK][BN]
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    int t;
    {   // contiguous run of logical ids per XCD; planes slowest, N tiles fastest (operand tiles of a plane re-used out of L2)
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int tile_n = t % ntile_n; t /= ntile_n;
    const int tile_m = t % ntile_m; const int plane = t / ntile_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const long long k_begin = (long long)blockIdx.y * kper;
    const long long k_end = (k_begin + kper < p.rows_per_plane) ? k_begin + kper : p.rows_per_plane;
    if (k_begin >= k_end) return;
    const int nK = (int)((k_end - k_begin) / BK);          // kper and rows_per_plane are multiples of BK
    const int cq = tid & 31, kr = tid >> 5;                // float4 column, first k row (8 rows per pass, 4 passes)
    const float* a_base = p.Q + ((long long)plane * p.rows_per_plane + k_begin + kr) * p.M + m0 + 4 * cq;
    const float* b_base = p.V + ((long long)plane * p.rows_per_plane + k_begin + kr) * p.N + n0 + 4 * cq;
    const long long a_pass = 8LL * p.M, b_pass = 8LL * p.N, a_stage = (long long)BK * p.M, b_stage = (long long)BK * p.N;
    f32x4 a_reg[4], b_reg[4];
    int k_next = 0;
    auto prefetch = [&]() {                                // past the end: re-read the last stage (never used)
        const int kc = k_next < nK ? k_next : nK - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) a_reg[i] = *(const f32x4*)(a_base + kc * a_stage + i * a_pass);
#pragma unroll
        for (int i = 0; i < 4; ++i) b_reg[i] = *(const f32x4*)(b_base + kc * b_stage + i * b_pass);
        ++k_next;
    };
    f32x16 acc[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
    const float* a_rd = As + MI * lr;
    const float* b_rd = Bs + wn * 32 + lr;
    prefetch();
    for (int kc = 0; kc < nK; ++kc) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)(As + (kr + 8 * i) * BM + 4 * cq) = a_reg[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)(Bs + (kr + 8 * i) * BN + 4 * cq) = b_reg[i];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
            fragA a[4]; float b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = *(const fragA*)(a_rd + (s * 8 + lh * 4 + j) * BM);
                b[j] = b_rd[(s * 8 + lh * 4 + j) * BN];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][mi], b[j], acc[mi], 0, 0, 0);
            if (s == 1) prefetch();
        }
        __syncthreads();
    }
    const bool use_atomic = gridDim.y > 1;
    float* out = p.dU + (long long)plane * p.M * p.N;
    const long long ldn = p.N;
    if (use_atomic) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + MI * ((r & 3) + 8 * (r >> 2) + 4 * lh) + mi;
                atomicAdd(out + (long long)row * ldn + n0 + wn * 32 + lr, acc[mi][r]);
            }
    } else {                                               // all old values first, then add and store (afi_gemm_bf16.h: the per-element form ran as serial round trips)
        float old[MI][16];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + MI * ((r & 3) + 8 * (r >> 2) + 4 * lh) + mi;
                old[mi][r] = out[(long long)row * ldn + n0 + wn * 32 + lr];
            }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + MI * ((r & 3) + 8 * (r >> 2) + 4 * lh) + mi;
                out[(long long)row * ldn + n0 + wn * 32 + lr] = old[mi][r] + acc[mi][r];
            }
    }
}

// ------------------------------------------------------------------------------------------------
// optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg)
// ------------------------------------------------------------------------------------------------
#include <vector>
#include <atomic>
#include <mutex>
namespace {
struct ProfRec { hipEvent_t a, b; int kind; double flops; long long m; int n, k, split, planes; };
// (process-wide by design: the launchers below see a stream, not a context.  One switch for the whole process, its records under a mutex:
//  launches of several threads may be bracketed at once; the readers are meant for one benchmarking thread after a synchronisation.)
struct ProfState {
    std::atomic<bool> on{false};
    std::mutex mu;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
} g_prof;
const char* kKindNames[] = {
    "pix_gemm<128x128,KC,linear> (1x1 / ragged-map / non-aligned Winograd GEMMs)", "pix_gemm<128x64,KC>", "pix_gemm<128x32,KC>", "pix_gemm<64x64,KC>",
    "pix_gemm<128x128,RC,linear> (conv dgrad, mid-size / ragged maps)", "pix_gemm<128x64,RC>", "pix_gemm<128x32,RC>", "pix_gemm<64x64,RC>",
    "wgrad_gemm<128x128>", "wgrad_gemm<64x128>", "wgrad_gemm<32x128>",
    "pix_gemm<128x128,KC,halo> (conv fwd)", "pix_gemm<128x128,RC,halo> (conv dgrad)",
    "gemm_nt<128x128> (batched Winograd GEMM)", "gemm_tn<128x128> (Winograd weight-gradient GEMM)",
    "pix_gemm_wk (small-map pixel GEMM, K split inside the block; grouped launches included)",
    "wgrad_group (grouped weight gradients of a small-map backward pass)",
    "gemm_nt_bf16<128x128> (batched Winograd GEMM on the bf16 MFMA: bf16x6 / bf16x3 / bf16 operands, fp32 accumulate)",
    "gemm_tn_bf16<128x128> (Winograd weight-gradient GEMM on the bf16 MFMA: bf16x6 / bf16x3 / bf16 operands, fp32 accumulate)",
    "wgrad_group6 (grouped weight gradients of a small-map backward pass on the bf16 MFMA, bf16x6 operands, fp32 accumulate)",
    "pix_gemm_wk6 (small-map pixel GEMM on the bf16 MFMA, bf16x6 operands on pre-split weight images, fp32 accumulate; grouped launches included)",
    "gemm_nt_f16x3 (batched Winograd GEMM on the f16 MFMA, 256x256 tiles where the grid fills the chip and 128x128 below: two scaled fp16 pieces per operand, three products, fp32 accumulate)",
    "gemm_tn_f16x3<128x128> (Winograd weight-gradient GEMM on the f16 MFMA: two scaled fp16 pieces per operand, three products, fp32 accumulate)"};   // one kind per kernel, as rocprofv3 lists them
constexpr int kNumKinds = 23;
hipEvent_t prof_event() {                                   // (callers hold g_prof.mu)
    if (g_prof.used == g_prof.pool.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        g_prof.pool.push_back(e);
    }
    return g_prof.pool[g_prof.used++];
}
struct ProfScope {
    hipStream_t st; int kind; double flops; hipEvent_t a{};
    long long m = 0; int n = 0, k = 0, split = 1, planes = 1;   // GEMM shape of the launch (rows over all planes, columns, K; split-K factor; planes), for afi_profile_dump
    ProfScope(hipStream_t s, int kd, double f) : st(s), kind(kd), flops(f) {
        if (g_prof.on) { std::lock_guard<std::mutex> lk(g_prof.mu); a = prof_event(); (void)hipEventRecord(a, st); started = true; }
    }
    bool live = true, started = false;
    void cancel() { live = false; }                        // nothing was launched under this scope (the caller falls back to another kernel)
    ~ProfScope() {
        if (started && live) { std::lock_guard<std::mutex> lk(g_prof.mu); hipEvent_t b = prof_event(); (void)hipEventRecord(b, st); g_prof.recs.push_back({a, b, kind, flops, m, n, k, split, planes}); }
    }
};
}  // namespace

extern "C" int afi_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (on) { g_prof.recs.clear(); g_prof.used = 0; }
    g_prof.on = on != 0;
    return AFI_OK;
}
extern "C" int afi_profile_num_kinds(void) { return kNumKinds; }
extern "C" const char* afi_profile_kind_name(int kind) { return (kind >= 0 && kind < kNumKinds) ? kKindNames[kind] : ""; }
// out[0] = launches, out[1] = total ms, out[2] = total algorithmic FLOP of kernel `kind` since afi_profile_enable(1)
extern "C" int afi_profile_get(int kind, double* out) {
    out[0] = out[1] = out[2] = 0.0;
    for (const ProfRec& r : g_prof.recs) {
        if (r.kind != kind) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) return AFI_ERR_LAUNCH;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return AFI_ERR_LAUNCH;
        out[0] += 1.0; out[1] += ms; out[2] += r.flops;
    }
    return AFI_OK;
}

// one CSV line per recorded launch: kind,rows,cols,k,split,planes,ms,tflops  (analysis aid: which shapes fill a kind's bucket;
// split is the split-K factor, or the bf16 parts setting (6 / 3 / 1) for the batched NT GEMM)
#include <stdio.h>
extern "C" int afi_profile_dump(const char* path) {
    FILE* f = fopen(path, "w");
    if (!f) return AFI_ERR_BAD_ARG;
    fprintf(f, "kind,rows,cols,k,split,planes,ms,tflops\n");
    for (const ProfRec& r : g_prof.recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) { fclose(f); return AFI_ERR_LAUNCH; }
        fprintf(f, "\"%s\",%lld,%d,%d,%d,%d,%.4f,%.2f\n", kKindNames[r.kind], r.m, r.n, r.k, r.split, r.planes, ms, ms > 0.f ? r.flops / (ms * 1e-3) / 1e12 : 0.0);
    }
    fclose(f);
    return AFI_OK;
}

// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
#include <stdlib.h>
// dynamic LDS beyond 64 KB is an opt-in per kernel AND per device (a process may drive several GPUs): done once for each pair
#include <mutex>
#include <set>
#include <utility>
static bool afi_opt_in_big_lds(const void* fn) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({dev, fn})) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    done.insert({dev, fn});
    return true;
}
template <int BM, int BN, int WM, int WN, bool B_RC, int BK = AFI_BK, bool HALO = false, bool GTAP = false>
static int launch_pix(const AfiPixGemm& p, hipStream_t st) {
    const long long M = (long long)p.N * p.H * p.W;
    const int ntm = HALO ? p.N * afi_cdiv(p.H, AFI_HALO_TY) * afi_cdiv(p.W, AFI_HALO_TX) : afi_cdiv(M, BM);
    const int ntn = afi_cdiv(p.Ncols, BN);
    const int ntiles = ntm * ntn;
    const int chunk = afi_cdiv(ntm, 8);                           // M tiles per XCD; grid = 8 XCDs x chunk x ntn
    const size_t lds = sizeof(float) * ((HALO ? AFI_HALO_PIX : BM) * (BK + 4) + (B_RC ? BK * BN : BN * (BK + 4))) + sizeof(int) * 3 * BM;
    const int kind = HALO ? (B_RC ? 12 : 11) : (B_RC ? 4 : 0) + (BM == 64 ? 3 : (BN == 128 ? 0 : (BN == 64 ? 1 : 2)));
    ProfScope prof(st, kind, 2.0 * (double)M * p.Ncols * p.ntaps * p.nKphase * p.Ck);
    prof.m = M; prof.n = p.Ncols; prof.k = p.ntaps * p.nKphase * p.Ck;
    if (lds > 64 * 1024) {   // beyond the default dynamic-LDS limit: opt in once per instantiation
        if (!afi_opt_in_big_lds((const void*)afi_pix_gemm_kernel<BM, BN, WM, WN, B_RC, BK, HALO, GTAP>)) return AFI_ERR_LAUNCH;
    }
    // split-K for small maps: with fewer tiles than ~2 per CU the serial K loop (72..288 stages of ~0.9 us) is pure latency;
    // spread it over blockIdx.y, keeping >= 4 stages per block and the slabs inside the caller's workspace
    AfiPixGemm q = p;
    q.splitK = 1;
    const int nK = p.ntaps * p.nKphase * afi_cdiv(p.Ck, BK);
    q.kper = nK;
    if (!HALO && p.partial && ntiles < 512) {
        int sk = afi_cdiv(1024, ntiles);
        if (sk > nK / 4) sk = nK / 4;
        const long long slab = M * ((p.Ncols + 3) & ~3);
        if (sk > 1 && slab * sk > p.partial_floats) sk = (int)(p.partial_floats / slab);
        if (sk > 1) {
            const int kper = afi_cdiv(nK, sk);
            q.splitK = afi_cdiv(nK, kper);                // no empty splits
            q.kper = kper;
        }
    }
    // split-K for MID-SIZE maps on the 128x128 tiles (0.5 .. 6 tiles per CU): whole tiles quantise badly onto 256 CUs x 3
    // resident blocks (546 tiles = 2.13 per CU run as long as 3 per CU: 93 of the kernel's 130 TFLOP/s; 264 tiles: 62), so cut
    // every tile's K range in 3-4 so the dispatcher has small pieces to balance with.  Measured (fwd, incl. the reduction
    // pass): 264 tiles 62 -> 85 TFLOP/s, 546: 93 -> 109, 1050: 104 -> 115; from ~1500 tiles on it only costs slab traffic.
    if (BM == 128 && BN == 128 && q.splitK == 1 && p.partial && ntiles >= 128 && ntiles < 1536) {
        int kper = afi_cdiv(nK, ntiles < 768 ? 4 : 3);
        if (HALO) kper = afi_cdiv(kper, 9) * 9;            // a split must start on a channel-chunk boundary (the halo is fetched at tap 0)
        if (kper < 9) kper = 9;
        const int sk = afi_cdiv(nK, kper);
        const long long slab = M * ((p.Ncols + 3) & ~3);
        if (sk > 1 && slab * sk <= p.partial_floats) { q.splitK = sk; q.kper = kper; }
    }
    prof.split = q.splitK;
    hipLaunchKernelGGL((afi_pix_gemm_kernel<BM, BN, WM, WN, B_RC, BK, HALO, GTAP>), dim3(chunk * ntn * 8, q.splitK), dim3(64 * WM * WN), lds, st, q, ntn, ntiles, chunk);
    if (q.splitK > 1) {
        const long long items = M * (((p.Ncols + 3) & ~3) >> 2);
        long long g = (items + 255) / 256; if (g > 2048) g = 2048;
        hipLaunchKernelGGL(afi_pix_splitk_epilogue_kernel<GTAP>, dim3((unsigned)g), dim3(256), 0, st, q);
    }
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// batched NT GEMM of the Winograd planes; returns AFI_ERR_UNSUPPORTED when a dimension is not tile-aligned (caller falls back)
int afi_launch_gemm_nt(const float* A, const float* B, float* C, int planes, long long rows_per_plane, int N, int K, hipStream_t st) {
    if (planes <= 0 || rows_per_plane <= 0 || N <= 0 || K <= 0) return AFI_ERR_BAD_ARG;
    if ((rows_per_plane % 128) || (N % 128) || (K % AFI_BK)) return AFI_ERR_UNSUPPORTED;
    AfiGemmNT g{A, B, C, rows_per_plane, planes, N, K};
    const long long M = rows_per_plane * planes;
    const int ntm = (int)(M / 128), ntn = N / 128, chunk = afi_cdiv(ntm, 8);
    const size_t lds = sizeof(float) * 2 * 128 * (AFI_BK + 4);
    ProfScope prof(st, 13, 2.0 * (double)M * N * K);
    prof.m = M; prof.n = N; prof.k = K; prof.planes = planes;
    // (two register sets for the A stream, PF = 2, spill under the 168-register cap of 3 blocks per CU: 95 instead of 262 TFLOP/s)
    hipLaunchKernelGGL((afi_gemm_nt_kernel<1>), dim3(chunk * ntn * 8), dim3(256), lds, st, g, ntn, ntm, chunk);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

#include "afi_gemm_bf16.h"
// the same GEMM on the bf16 matrix cores (split = 1: bf16 operands; 3: split-bf16, three MFMAs per k-step); fp32 planes in and out
// B (transformed weights) [planes][N][K] fp32 -> the pre-split bf16 LDS-image order the DMA kernel stages verbatim (afi_gemm_bf16.h)
int afi_launch_split_bf16_tiles(const float* B, void* out, int planes, int N, int K, int split, hipStream_t st) {
    if (!B || !out || planes <= 0 || (N % 128) || (K % 32) || (split != 1 && split != 3 && split != 6)) return AFI_ERR_BAD_ARG;
    const long long total = (long long)planes * N * (K / 4);
    const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    unsigned char* o = (unsigned char*)out;
    if (split == 6) hipLaunchKernelGGL((afi_split_bf16_tiles_kernel<6, 128>), grid, blk, 0, st, B, o, planes, N, K);
    else if (split == 3) hipLaunchKernelGGL((afi_split_bf16_tiles_kernel<3, 128>), grid, blk, 0, st, B, o, planes, N, K);
    else hipLaunchKernelGGL((afi_split_bf16_tiles_kernel<1, 128>), grid, blk, 0, st, B, o, planes, N, K);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
// the DMA form: A fp32 planes, B pre-split (afi_launch_split_bf16_tiles)
int afi_launch_gemm_nt_bf16_dma(const float* A, const void* Bsplit, float* C, int planes, long long rows_per_plane, int N, int K, int split, hipStream_t st) {
    if (planes <= 0 || rows_per_plane <= 0 || N <= 0 || K <= 0 || (split != 1 && split != 3 && split != 6)) return AFI_ERR_BAD_ARG;
    if ((rows_per_plane % 128) || (N % 128) || (K % 32)) return AFI_ERR_UNSUPPORTED;
    AfiGemmNT g{A, (const float*)Bsplit, C, rows_per_plane, planes, N, K};
    const long long M = rows_per_plane * planes;
    const int ntm = (int)(M / 128), ntn = N / 128, chunk = afi_cdiv(ntm, 8);
    const size_t stage = 16384u + (split == 6 ? 3u : (split == 3 ? 2u : 1u)) * 8192u, epi = sizeof(float) * 64u * (128u + 4u);
    // (four blocks per CU fill every register and all of LDS: kernels of the other stream only get on a CU when a block of this one retires.  Padding
    //  the LDS request to hold it at three blocks, to leave them room: the kernel alone 357 -> 426 us, the two-stream step 112.8 -> 118.4 ms.)
    const size_t lds = stage > epi ? stage : epi;
    ProfScope prof(st, 17, 2.0 * (double)M * N * K);
    prof.m = M; prof.n = N; prof.k = K; prof.split = split; prof.planes = planes;
    const dim3 grid(chunk * ntn * 8), blk(256);
    if (split == 6) hipLaunchKernelGGL((afi_gemm_nt_bf16_dma_kernel<6, 4>), grid, blk, lds, st, g, ntn, ntm, chunk);
    else if (split == 3) hipLaunchKernelGGL((afi_gemm_nt_bf16_dma_kernel<3, 4>), grid, blk, lds, st, g, ntn, ntm, chunk);
    else hipLaunchKernelGGL((afi_gemm_nt_bf16_dma_kernel<1, 4>), grid, blk, lds, st, g, ntn, ntm, chunk);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// ---- f16x3 (afi_gemm_f16.h): two fp16 pieces per operand, three MFMAs per k-step, power-of-two scales per operand and plane
#include "afi_gemm_f16.h"
// per-plane maxima of X [planes][per_plane] into out[planes]; `out` must already be zero (the launch only raises it)
int afi_launch_absmax_planes(const float* X, long long per_plane, int planes, float* out, hipStream_t st) {
    if (!X || !out || planes <= 0 || per_plane <= 0 || (per_plane & 3)) return AFI_ERR_BAD_ARG;
    long long g = (per_plane / 4 + 255) / 256;
    if (g > 512) g = 512;
    hipLaunchKernelGGL(afi_absmax_planes_kernel, dim3((unsigned)g, planes), dim3(256), 0, st, X, per_plane, out);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
// B (transformed weights) [planes][N][K] fp32 -> header (scales, maxima) + fp16 pieces in the NT kernel's LDS-image order.
// wkind = 0: the exact per-plane maxima of B by a pass of their own (three launches: zero the header, maxima, split; the stand-alone GEMM entry point).
// wkind = 5 / 6 (F(2x2) / F(4x4) weight planes): the header's slot [64] already holds the largest magnitude of the weight tensor the planes were
// transformed from -- the caller zero-filled the header (afi_f16_image_begin) and the weight transform raised it -- one launch.
long long afi_f16_image_bytes(int planes, int N, int K) { return AFI_F16_HDR_BYTES + (long long)planes * N * K * 4; }
AfiF16Bound afi_f16_bound(const float* amax, int kind);
int afi_f16_image_begin(void* out, hipStream_t st) { return hipMemsetAsync(out, 0, AFI_F16_HDR_BYTES, st) == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH; }
float* afi_f16_image_wmax(void* out) { return (float*)out + 64; }
int afi_launch_split_f16_tiles(const float* B, void* out, int planes, int N, int K, hipStream_t st, int wkind) {
    if (!B || !out || planes <= 0 || planes > 36 || (N % 128) || (K % 32)) return AFI_ERR_BAD_ARG;
    float* bmax = (float*)out + 64;
    if (wkind == 0) {
        AFI_TRY(afi_f16_image_begin(out, st));
        AFI_TRY(afi_launch_absmax_planes(B, (long long)N * K, planes, bmax, st));
    }
    const long long total = (long long)planes * N * (K / 4);
    hipLaunchKernelGGL((afi_split_f16_tiles_kernel<128>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, B, (unsigned char*)out, afi_f16_bound(bmax, wkind), planes, N, K);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
// the bounds of the Winograd transforms' planes relative to the largest magnitude of the tensor they transform (afi_gemm_f16.h): the
// product of the absolute row sums of the transform matrix.  kind: 0 exact per-plane maxima (stride 1, factor 1), 1 F(2x2) input
// (B^T d B), 2 F(4x4) input, 3 F(2x2) dY (G' e G'^T), 4 F(4x4) dY, 5 F(2x2) weights (G g G^T), 6 F(4x4) weights
AfiF16Bound afi_f16_bound(const float* amax, int kind) {
    AfiF16Bound b;
    b.amax = amax; b.stride = kind == 0 ? 1 : 0; b.pad_ = 0;
    // (absolute row sums of B^T, G', G for the interpolation points of csrc/winograd.hip, AFI_WINO4_POINTS = 1: {0, 1, -1, 1/2, -2, inf})
#if AFI_WINO4_POINTS == 1
    static const float r_in4[6] = {7.f, 5.f, 5.f, 6.f, 3.f, 7.f};
    static const float r_dy4[6] = {1.f, 4.f / 3.f, 4.f / 3.f, 2.f, 1.f, 1.f};
    static const float r_w4[6] = {1.f, 1.f, 1.f, 28.f / 15.f, 7.f / 15.f, 1.f};
#else
    static const float r_in4[6] = {10.f, 10.f, 10.f, 6.f, 6.f, 10.f};
    static const float r_dy4[6] = {0.25f, 4.f / 6.f, 4.f / 6.f, 15.f / 24.f, 15.f / 24.f, 1.f};
    static const float r_w4[6] = {0.25f, 0.5f, 0.5f, 7.f / 24.f, 7.f / 24.f, 1.f};
#endif
    static const float r_w2[4] = {1.f, 1.5f, 1.5f, 1.f};
    for (int a = 0; a < 36; ++a) {
        float c = 1.f;
        if (kind == 1) c = 4.f;
        else if (kind == 2) c = r_in4[a / 6] * r_in4[a % 6];
        else if (kind == 4) c = r_dy4[a / 6] * r_dy4[a % 6];
        else if (kind == 5) c = a < 16 ? r_w2[a / 4] * r_w2[a % 4] : 1.f;
        else if (kind == 6) c = r_w4[a / 6] * r_w4[a % 6];
        b.cmul[a] = c;
    }
    return b;
}
// A/B builds only (-DAFI_ABLATIONS, e.g. AFI_HIPCC_FLAGS=-DAFI_ABLATIONS python __graft_entry__.py --force into a copy selected through AFI_LIB_PATH;
// tools/micro/nt_f16_ablate.py): ONE process-wide switch naming which ablated instantiation of the 128 x 128 kernel the next launches take
// (afi_gemm_f16.h: ABL, wrong results but for bit 16), + 32 = never the 256 x 256 tile.  The product build compiles neither the setter nor the
// eleven ablated kernels: nothing in the shipped library is process-global (include/afigan_hip.h), and no exported call can make the GEMM wrong.
#ifdef AFI_ABLATIONS
static int g_nt_abl = 0;
extern "C" void afi_debug_set_nt_ablation(int v) { g_nt_abl = v; }
#else
static constexpr int g_nt_abl = 0;
#endif
// a_pre: A holds the planes already split into fp16 pieces (winograd.hip, afi_store_split4) with the scales of `ab`
int afi_launch_gemm_nt_f16x3(const float* A, const void* Bimg, float* C, int planes, long long rows_per_plane, int N, int K, const AfiF16Bound& ab, hipStream_t st,
                             bool a_pre, long long nt256_min_tiles, bool local_sums) {
    if (planes <= 0 || planes > 36 || rows_per_plane <= 0 || N <= 0 || K <= 0 || !ab.amax) return AFI_ERR_BAD_ARG;
    if ((rows_per_plane % 128) || (N % 128) || (K % 32)) return AFI_ERR_UNSUPPORTED;
    AfiGemmNT g{A, (const float*)Bimg, C, rows_per_plane, planes, N, K};
    const long long M = rows_per_plane * planes;
    const int ntm = (int)(M / 128), ntn = N / 128, chunk = afi_cdiv(ntm, 8);
    const size_t stage = 16384u + 2u * 8192u, epi = sizeof(float) * 64u * (128u + 4u);
    const size_t lds = stage > epi ? stage : epi;
    ProfScope prof(st, 21, 2.0 * (double)M * N * K);
    prof.m = M; prof.n = N; prof.k = K; prof.split = 2; prof.planes = planes;
    // the 256 x 256 tile (half the operand bytes per product) where it fills the chip: 256-column multiples, at least two rounds of CUs
    const int tpp = afi_cdiv(rows_per_plane, 256);
    const long long tiles256 = (long long)planes * tpp * (N / 256);
    if (local_sums && a_pre && !(N % 256)) {              // the k-step-local summation order lives in the 256 x 256 kernel on pre-split planes: taken whatever the tile count
        if (!afi_opt_in_big_lds((const void*)afi_gemm_nt_f16x3_w16_kernel<true, true>)) return AFI_ERR_LAUNCH;
        const int ntm2 = planes * tpp, ntn2 = N / 256, chunk2 = afi_cdiv(ntm2, 8);
        prof.split = 7;                                    // (afi_profile_dump: 7 = the k-step-local form)
        hipLaunchKernelGGL((afi_gemm_nt_f16x3_w16_kernel<true, true>), dim3(chunk2 * ntn2 * 8), dim3(1024), 16 * 16 * 132 * 4, st, g, ab, ntn2, ntm2, chunk2, tpp);
        return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
    }
    if (nt256_min_tiles > 0 && !(g_nt_abl & 32) && !(N % 256) && tiles256 >= nt256_min_tiles) {
        if (!afi_opt_in_big_lds(a_pre ? (const void*)afi_gemm_nt_f16x3_w16_kernel<true> : (const void*)afi_gemm_nt_f16x3_w16_kernel<false>)) return AFI_ERR_LAUNCH;
        const int ntm2 = planes * tpp, ntn2 = N / 256, chunk2 = afi_cdiv(ntm2, 8);
        prof.split = a_pre ? 4 : 3;                        // (afi_profile_dump: 2 / 3 = the 128 / 256 tile splitting A in registers, 5 / 4 = on pre-split planes)
        if (a_pre) hipLaunchKernelGGL(afi_gemm_nt_f16x3_w16_kernel<true>, dim3(chunk2 * ntn2 * 8), dim3(1024), 16 * 16 * 132 * 4, st, g, ab, ntn2, ntm2, chunk2, tpp);
        else hipLaunchKernelGGL(afi_gemm_nt_f16x3_w16_kernel<false>, dim3(chunk2 * ntn2 * 8), dim3(1024), 16 * 16 * 132 * 4, st, g, ab, ntn2, ntm2, chunk2, tpp);
        return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
    }
    if (a_pre) {
        prof.split = 5;
        hipLaunchKernelGGL((afi_gemm_nt_f16x3_kernel<4, 0, true>), dim3(chunk * ntn * 8), dim3(256), lds, st, g, ab, ntn, ntm, chunk);
        return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
    }
#ifdef AFI_ABLATIONS
    const int abl = g_nt_abl & 31;
    if (abl == 1) hipLaunchKernelGGL((afi_gemm_nt_f16x3_kernel<4, 1>), dim3(chunk * ntn * 8), dim3(256), lds, st, g, ab, ntn, ntm, chunk);
    else if (abl == 2) hipLaunchKernelGGL((afi_gemm_nt_f16x3_kernel<4, 2>), dim3(chunk * ntn * 8), dim3(256), lds, st, g, ab, ntn, ntm, chunk);
    else if (abl == 4) hipLaunchKernelGGL((afi_gemm_nt_f16x3_kernel<4, 4>), dim3(chunk * ntn * 8), dim3(256), lds, st, g, ab, ntn, ntm, chunk);
    else if (abl == 8) hipLaunchKernelGGL((afi_gemm_nt_f16x3_kernel<4, 8>), dim3(chunk * ntn * 8), dim3(256), lds, st, g, ab, ntn, ntm, chunk);
    else if (abl == 16) hipLaunchKernelGGL((afi_gemm_nt_f16x3_kernel<4, 16>), dim3(chunk * ntn * 8), dim3(256), lds, st, g, ab, ntn, ntm, chunk);
    else if (abl == 12) hipLaunchKernelGGL((afi_gemm_nt_f16x3_kernel<4, 12>), dim3(chunk * ntn * 8), dim3(256), lds, st, g, ab, ntn, ntm, chunk);
    else if (abl == 6) hipLaunchKernelGGL((afi_gemm_nt_f16x3_kernel<4, 6>), dim3(chunk * ntn * 8), dim3(256), lds, st, g, ab, ntn, ntm, chunk);
    else if (abl == 14) hipLaunchKernelGGL((afi_gemm_nt_f16x3_kernel<4, 14>), dim3(chunk * ntn * 8), dim3(256), lds, st, g, ab, ntn, ntm, chunk);
    else if (abl == 10) hipLaunchKernelGGL((afi_gemm_nt_f16x3_kernel<4, 10>), dim3(chunk * ntn * 8), dim3(256), lds, st, g, ab, ntn, ntm, chunk);
    else if (abl == 20) hipLaunchKernelGGL((afi_gemm_nt_f16x3_kernel<4, 20>), dim3(chunk * ntn * 8), dim3(256), lds, st, g, ab, ntn, ntm, chunk);
    else if (abl == 24) hipLaunchKernelGGL((afi_gemm_nt_f16x3_kernel<4, 24>), dim3(chunk * ntn * 8), dim3(256), lds, st, g, ab, ntn, ntm, chunk);
    else
#endif
    hipLaunchKernelGGL((afi_gemm_nt_f16x3_kernel<4>), dim3(chunk * ntn * 8), dim3(256), lds, st, g, ab, ntn, ntm, chunk);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
// pre: Q and V hold the planes already split into fp16 pieces
#ifndef AFI_TN_MAX_ROUNDS
// the split-K plan looks at no more than this many rounds of resident blocks (10: split 4 / 8 instead of 2 / 5 on D's largest layers fills the last
// round, and pays for it in atomics: 70.3-70.5 against 70.0-70.2 ms per step, three alternating rounds)
#define AFI_TN_MAX_ROUNDS 6
#endif
int afi_launch_gemm_tn_f16x3(const float* Q, const float* V, float* dU, int planes, long long rows_per_plane, int M, int N, const AfiF16Bound& qb, const AfiF16Bound& vb,
                             hipStream_t st, bool pre, bool deterministic) {
    if (planes <= 0 || planes > 36 || rows_per_plane <= 0 || M <= 0 || N <= 0 || !qb.amax || !vb.amax) return AFI_ERR_BAD_ARG;
    if ((rows_per_plane % 32) || (M % 128) || (N % 128)) return AFI_ERR_UNSUPPORTED;
    const int ntm = M / 128, ntn = N / 128;
    const long long tiles = (long long)ntm * ntn * planes;
    const int slots = AFI_TN_WAVES == 8 ? 768 : (AFI_TN_RING == 3 ? 768 : (AFI_TN_RING == 2 ? 1024 : 512));        // three resident blocks per CU (three 16 KB buffers, <= 168 registers); A/B ring of four: two
    int splitK = 1;
    {
        const int maxsplit = (int)(rows_per_plane / (16 * 32)) > 0 ? (int)(rows_per_plane / (16 * 32)) : 1;
        double best = -1.0;
        for (int s2 = 1; s2 <= maxsplit && s2 <= 128; ++s2) {
            const long long blocks = tiles * s2;
            if (blocks < 2 * slots && s2 < maxsplit) continue;
            if (blocks > AFI_TN_MAX_ROUNDS * slots && best >= 0.0) break;
            const long long rounds = (blocks + slots - 1) / slots;
            const double fill = (double)blocks / (double)(rounds * slots);
            if (fill > best + 1e-3) { best = fill; splitK = s2; }
        }
    }
    if (deterministic) splitK = 1;
    int kper = (int)((rows_per_plane + splitK - 1) / splitK);
    kper = ((kper + 31) / 32) * 32;
    splitK = (int)((rows_per_plane + kper - 1) / kper);
    AfiGemmTN g{Q, V, dU, rows_per_plane, planes, M, N};
    ProfScope prof(st, 22, 2.0 * (double)rows_per_plane * planes * M * N);
    prof.m = (long long)M * planes; prof.n = N; prof.k = (int)rows_per_plane; prof.split = splitK; prof.planes = planes;
    if (pre) hipLaunchKernelGGL(afi_gemm_tn_f16x3_pre_kernel, dim3((unsigned)tiles, splitK), dim3(64 * AFI_TN_WAVES), (unsigned)AFI_TN_RING * 4u * 4096u, st, g, qb, vb, ntm, ntn, kper);
    else hipLaunchKernelGGL(afi_gemm_tn_f16x3_kernel, dim3((unsigned)tiles, splitK), dim3(256), 2u * 4u * 4096u, st, g, qb, vb, ntm, ntn, kper);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}
int afi_launch_pix_gemm_sk(const AfiPixGemm& p, int b_rc, hipStream_t st);   // smallmap.hip
int afi_launch_pix_gemm_wk_group(const AfiPixGemm* probs, int n, int b_rc, hipStream_t st);
int afi_launch_wgrad_group(const AfiWgradGemm* probs, int n, int wide, hipStream_t st);
int afi_launch_wgrad6_group(const AfiWgradGemm* probs, int n, hipStream_t st);
// grouped small-map launches, bracketed for the live roofline like every other GEMM launch
int afi_launch_pix_gemm_group(const AfiPixGemm* probs, int n, int b_rc, hipStream_t st) {
    double fl = 0.0; long long m = 0; int nn = 0;
    for (int i = 0; i < n; ++i) {
        const long long M = (long long)probs[i].N * probs[i].H * probs[i].W;
        fl += 2.0 * (double)M * probs[i].Ncols * probs[i].ntaps * probs[i].nKphase * probs[i].Ck;
        m = M; nn += probs[i].Ncols;
    }
    bool all6 = n > 0;
    for (int i = 0; i < n; ++i) all6 = all6 && probs[i].Bimg != nullptr;
    ProfScope prof(st, all6 ? 20 : 15, fl);
    prof.m = m; prof.n = nn; prof.k = n > 0 ? probs[0].ntaps * probs[0].nKphase * probs[0].Ck : 0;
    const int rc = afi_launch_pix_gemm_wk_group(probs, n, b_rc, st);
    if (rc == AFI_ERR_UNSUPPORTED) prof.cancel();          // nothing was launched: the caller falls back to one launch per problem
    return rc;
}
int afi_launch_wgrad_gemm_group(const AfiWgradGemm* probs, int n, int wide, hipStream_t st) {
    if (n <= 0) return AFI_OK;                             // (nothing to launch: no profile bracket either)
    double fl = 0.0;
    for (int i = 0; i < n; ++i) fl += 2.0 * (double)probs[i].N * probs[i].H * probs[i].W * probs[i].Mrows * probs[i].Ncols * probs[i].ntaps;
    ProfScope prof(st, 16, fl);
    prof.m = n;
    return afi_launch_wgrad_group(probs, n, wide, st);
}
// the same group on the bf16 matrix cores (bf16x6); AFI_ERR_UNSUPPORTED = nothing launched (the caller falls back to the fp32 groups)
int afi_launch_wgrad_gemm_group6(const AfiWgradGemm* probs, int n, hipStream_t st) {
    if (n <= 0) return AFI_OK;
    double fl = 0.0;
    for (int i = 0; i < n; ++i) fl += 2.0 * (double)probs[i].N * probs[i].H * probs[i].W * probs[i].Mrows * probs[i].Ncols * probs[i].ntaps;
    ProfScope prof(st, 19, fl);
    prof.m = n; prof.split = 6;
    const int rc = afi_launch_wgrad6_group(probs, n, st);
    if (rc == AFI_ERR_UNSUPPORTED) prof.cancel();
    return rc;
}
int afi_launch_pix_gemm(const AfiPixGemm& p_in, int b_rc, hipStream_t st) {
    AfiPixGemm p = p_in;
    const long long M = (long long)p.N * p.H * p.W;
    if (M <= 0 || p.Ncols <= 0 || p.Ck <= 0) return AFI_ERR_BAD_ARG;
    if (p.a_bn.mean) return AFI_ERR_UNSUPPORTED;            // an operand read through a BatchNorm affine: the Winograd input transforms only
    if (p.b_sImg != 0 && ((long long)p.H * p.W) % 128 != 0) return AFI_ERR_BAD_ARG;   // per-image weights: tiles must not straddle images
    if (b_rc && (p.Ncols & 3)) return AFI_ERR_UNSUPPORTED;       // RC weight rows are read as float4 along n
    if (!b_rc && (p.Ck & 3)) return AFI_ERR_UNSUPPORTED;         // KC weight rows are read as float4 along c
    if (p.gtap) {
        // generic tap table (stride-2 conv forward / its dgrad phases): 1..9 taps with final offsets, two tile shapes
        if (p.ntaps < 1 || p.ntaps > 9 || p.a_sgn != 1 || p.nKphase != 1 || (p.a_stride != 1 && p.a_stride != 2)) return AFI_ERR_BAD_ARG;
        const bool small = M <= 64 * 256 || p.Ncols <= 64;
        if (!b_rc) return small ? launch_pix<64, 64, 2, 2, false, AFI_BK, false, true>(p, st) : launch_pix<128, 128, 2, 2, false, AFI_BK, false, true>(p, st);
        return small ? launch_pix<64, 64, 2, 2, true, AFI_BK, false, true>(p, st) : launch_pix<128, 128, 2, 2, true, AFI_BK, false, true>(p, st);
    }
    if (p.ntaps != 1 && p.ntaps != 9) return AFI_ERR_BAD_ARG;
    // tile choice: fill the N side first (weights are shared by every block), shrink M tiles for small maps
    // tiny problems (< 128 tiles of 128x128): 64x64 tiles; from there on 128x128 tiles + the mid-size split-K (D fwd+bwd at P4: 7.46 -> 7.04 ms)
    const bool smallM = (long long)afi_cdiv(M, 128) * afi_cdiv(p.Ncols, 128) < 128;
    // small maps (csrc/smallmap.hip): long K -> the stream-K kernel (equal MFMA count per CU whatever the shape); short K (<= 16
    // stages, e.g. the 32-channel data gradients of the dense blocks) -> ONE launch of whole tiles, no split and no second pass
    if (smallM && p.b_sImg == 0) {
        ProfScope prof(st, p.Bimg ? 20 : 15, 2.0 * (double)M * p.Ncols * p.ntaps * p.nKphase * p.Ck);
        prof.m = M; prof.n = p.Ncols; prof.k = p.ntaps * p.nKphase * p.Ck;
        const int rc = afi_launch_pix_gemm_sk(p, b_rc, st);
        if (rc != AFI_ERR_UNSUPPORTED || p.Bimg) return rc;  // (a problem that carries a weight image is defined by it: never fall back to reading B)
        prof.cancel();
    }
    // the kernels below read p.B.  A problem that exists only as a weight image (the dense block's four growth convs side by side as one data
    // gradient: nets.hip gives it a null B) cannot run on them: refuse instead of reading some other matrix out of bounds.  (A problem that
    // carries an image AND a valid B of its own shape -- every other small-map descriptor on a map too large for the small-map rule -- runs here.)
    if (!p.B) return AFI_ERR_UNSUPPORTED;
    // halo variant: 3x3 stride-1 gathers on maps big enough that the 8x16 patch grid wastes < 12 % of the MFMA work
    const long long padded = (long long)p.N * afi_cdiv(p.H, AFI_HALO_TY) * AFI_HALO_TY * afi_cdiv(p.W, AFI_HALO_TX) * AFI_HALO_TX;
    const bool halo = p.ntaps == 9 && p.nKphase == 1 && p.a_up == 1 && !smallM && p.Ncols > 64 && padded * 100 <= M * 112;
    if (!b_rc) {
        if (p.Ncols <= 32) return launch_pix<128, 32, 4, 1, false>(p, st);
        if (p.Ncols <= 64) return smallM ? launch_pix<64, 64, 2, 2, false>(p, st) : launch_pix<128, 64, 2, 2, false>(p, st);
        if (smallM) return launch_pix<64, 64, 2, 2, false>(p, st);
        if (halo) return launch_pix<128, 128, 2, 2, false, 32, true>(p, st);
        return launch_pix<128, 128, 2, 2, false>(p, st);
    } else {
        if (p.Ncols <= 32) return launch_pix<128, 32, 4, 1, true>(p, st);
        if (p.Ncols <= 64) return smallM ? launch_pix<64, 64, 2, 2, true>(p, st) : launch_pix<128, 64, 2, 2, true>(p, st);
        if (smallM) return launch_pix<64, 64, 2, 2, true>(p, st);
        if (halo) return launch_pix<128, 128, 2, 2, true, 32, true>(p, st);
        return launch_pix<128, 128, 2, 2, true>(p, st);
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_wgrad(const AfiWgradGemm& p, hipStream_t st) {
    const long long P = (long long)p.N * p.H * p.W;
    const int ntm = afi_cdiv(p.Mrows, BM), ntn = afi_cdiv(p.Ncols, BN);
    const long long tiles = (long long)ntm * ntn * p.ntaps;
    // split the pixel (K) range until the grid covers the chip ~4x, but keep >= 8 stages per block
    int splitK = p.splitK;
    if (splitK <= 0) {
        splitK = 1;
        const long long want = 1024;
        if (tiles < want) splitK = (int)((want + tiles - 1) / tiles);
        const int maxsplit = (int)((P + 8 * AFI_BK - 1) / (8 * AFI_BK));
        // Balance: the split is free here (partial sums meet in atomics), so pick the one whose block count fills whole
        // rounds of the chip's resident slots (256 CUs x 3 blocks): 2-6 rounds, best fill, fewest splits on ties.  "Cover the chip
        // ~4x" alone left 1.4-1.5 rounds for every big layer (D1: 288 tiles x 4 = 1152 blocks on 768 slots).
        if (BM == 128) {
            const long long slots = 768;
            double best = -1.0; int best_s = 0;
            for (int s2 = 1; s2 <= maxsplit && s2 <= 128; ++s2) {
                const long long blocks = tiles * s2;
                if (blocks < 2 * slots) continue;
                if (blocks > 6 * slots) break;
                const long long rounds = (blocks + slots - 1) / slots;
                const double fill = (double)blocks / (double)(rounds * slots);
                if (fill > best + 1e-3) { best = fill; best_s = s2; }
            }
            if (best_s > 0) splitK = best_s;
        }
        if (splitK > maxsplit) splitK = maxsplit;
        if (splitK < 1) splitK = 1;
    }
    int kper = (int)((P + splitK - 1) / splitK);
    kper = ((kper + AFI_BK - 1) / AFI_BK) * AFI_BK;
    splitK = (int)((P + kper - 1) / kper);
    const size_t lds = sizeof(float) * AFI_BK * (BM + BN);
    ProfScope prof(st, BM >= 128 ? 8 : (BM == 64 ? 9 : 10), 2.0 * (double)P * p.Mrows * p.Ncols * p.ntaps);
    prof.m = (long long)p.Mrows * p.ntaps; prof.n = p.Ncols; prof.k = (int)(P > 2147483647LL ? 2147483647LL : P); prof.split = splitK;
    if (lds >= 64 * 1024) {
        if (!afi_opt_in_big_lds((const void*)afi_wgrad_gemm_kernel<BM, BN, WM, WN>)) return AFI_ERR_LAUNCH;
    }
    hipLaunchKernelGGL((afi_wgrad_gemm_kernel<BM, BN, WM, WN>), dim3((unsigned)tiles, splitK), dim3(64 * WM * WN), lds, st, p, ntm, ntn, kper);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// batched TN GEMM of the Winograd weight gradient; AFI_ERR_UNSUPPORTED when a dimension is not tile-aligned (caller falls back)
// deterministic: one block per tile over the whole K range (no split, no atomics: the same bits every run)
int afi_launch_gemm_tn(const float* Q, const float* V, float* dU, int planes, long long rows_per_plane, int M, int N, hipStream_t st, bool deterministic) {
    if (planes <= 0 || rows_per_plane <= 0 || M <= 0 || N <= 0) return AFI_ERR_BAD_ARG;
    if ((rows_per_plane % AFI_BK) || (M % 128) || (N % 128)) return AFI_ERR_UNSUPPORTED;
    const int ntm = M / 128, ntn = N / 128;
    const long long tiles = (long long)ntm * ntn * planes;
    // split the K range so the block count fills whole rounds of the 768 resident-block slots (2..6 rounds), >= 8 stages per block
    int splitK = 1;
    {
        const int maxsplit = (int)(rows_per_plane / (8 * AFI_BK)) > 0 ? (int)(rows_per_plane / (8 * AFI_BK)) : 1;
        double best = -1.0;
        for (int s2 = 1; s2 <= maxsplit && s2 <= 128; ++s2) {
            const long long blocks = tiles * s2;
            if (blocks < 2 * 768 && s2 < maxsplit) continue;
            if (blocks > 6 * 768 && best >= 0.0) break;
            const long long rounds = (blocks + 767) / 768;
            const double fill = (double)blocks / (double)(rounds * 768);
            if (fill > best + 1e-3) { best = fill; splitK = s2; }
        }
    }
    if (deterministic) splitK = 1;
    int kper = (int)((rows_per_plane + splitK - 1) / splitK);
    kper = ((kper + AFI_BK - 1) / AFI_BK) * AFI_BK;
    splitK = (int)((rows_per_plane + kper - 1) / kper);
    AfiGemmTN g{Q, V, dU, rows_per_plane, planes, M, N};
    ProfScope prof(st, 14, 2.0 * (double)rows_per_plane * planes * M * N);
    prof.m = (long long)M * planes; prof.n = N; prof.k = (int)rows_per_plane; prof.split = splitK; prof.planes = planes;
    hipLaunchKernelGGL(afi_gemm_tn_kernel, dim3((unsigned)tiles, splitK), dim3(256), sizeof(float) * AFI_BK * 256, st, g, ntm, ntn, kper);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

int afi_launch_gemm_tn_bf16(const float* Q, const float* V, float* dU, int planes, long long rows_per_plane, int M, int N, int split, hipStream_t st, bool deterministic) {
    if (planes <= 0 || rows_per_plane <= 0 || M <= 0 || N <= 0 || (split != 1 && split != 3 && split != 6)) return AFI_ERR_BAD_ARG;
    if ((rows_per_plane % 32) || (M % 128) || (N % 128)) return AFI_ERR_UNSUPPORTED;
    const int ntm = M / 128, ntn = N / 128;
    const long long tiles = (long long)ntm * ntn * planes;
    // resident blocks: three per CU for the six-product form (one 48 KB buffer, 166 registers), two for the double-buffered others;
    // a stage is ~16x shorter than the fp32 kernel's, so blocks keep >= 16 stages each
    const int slots = split == 6 ? 768 : 512;
    int splitK = 1;
    {
        const int maxsplit = (int)(rows_per_plane / (16 * 32)) > 0 ? (int)(rows_per_plane / (16 * 32)) : 1;
        double best = -1.0;
        for (int s2 = 1; s2 <= maxsplit && s2 <= 128; ++s2) {
            const long long blocks = tiles * s2;
            if (blocks < 2 * slots && s2 < maxsplit) continue;
            if (blocks > 6 * slots && best >= 0.0) break;
            const long long rounds = (blocks + slots - 1) / slots;
            const double fill = (double)blocks / (double)(rounds * slots);
            if (fill > best + 1e-3) { best = fill; splitK = s2; }
        }
    }
    if (deterministic) splitK = 1;
    int kper = (int)((rows_per_plane + splitK - 1) / splitK);
    kper = ((kper + 31) / 32) * 32;
    splitK = (int)((rows_per_plane + kper - 1) / kper);
    AfiGemmTN g{Q, V, dU, rows_per_plane, planes, M, N};
    ProfScope prof(st, 18, 2.0 * (double)rows_per_plane * planes * M * N);
    prof.m = (long long)M * planes; prof.n = N; prof.k = (int)rows_per_plane; prof.split = splitK; prof.planes = planes;
    const bool db = split != 6;                            // six-product form: one 48 KB buffer
    const size_t lds = (db ? 2u : 1u) * 2u * (split == 6 ? 3u : (split == 3 ? 2u : 1u)) * 8192u;
    const dim3 grid((unsigned)tiles, splitK), blk(256);
    if (split == 6) hipLaunchKernelGGL(afi_gemm_tn_bf16x6_pipe_kernel, grid, blk, lds, st, g, ntm, ntn, kper);   // (the same 48 KB: two buffers of 24)
    else if (split == 3) hipLaunchKernelGGL((afi_gemm_tn_bf16_kernel<3, true>), grid, blk, lds, st, g, ntm, ntn, kper);
    else hipLaunchKernelGGL((afi_gemm_tn_bf16_kernel<1, true>), grid, blk, lds, st, g, ntm, ntn, kper);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

int afi_launch_wgrad_gemm(const AfiWgradGemm& p, hipStream_t st) {
    const long long P = (long long)p.N * p.H * p.W;
    if (P <= 0 || p.Mrows <= 0 || p.Ncols <= 0) return AFI_ERR_BAD_ARG;
    if (p.ntaps != 1 && p.ntaps != 9 && p.ntaps != 16 && p.ntaps != 36) return AFI_ERR_BAD_ARG;     // 16 / 36: Winograd transform points (no spatial shift, operand planes)
    if ((p.Ncols & 3) || (p.dy_up == 2 && (p.CoutPhase & 3))) return AFI_ERR_UNSUPPORTED;   // float4 granularity
    if (p.Mrows <= 32) return launch_wgrad<32, 128, 1, 4>(p, st);
    if (p.Mrows <= 64) return launch_wgrad<64, 128, 2, 2>(p, st);
    // 4 waves side by side (each 128 rows x 32 columns): the dY fragment is one un-fusable ds_read_b128 per k-row; with the
    // 2x2 layout hipcc fuses pairs of ds_read_b64 into ds_read2st64_b64, which runs at half the LDS rate (107.9 vs 106.9 TFLOP/s)
    return launch_wgrad<128, 128, 1, 4>(p, st);
}
