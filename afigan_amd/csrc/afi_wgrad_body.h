// Body of the weight-gradient GEMM (both operands pixel-major, K runs over pixels), shared by the per-layer kernel (igemm.hip)
// and the grouped kernel that runs every weight gradient of a small-map backward pass in one launch (smallmap.hip).
#pragma once
#include "afi_common.h"
#ifndef AFI_BK
#define AFI_BK 32
#endif
// the including translation unit defines the 16-byte zero page `afi_zeros` (a non-const __device__ array) before this header

// pixel range [k_begin, k_end) of logical tile t; k_begin a multiple of 32.  Ends on a block barrier: may be called again by the same block.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void afi_wgrad_gemm_range(const AfiWgradGemm& p, int ntile_m, int ntile_n, int t, long long k_begin, long long k_end, bool use_atomic) {
    constexpr int BK = AFI_BK;
    constexpr int NT = 64 * WM * WN;                      // 256 threads (4 waves) or 512 (8 waves: the 256x256 tile)
    constexpr int MI = BM / (32 * WM), NI = BN / (32 * WN);
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves per block");
    static_assert(MI == 1 || MI == 2 || MI == 4, "vector fragment reads");
    static_assert(NI == 1 || NI == 2 || NI == 4, "vector fragment reads");
    constexpr int A_F4 = BM / 4, B_F4 = BN / 4;
    constexpr int A_LOADS = (BK * A_F4) / NT, B_LOADS = (BK * B_F4) / NT;
    static_assert(A_LOADS >= 1 && B_LOADS >= 1, "tile too small");
    constexpr int A_RPP = NT / A_F4, B_RPP = NT / B_F4;    // k-rows (pixels) covered per load pass
    typedef float fragA __attribute__((ext_vector_type(MI)));
    typedef float fragB __attribute__((ext_vector_type(NI)));

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [BK][BM]   (pixel-major, like the tensors: no transpose anywhere)
    float* Bs = smem + BK * BM;       // [BK][BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    // opaque copy of the zero page's address (a known address lets hipcc load the page once and branch around the real loads)
    typedef const __attribute__((address_space(1))) float gfloat;      // keeps the gathers global_load (a generic pointer would make them flat_load)
    typedef const __attribute__((address_space(1))) f32x4 gf32x4;
    gfloat* zpage = (gfloat*)afi_zeros;
    asm volatile("" : "+v"(zpage));

    // t: logical tile id (the caller applies the XCD-aware remap); taps are fastest, then ci tiles: the blocks that re-read one dY
    // tile (9 taps x N tiles) and overlapping X rows sit behind the same L2 instead of pulling 8 copies through the fabric
    int tap, tile_n, tile_m;
    if (p.ntaps <= 9) {
        tap = t % p.ntaps; t /= p.ntaps;                   // taps fastest: the 9 blocks of one (m, n) tile share dY and most of X in L2
        tile_n = t % ntile_n; tile_m = t / ntile_n;
    } else {
        // Winograd planes share nothing with each other; inside a plane the N tiles of an M tile share the dY-side tile and the M
        // tiles of an N tile the X-side tile: planes slowest, so an XCD's run of blocks works through whole planes out of its L2
        // (planes fastest had every block stream both of its operand tiles from HBM: ~20 GB per launch on the largest layer)
        tile_n = t % ntile_n; t /= ntile_n;
        tile_m = t % ntile_m; tap = t / ntile_m;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    int dy = 0, dx = 0;
    if (p.ntaps == 9) { dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }

    const int HW = p.H * p.W;
    if (k_begin >= k_end) return;
    const int nK = (int)((k_end - k_begin + BK - 1) / BK);

    // ---- loader state: each load pass walks the pixels k_begin + kr + RPP*i + 32*stage.  (y, x) and the 64-bit element
    //      offset are advanced incrementally (adds of precomputed constants, no division / 64-bit multiply per stage).
    const int a_cq = tid % A_F4, a_kr = tid / A_F4;
    const int b_cq = tid % B_F4, b_kr = tid / B_F4;
    const int a_col = m0 + 4 * a_cq;                       // co' of this thread's float4
    int a_ph = 0, a_ch = a_col;
    if (p.dy_up == 2) { a_ph = a_col / p.CoutPhase; a_ch = a_col - a_ph * p.CoutPhase; }
    const bool a_col_ok = a_col < p.Mrows;
    const int b_col = n0 + 4 * b_cq;
    const bool b_col_ok = b_col < p.Ncols;
    const long long a_eH = (long long)p.dy_up * p.DY.sH, a_eW = (long long)p.dy_up * p.DY.sW;
    const int adv_y = BK / p.W, adv_x = BK - adv_y * p.W;
    const bool single_wrap = adv_y + 1 <= p.H;             // a stage of 32 pixels crosses at most one image boundary
    const long long a_adv = adv_y * a_eH + adv_x * a_eW, a_wrapx = a_eH - p.W * a_eW, a_wrapy = p.DY.sN - p.H * a_eH;
    const long long b_eH = (long long)p.x_stride * p.X.sH, b_eW = (long long)p.x_stride * p.X.sW;   // stride-2 conv: X is [N, xH, xW]
    const long long b_adv = adv_y * b_eH + adv_x * b_eW, b_wrapx = b_eH - p.W * b_eW, b_wrapy = p.X.sN - p.H * b_eH;

    long long a_off[A_LOADS], b_off[B_LOADS];
    int ay[A_LOADS], ax[A_LOADS], by[B_LOADS], bx[B_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const long long pix = k_begin + a_kr + A_RPP * i;
        const int n = (int)(pix / HW); const int rem = (int)(pix - (long long)n * HW);
        ay[i] = rem / p.W; ax[i] = rem - ay[i] * p.W;
        a_off[i] = (long long)n * p.DY.sN + ay[i] * a_eH + ax[i] * a_eW + (a_ph >> 1) * p.DY.sH + (a_ph & 1) * p.DY.sW + a_ch + (long long)tap * p.dy_sTap;
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
        const long long pix = k_begin + b_kr + B_RPP * i;
        const int n = (int)(pix / HW); const int rem = (int)(pix - (long long)n * HW);
        by[i] = rem / p.W; bx[i] = rem - by[i] * p.W;
        b_off[i] = (long long)n * p.X.sN + by[i] * b_eH + dy * p.X.sH + bx[i] * b_eW + dx * p.X.sW + b_col + (long long)tap * p.x_sTap;
    }
    long long k_pix = k_begin;                             // first pixel of the NEXT stage to gather

    f32x4 a_reg[A_LOADS], b_reg[B_LOADS];
    auto prefetch = [&](bool more) {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const bool ok = more && a_col_ok && (k_pix + a_kr + A_RPP * i < k_end);
            gfloat* src = ok ? (gfloat*)(p.DY.p + a_off[i]) : zpage;
            a_reg[i] = *(gf32x4*)src;
            ax[i] += adv_x; ay[i] += adv_y; a_off[i] += a_adv;
            if (ax[i] >= p.W) { ax[i] -= p.W; ++ay[i]; a_off[i] += a_wrapx; }
            if (single_wrap) { if (ay[i] >= p.H) { ay[i] -= p.H; a_off[i] += a_wrapy; } }      // (uniform) one image wrap at most: no exec-masked loop in the MFMA stream
            else while (ay[i] >= p.H) { ay[i] -= p.H; a_off[i] += a_wrapy; }
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int yy = by[i] * p.x_stride + dy, xx = bx[i] * p.x_stride + dx;
            const bool ok = more && b_col_ok && (k_pix + b_kr + B_RPP * i < k_end) && (unsigned)yy < (unsigned)p.xH && (unsigned)xx < (unsigned)p.xW;
            gfloat* src = ok ? (gfloat*)(p.X.p + b_off[i]) : zpage;
            b_reg[i] = *(gf32x4*)src;
            bx[i] += adv_x; by[i] += adv_y; b_off[i] += b_adv;
            if (bx[i] >= p.W) { bx[i] -= p.W; ++by[i]; b_off[i] += b_wrapx; }
            if (single_wrap) { if (by[i] >= p.H) { by[i] -= p.H; b_off[i] += b_wrapy; } }
            else while (by[i] >= p.H) { by[i] -= p.H; b_off[i] += b_wrapy; }
        }
        k_pix += BK;
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) *(f32x4*)(As + (a_kr + A_RPP * i) * BM + 4 * a_cq) = a_reg[i];
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) *(f32x4*)(Bs + (b_kr + B_RPP * i) * BN + 4 * b_cq) = b_reg[i];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // Fragment reads: a wave's MI (NI) 32-row blocks are INTERLEAVED (block mi = rows base + MI*lane + mi), so one
    // ds_read_b64 / b128 of MI consecutive floats feeds all MI blocks of a k-step (both operands are pixel-major, i.e.
    // row-contiguous in LDS; per-row ds_read_b32 would need MI+NI LDS instructions per k-step instead of 2).
    const float* a_rd = As + (wm * MI * 32 + MI * lr);
    const float* b_rd = Bs + (wn * NI * 32 + NI * lr);
    prefetch(true);
    for (int kc = 0; kc < nK; ++kc) {
        stage_store();
        __syncthreads();
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
            fragA a[4]; fragB b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = *(const fragA*)(a_rd + (s * 8 + lh * 4 + j) * BM);
                b[j] = *(const fragB*)(b_rd + (s * 8 + lh * 4 + j) * BN);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][mi], b[j][ni], acc[mi][ni], 0, 0, 0);
                    }
            if (s == 1) prefetch(kc + 1 < nK);              // next stage's gather in the middle of this stage's MFMAs
        }
        __syncthreads();
    }

    // dW += alpha * acc.  The descriptor fields are copied out first and, on a tile this block owns alone, ALL old values are loaded before
    // the first store: written as `*dst += v` per element the compiler emitted load / s_waitcnt vmcnt(0) / store (and re-read the
    // descriptor from the argument block) for each of a thread's elements in turn -- serial round trips at the end of every tile.
    const int pM = p.Mrows, pN = p.Ncols;
    const float alpha = p.alpha;
    float* const base = p.DW + (long long)tap * p.dw_sTap;
    const long long sRow = p.dw_sRow;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        float old[16][NI];
        if (!use_atomic) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * MI * 32 + MI * ((r & 3) + 8 * (r >> 2) + 4 * lh) + mi;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int col = n0 + wn * NI * 32 + NI * lr + ni;
                    old[r][ni] = (row < pM && col < pN) ? base[(long long)row * sRow + col] : 0.f;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * MI * 32 + MI * ((r & 3) + 8 * (r >> 2) + 4 * lh) + mi;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int col = n0 + wn * NI * 32 + NI * lr + ni;
                if (row < pM && col < pN) {
                    float* dst = base + (long long)row * sRow + col;
                    const float v = alpha * acc[mi][ni][r];
                    if (use_atomic) atomicAdd(dst, v); else *dst = old[r][ni] + v;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void afi_wgrad_gemm_body(const AfiWgradGemm& p, int ntile_m, int ntile_n, int kper, int t, int ksplit, bool use_atomic) {
    const long long P = (long long)p.N * p.H * p.W;
    const long long k_begin = (long long)ksplit * kper;
    const long long k_end = (k_begin + kper < P) ? k_begin + kper : P;
    afi_wgrad_gemm_range<BM, BN, WM, WN>(p, ntile_m, ntile_n, t, k_begin, k_end, use_atomic);
}
