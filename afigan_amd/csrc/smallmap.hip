// Small-map schedule of the AF interpolator (SURVEY 8a rows 2-9 at config-1 sizes: 1 x 256 x 25 x 34 -> 50 x 68).
//
// At 850 .. 3400 pixels a 3x3 conv is a GEMM with M = 850, N = 32 .. 1024, K = 2304 .. 9216: 7 .. 56 output tiles for 256 CUs,
// so whole tiles cannot fill the chip and a fixed split-K quantises badly (56 tiles x 5 splits = 280 blocks = 1.09 per CU).
// This file holds the three pieces of the small-map schedule:
//
//   afi_pix_gemm_sk_kernel   STREAM-K form of the pixel GEMM (same descriptor, gather, LDS layout and MFMA loop as
//                            afi_pix_gemm_kernel): the work is the list of (tile, K-stage) units; each of G persistent blocks
//                            takes an equal contiguous run of units, so every CU carries the same number of MFMAs whatever the
//                            shape; a block whose run ends inside a tile leaves a raw partial tile in a slab.  Blocks reach their
//                            prologue / MFMA / store phases at different times (runs start at arbitrary stages), which removes the
//                            lock-step of the one-tile-per-block split-K grid, where all blocks load, multiply and store together.
//   afi_pix_sk_reduce_kernel sums a tile's slabs in a FIXED order (bit-reproducible, no atomics) and applies the fused epilogue.
//   afi_wgrad_group_sk_kernel   every weight gradient of one backward pass (19 convs, 23 parameter tensors) in ONE launch: the blocks
//                            walk a table of AfiWgradGemm problems, each block owns a whole dW tile (all pixels), so there are no
//                            atomics, no zero-fills, and the grid fills the chip (the per-layer launches were 7 .. 36 tiles each).
//   afi_colsum_group_kernel  the bias gradients of the same pass in one launch.
#include "afi_common.h"

#define AFI_BK 32
// zero page of the branch-free gathers.  EXTERNAL linkage and non-const on purpose: with internal linkage hipcc proves the array is
// never written, folds loads from it to 0.0f, and turns every `ok ? ptr : zeros` address select into an exec-masked branch around
// the load plus `s_waitcnt vmcnt(0)` (seen in the ISA: the K loop then waits for the loads it has just issued, 2x its matrix time)
__device__ __attribute__((aligned(16))) float afi_zeros_smallmap[4] = {0.f, 0.f, 0.f, 0.f};
#define afi_zeros afi_zeros_smallmap

#include "afi_epilogue.h"
#include "afi_wgrad_body.h"
#include "afi_bf16_split.h"
#include "afi_bilinear.h"
#include "afi_convt_pack.h"
#include <stdlib.h>
#include <stdio.h>
#include <vector>
#include <mutex>
#include <set>
#include <type_traits>

// stream-K partition: unit = one BK-deep K stage of one tile; tile t owns units [t*nK, (t+1)*nK); logical block b owns
// [b*U/G, (b+1)*U/G).  G <= U, so no block is empty and the blocks that touch a tile are consecutive.  The launcher prefers a G
// for which every block's run is one equal slice of ONE tile (q = U/G divides nK): then a block is a single segment.
// Everything is 32-bit (small maps: U * G < 2^31) and the two divisions per row go through host-made reciprocals: at one or two
// waves per SIMD every VALU instruction of the prologue costs 4+ cycles in which the matrix pipe idles.
struct AfiSkArgs {
    int ntile_m, ntile_n;      // tiles of BM x BN (tile id = tile_n * ntile_m + tile_m: M fastest, a weight panel stays in L2)
    int nK;                    // K stages per tile = ntaps * nKphase * ceil(Ck / 32)
    int G;                     // logical blocks == gridDim.x
    int U;                     // ntile_m * ntile_n * nK
    int bm, bn;                // tile shape (for the reduction pass)
    int M, HW;                 // pixels, pixels per image
    unsigned rcp_HW, rcp_W, rcp_taps;   // floor(2^32 / d) + 1: n / d == umulhi(n, rcp) for n * d < 2^32 (d == 1: see afi_udiv)
    int kph_shift;             // log2(nKphase)   (1 or 4 phases)
    unsigned long long* dbg;   // diagnostic build only (AFI_SK_DIAG): [G][10] cycle stamps per block; production launches pass nullptr and compile no stamp
};
__device__ __forceinline__ unsigned afi_udiv(unsigned n, unsigned d, unsigned rcp) { return d == 1 ? n : __umulhi(n, rcp); }
// the block that owns unit x: the largest b with b*U/G <= x
__device__ __forceinline__ int afi_sk_owner(int x, int G, int U) { return (int)(((unsigned)(x + 1) * (unsigned)G + (unsigned)U - 1u) / (unsigned)U) - 1; }

#define AFI_STAMP(i) do { if constexpr (DIAG) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) sk.dbg[(long long)blockIdx.x * 10 + (i)] = t_; } } while (0)
template <int BM, int BN, int WM, int WN, bool B_RC, bool DIAG = false>
__global__ __launch_bounds__(256, 3) void afi_pix_gemm_sk_kernel(const AfiPixGemm p, const AfiSkArgs sk) {
    constexpr int BK = AFI_BK, LDK = BK + 4, NT = 256, D = 3;
    constexpr int MI = BM / (32 * WM), NI = BN / (32 * WN);
    static_assert(WM * WN == 4 && MI == 1 && NI == 1, "4 waves, one 32x32 accumulator block each");
    constexpr int K_F4 = BK / 4, K_RPP = NT / K_F4;       // KC operands: float4 per row of a stage, rows per load pass
    constexpr int A_LOADS = BM / K_RPP;
    constexpr int B_F4 = BN / 4;
    constexpr int B_LOADS = B_RC ? (BK * B_F4) / NT : BN / K_RPP;
    static_assert(A_LOADS >= 1 && B_LOADS >= 1, "tile too small");
    constexpr int B_ROWS_PER_PASS = NT / B_F4;
    constexpr int A_TILE = BM * LDK, B_TILE = B_RC ? BK * BN : BN * LDK, STAGE = A_TILE + B_TILE;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // two stage buffers: ONE barrier per stage

    if constexpr (DIAG) { if (threadIdx.x == 0) sk.dbg[(long long)blockIdx.x * 10 + 6] = __builtin_amdgcn_s_memrealtime(); }
    AFI_STAMP(0);
    bool first_seg = true;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    const int M = sk.M, HW = sk.HW;
    const int Ck4 = (p.Ck + 3) & ~3;
    const int ldp = (p.Ncols + 3) & ~3;
    const int aq = tid % K_F4, ar = tid / K_F4;           // KC: float4 column, first row
    const int b_cq = tid % B_F4, b_kr = tid / B_F4;       // RC: float4 column, first k-row
    // the zero page's address, made opaque: knowing it, hipcc loads the page ONCE, keeps it as the default value of every gather and
    // wraps the real loads in exec-masked branches behind `s_waitcnt vmcnt(0)` (seen in the ISA) -- the prefetch depth was gone
    typedef const __attribute__((address_space(1))) float gfloat;      // keeps the gathers global_load (a generic pointer would make them flat_load)
    typedef const __attribute__((address_space(1))) f32x4 gf32x4;
    gfloat* zpage = (gfloat*)afi_zeros;
    asm volatile("" : "+v"(zpage));

    // blocks b and b + 8 share an XCD: give each XCD a contiguous range of logical blocks, i.e. of tiles (weight panels and
    // neighbouring activation rows then meet in one L2); bijective for any G
    int lb;
    {
        const int q8 = sk.G >> 3, r8 = sk.G & 7, xcd = blockIdx.x & 7;
        lb = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (int)(blockIdx.x >> 3);
    }
    int u = (int)((unsigned)lb * (unsigned)sk.U / (unsigned)sk.G);
    const int u_end = (int)((unsigned)(lb + 1) * (unsigned)sk.U / (unsigned)sk.G);

    while (u < u_end) {                                    // (uniform) one segment = the part of this block's run inside one tile
        const int t = (int)((unsigned)u / (unsigned)sk.nK);
        const int kc0 = u - t * sk.nK;
        const int nK = min(sk.nK - kc0, u_end - u);
        const int tile_n = (int)((unsigned)t / (unsigned)sk.ntile_m), tile_m = t - tile_n * sk.ntile_m;
        const int m0 = tile_m * BM, n0 = tile_n * BN;

        // ---- loader state: each thread decodes the rows IT gathers (no row table, no barrier): centre offset + 9-bit tap mask
        long long a_off[A_LOADS]; unsigned a_mask[A_LOADS];
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int m = m0 + ar + K_RPP * i;
            const unsigned img = afi_udiv((unsigned)m, (unsigned)HW, sk.rcp_HW);
            const int rem = m - (int)img * HW;
            const int y = (int)afi_udiv((unsigned)rem, (unsigned)p.W, sk.rcp_W), x = rem - y * p.W;
            unsigned mk = 1u;
            if (p.ntaps == 9) {
                const unsigned cm = ((unsigned)(x - p.a_sgn) < (unsigned)p.W ? 1u : 0u) | 2u | ((unsigned)(x + p.a_sgn) < (unsigned)p.W ? 4u : 0u);
                mk = ((unsigned)(y - p.a_sgn) < (unsigned)p.H ? cm : 0u) | (cm << 3) | ((unsigned)(y + p.a_sgn) < (unsigned)p.H ? (cm << 6) : 0u);
            }
            a_mask[i] = m < M ? mk : 0u;
            a_off[i] = (long long)(m < M ? (int)img : 0) * p.A.sN + (long long)(y * p.a_up) * p.A.sH + (long long)(x * p.a_up) * p.A.sW + 4 * aq;
        }
        long long b_off[B_LOADS]; unsigned b_okm[B_LOADS], b_tail[B_LOADS];
        const int tail_c0 = (p.Ck / BK) * BK;              // first channel of a partial last chunk (== Ck rounded down: no partial chunk -> never reached with valid lanes masked)
        const unsigned a_tail = (tail_c0 + 4 * aq) < Ck4 ? 1u : 0u;
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            if constexpr (!B_RC) {
                const int n = n0 + ar + K_RPP * i;
                b_okm[i] = n < p.Ncols ? 1u : 0u; b_tail[i] = 1u;
                b_off[i] = (long long)n * p.b_sRow + 4 * aq;
            } else {
                const int n = n0 + 4 * b_cq;
                b_okm[i] = n < p.Ncols ? 1u : 0u;
                b_tail[i] = (tail_c0 + b_kr + B_ROWS_PER_PASS * i) < p.Ck ? 1u : 0u;
                b_off[i] = (long long)(b_kr + B_ROWS_PER_PASS * i) * p.b_sRow + n;
            }
        }

        // D register sets: a stage's gather is issued D stages before it is written to LDS.  At these sizes nearly every load is the
        // first touch of its line in this XCD's L2 (served by the Infinity Cache: ~2000 cycles under load, measured with cycle stamps),
        // i.e. two stages' worth of MFMAs; one set in flight left the loop at 2x its matrix time
        f32x4 a_reg[D][A_LOADS], b_reg[D][B_LOADS];
        // K order: channel chunk outermost, then phase, tap innermost (the taps of one chunk re-read the same pixels: L2 / L1 hits)
        const int kq = (int)afi_udiv((unsigned)kc0, (unsigned)p.ntaps, sk.rcp_taps);
        int k_tap = kc0 - kq * p.ntaps, k_kph = kq & (p.nKphase - 1), k_c0 = (kq >> sk.kph_shift) * BK;   // NEXT stage to gather
        auto stage_advance = [&]() {
            if (++k_tap == p.ntaps) {
                k_tap = 0;
                if (++k_kph == p.nKphase) { k_kph = 0; k_c0 += BK; }
            }
        };
        // Validity is kept as integer bit masks and every gather is ONE address select + ONE load: a bool && chain with a uniform and
        // a per-lane part makes hipcc build if/else around the loads, both arms writing the same registers behind `s_waitcnt vmcnt(0)`
        // (which drains the whole prefetch ring).  Only the last channel chunk can be partial: tail_c0 / *_tail describe it.
        auto load_all = [&](int set, bool more) {          // gather of stage (k_c0, k_kph, k_tap); !more: past the segment's end, every lane reads the zero page
            int dy = 0, dx = 0;
            if (p.ntaps == 9) { dy = k_tap / 3 - 1; dx = k_tap - (k_tap / 3) * 3 - 1; }
            const long long a_delta = (long long)(dy * p.a_sgn * p.a_up + (k_kph >> 1)) * p.A.sH + (long long)(dx * p.a_sgn * p.a_up + (k_kph & 1)) * p.A.sW + k_c0;
            const bool is_tail = k_c0 >= tail_c0;          // (uniform)
            // `more` goes through a VGPR: the loads must be issued UNCONDITIONALLY (a uniform branch around them makes the number of
            // loads in flight path-dependent, and hipcc then falls back from counted vmcnt(N) waits to vmcnt(0) at every LDS store)
            unsigned mm = more ? 1u : 0u;
            asm volatile("" : "+v"(mm));
            const unsigned ta = (is_tail ? a_tail : 1u) & mm;
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                const unsigned ok = (a_mask[i] >> k_tap) & ta;
                gfloat* src = ok ? (gfloat*)(p.A.p + (a_off[i] + a_delta)) : zpage;   // branch-free: masked lanes read zeros
                a_reg[set][i] = *(gf32x4*)src;
            }
            if constexpr (!B_RC) {
                const long long b_delta = (long long)k_tap * p.b_sTap + k_c0;
#pragma unroll
                for (int i = 0; i < B_LOADS; ++i) {
                    const unsigned ok = b_okm[i] & ta;
                    gfloat* src = ok ? (gfloat*)(p.B + (b_off[i] + b_delta)) : zpage;
                    b_reg[set][i] = *(gf32x4*)src;
                }
            } else {
                const long long b_delta = (long long)(k_kph * p.Ck + k_c0) * p.b_sRow + (long long)k_tap * p.b_sTap;
#pragma unroll
                for (int i = 0; i < B_LOADS; ++i) {
                    const unsigned ok = b_okm[i] & (is_tail ? b_tail[i] : 1u) & mm;
                    gfloat* src = ok ? (gfloat*)(p.B + (b_off[i] + b_delta)) : zpage;
                    b_reg[set][i] = *(gf32x4*)src;
                }
            }
        };
        auto stage_store = [&](int set, int buf) {
            float* As = smem + buf * STAGE;
            float* Bs = As + A_TILE;
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) *(f32x4*)(As + (ar + K_RPP * i) * LDK + 4 * aq) = a_reg[set][i];
            if constexpr (!B_RC) {
#pragma unroll
                for (int i = 0; i < B_LOADS; ++i) *(f32x4*)(Bs + (ar + K_RPP * i) * LDK + 4 * aq) = b_reg[set][i];
            } else {
#pragma unroll
                for (int i = 0; i < B_LOADS; ++i) *(f32x4*)(Bs + (b_kr + B_ROWS_PER_PASS * i) * BN + 4 * b_cq) = b_reg[set][i];
            }
        };

        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;

        // double-buffered LDS, D register sets:  stage k's MFMAs read buffer k & 1;  behind them the registers of stage k+1 (gathered D
        // stages earlier) go to the other buffer, the gather of stage k+1+D is issued into the freed set, ONE barrier closes the stage
        if (first_seg) AFI_STAMP(1);
#pragma unroll
        for (int d = 0; d < D; ++d) { load_all(d, d < nK); stage_advance(); }
        if (first_seg) AFI_STAMP(2);
        stage_store(0, 0);
        load_all(0, D < nK); stage_advance();
        __syncthreads();
        if (first_seg) AFI_STAMP(3);
        for (int kb = 0; kb < nK; kb += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int kc = kb + d;
                if (kc >= nK) break;                       // (uniform)
                const float* As = smem + (kc & 1) * STAGE;
                const float* Bs = As + A_TILE;
                // two fragment register sets: slice s+1 is read from LDS while the MFMAs of slice s issue
                f32x4 fa[2], fb[2];
                auto frag = [&](int set, int s) {
                    fa[set] = *(const f32x4*)(As + (wm * 32 + lr) * LDK + s * 8 + lh * 4);
                    if constexpr (!B_RC) {
                        fb[set] = *(const f32x4*)(Bs + (wn * 32 + lr) * LDK + s * 8 + lh * 4);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) fb[set][j] = Bs[(s * 8 + lh * 4 + j) * BN + wn * 32 + lr];
                    }
                };
                frag(0, 0);
#pragma unroll
                for (int s = 0; s < BK / 8; ++s) {
                    if (s + 1 < BK / 8) frag((s + 1) & 1, s + 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][j], fb[s & 1][j], acc, 0, 0, 0);
                    // behind the first k-slice's MFMAs (in their shadow, not after the last one): next stage -> other buffer, refill the set
                    if (s == 0) {                          // (compile time) unconditional, also behind the last stage: its store lands in the idle buffer
                        stage_store((d + 1) % D, (kc + 1) & 1);
                        load_all((d + 1) % D, kc + 1 + D < nK); stage_advance();
                    }
                }
                __syncthreads();
            }
        }
        if (first_seg) AFI_STAMP(4);

        const int tile_first = t * sk.nK;
        const bool whole = (kc0 == 0 && nK == sk.nK);      // (uniform) this block multiplied the tile's whole K range: fused epilogue, no slab
        if (!whole) {
            // partial tile: straight from the accumulators into this block's slab (a lane holds one column, 16 rows: every store
            // instruction writes two 128-B row segments; the slab layout is ours, the reduction pass reads it back as float4 rows)
            const int slot = lb - afi_sk_owner(tile_first, sk.G, sk.U);
            float* slab = p.partial + (long long)slot * M * ldp;
            // each wave turns its own 32x32 block through a private LDS patch into float4 rows (no block barrier: the loop's last
            // barrier freed the stage buffers, and a wave's LDS instructions execute in order)
            float* Cw = smem + wave * (32 * LDK);
#pragma unroll
            for (int r = 0; r < 16; ++r) Cw[((r & 3) + 8 * (r >> 2) + 4 * lh) * LDK + lr] = acc[r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int item = lane + 64 * i, rl = item >> 3, c4 = item & 7;
                const int m = m0 + wm * 32 + rl, col = n0 + wn * 32 + 4 * c4;
                if (m < M && col < ldp) *(f32x4*)(slab + (long long)m * ldp + col) = *(const f32x4*)(Cw + rl * LDK + 4 * c4);
            }
            __syncthreads();                                // the next segment's prologue refills the stage buffers
        } else {
            // whole tile: accumulators -> LDS -> float4 rows -> fused epilogue (the last barrier of the loop freed both buffers)
            constexpr int LDC = BN + 4, C_F4 = BN / 4;
            static_assert(BM * LDC <= 2 * STAGE, "C staging tile must fit in the operand buffers");
            float* Cs = smem;
#pragma unroll
            for (int r = 0; r < 16; ++r) Cs[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + wn * 32 + lr] = acc[r];
            __syncthreads();
            for (int item = tid; item < BM * C_F4; item += NT) {
                const int rl = item / C_F4, c4 = item - rl * C_F4;
                const int m = m0 + rl, col = n0 + 4 * c4;
                if (m >= M || col >= p.Ncols) continue;
                const unsigned img = afi_udiv((unsigned)m, (unsigned)HW, sk.rcp_HW);
                const int rem = m - (int)img * HW;
                const int y = (int)afi_udiv((unsigned)rem, (unsigned)p.W, sk.rcp_W), x = rem - y * p.W;
                afi_epilogue_store<false>(p, (int)img, y, x, col, *(const f32x4*)(Cs + rl * LDC + 4 * c4));
            }
            __syncthreads();
        }
        if (first_seg) AFI_STAMP(5);
        first_seg = false;
        u += nK;
    }
    AFI_STAMP(8);
    if constexpr (DIAG) { if (threadIdx.x == 0) sk.dbg[(long long)blockIdx.x * 10 + 7] = __builtin_amdgcn_s_memrealtime(); }
}

// Second pass: element (m, col) belongs to tile (m / bm, col / bn); its slabs are those of the consecutive blocks that own the
// tile's units, summed in block order, then the fused epilogue (bias, activation, residuals, bilinear skip, mask, pixel shuffle).
__global__ __launch_bounds__(256) void afi_pix_sk_reduce_kernel(const AfiPixGemm p, const AfiSkArgs sk, unsigned rcp_cf4, unsigned rcp_bm, unsigned rcp_bn) {
    const int M = sk.M, HW = sk.HW;
    const int ldp = (p.Ncols + 3) & ~3;
    const int C_F4 = ldp >> 2;
    const int total = M * C_F4;
    const long long slab = (long long)M * ldp;
    for (int it = (int)blockIdx.x * 256 + (int)threadIdx.x; it < total; it += (int)gridDim.x * 256) {
        const int m = (int)afi_udiv((unsigned)it, (unsigned)C_F4, rcp_cf4);
        const int col = (it - m * C_F4) * 4;
        const int t = (int)afi_udiv((unsigned)col, (unsigned)sk.bn, rcp_bn) * sk.ntile_m + (int)afi_udiv((unsigned)m, (unsigned)sk.bm, rcp_bm);
        const int b0 = afi_sk_owner(t * sk.nK, sk.G, sk.U), b1 = afi_sk_owner(t * sk.nK + sk.nK - 1, sk.G, sk.U);
        const int nparts = b1 - b0 + 1;
        const float* src = p.partial + (long long)m * ldp + col;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        int ks = 0;
        for (; ks + 4 <= nparts; ks += 4) {               // four slabs in flight (the pass is latency-bound), fixed order of the adds
            const f32x4 t0 = *(const f32x4*)(src + (ks + 0) * slab), t1 = *(const f32x4*)(src + (ks + 1) * slab);
            const f32x4 t2 = *(const f32x4*)(src + (ks + 2) * slab), t3 = *(const f32x4*)(src + (ks + 3) * slab);
            v += t0; v += t1; v += t2; v += t3;
        }
        for (; ks < nparts; ++ks) v += *(const f32x4*)(src + ks * slab);
        const unsigned img = afi_udiv((unsigned)m, (unsigned)HW, sk.rcp_HW);
        const int rem = m - (int)img * HW;
        const int y = (int)afi_udiv((unsigned)rem, (unsigned)p.W, sk.rcp_W), x = rem - y * p.W;
        afi_epilogue_store<false>(p, (int)img, y, x, col, v);
    }
}

static unsigned sk_rcp(unsigned d) { return d <= 1 ? 0u : (unsigned)((1ULL << 32) / d) + 1u; }

template <int BM, int BN, int WM, int WN, bool B_RC>
static int launch_sk(const AfiPixGemm& p, hipStream_t st) {
    constexpr int bpc = 2;                                 // persistent blocks per CU
    constexpr int minq = 4;                                // at least this many K stages per block
    const long long M = (long long)p.N * p.H * p.W;
    AfiSkArgs sk;
    sk.ntile_m = afi_cdiv(M, BM); sk.ntile_n = afi_cdiv(p.Ncols, BN);
    sk.nK = p.ntaps * p.nKphase * afi_cdiv(p.Ck, AFI_BK);
    const long long U = (long long)sk.ntile_m * sk.ntile_n * sk.nK;
    const long long Gcap = 256LL * bpc;
    if (U * Gcap >= (1LL << 31) || M * (long long)p.H * p.W >= (1LL << 32) || M * (((p.Ncols + 3) & ~3) >> 2) >= (1LL << 31)) return AFI_ERR_UNSUPPORTED;   // 32-bit index math
    sk.U = (int)U; sk.bm = BM; sk.bn = BN; sk.M = (int)M; sk.HW = p.H * p.W;
    sk.rcp_HW = sk_rcp((unsigned)sk.HW); sk.rcp_W = sk_rcp((unsigned)p.W); sk.rcp_taps = sk_rcp((unsigned)p.ntaps);
    sk.kph_shift = p.nKphase == 4 ? 2 : 0;
    if (p.nKphase != 1 && p.nKphase != 4) return AFI_ERR_UNSUPPORTED;
    const long long slab = M * ((p.Ncols + 3) & ~3);
    const long long ntiles = (long long)sk.ntile_m * sk.ntile_n;
    long long G;
    if (sk.nK <= 16 || ntiles >= Gcap) {
        G = ntiles;                                       // short K or enough tiles: whole tiles, fused epilogue, no second pass
    } else {
        // equal slices of ONE tile per block where a divisor of nK fits: splits = the largest s with s | nK, ntiles * s <= Gcap and
        // nK / s >= minq; otherwise the general stream-K partition with G = Gcap
        int best = 1;
        for (int s2 = 1; s2 <= sk.nK; ++s2)
            if (sk.nK % s2 == 0 && ntiles * s2 <= Gcap && sk.nK / s2 >= minq) best = s2;
        G = ntiles * best;
        if (G * 10 < Gcap * 8) {                          // the best divisor leaves > 20 % of the block slots empty: general partition
            G = Gcap;
            if (G > U / minq) G = U / minq;
            if (G < ntiles) G = ntiles;
        }
        for (;;) {                                        // the slabs of the most-shared tile must fit the caller's scratch
            const long long qmin = U / G;                 // fewest units a block owns
            const long long smax = (sk.nK + qmin - 1) / qmin + 1;
            if (G <= ntiles || smax * slab <= p.partial_floats) break;
            G -= (G - ntiles) > 8 ? 8 : (G - ntiles);
        }
    }
    sk.G = (int)G;
    const bool second_pass = G > ntiles;
    if (second_pass && (!p.partial || p.partial_floats <= 0)) return AFI_ERR_UNSUPPORTED;
    const size_t lds = sizeof(float) * 2 * (BM * (AFI_BK + 4) + (B_RC ? AFI_BK * BN : BN * (AFI_BK + 4)));
    sk.dbg = nullptr;
    hipLaunchKernelGGL((afi_pix_gemm_sk_kernel<BM, BN, WM, WN, B_RC>), dim3((unsigned)G), dim3(256), lds, st, p, sk);
    if (second_pass) {
        const int cf4 = ((p.Ncols + 3) & ~3) >> 2;
        const long long items = M * cf4;
        long long g = (items + 255) / 256; if (g > 2048) g = 2048;
        hipLaunchKernelGGL(afi_pix_sk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, st, p, sk, sk_rcp((unsigned)cf4), sk_rcp((unsigned)BM), sk_rcp((unsigned)BN));
    }
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

template <bool B_RC> static int launch_wk(const AfiPixGemm& p, hipStream_t st);
// Small-map form of a pixel GEMM: stream-K / even split-K with a slab reduction for long K, whole tiles with the fused epilogue for
// short K.  AFI_ERR_UNSUPPORTED = "not this path" (the caller falls back to the tiled kernels).
int afi_launch_pix_gemm_sk(const AfiPixGemm& p, int b_rc, hipStream_t st) {
    if (p.gtap || p.b_sImg != 0) return AFI_ERR_UNSUPPORTED;
    if (p.ntaps != 1 && p.ntaps != 9) return AFI_ERR_UNSUPPORTED;
    {   // K split inside the block (no second pass); the stream-K form below takes what its 32-bit index math refuses
        const int rc = b_rc ? launch_wk<true>(p, st) : launch_wk<false>(p, st);
        if (rc != AFI_ERR_UNSUPPORTED || p.Bimg) return rc; // (a problem that carries a weight image is defined by it: the kernels below would read B)
    }
    if (p.Ncols <= 32) return b_rc ? launch_sk<128, 32, 4, 1, true>(p, st) : launch_sk<128, 32, 4, 1, false>(p, st);
    return b_rc ? launch_sk<64, 64, 2, 2, true>(p, st) : launch_sk<64, 64, 2, 2, false>(p, st);
}

// ------------------------------------------------------------------------------------------------
// afi_pix_gemm_wk_kernel: the small-map pixel GEMM with the K split INSIDE the block.
//   One block = one 32 x 32 output tile = 8 waves (two per SIMD); wave w multiplies the tile's K stages [w*nK/8, (w+1)*nK/8) into its
//   own 32x32 accumulator block; the eight partial blocks meet in LDS and the fused epilogue stores the tile.  850 x 256 outputs are
//   216 tiles, so the grid fills the chip with whole tiles and there is NO cross-block reduction: no slabs, no second launch (the
//   stream-K form above needs one, 5 us each at these sizes: 24 launches per interpolator forward + backward).
//   Each wave stages its OWN operand slices (32 rows x 32 k of A and of B) through a private LDS patch, so the K loop has no
//   block barrier at all: global -> registers (two sets in flight) -> LDS -> all fragments of a stage in registers (two sets) ->
//   16 MFMAs; a wave's LDS instructions execute in order, which is all the synchronisation the loop needs.
//   Price: no operand sharing between waves, i.e. 2x the L2 -> CU traffic of a 64x64 tile (32 B/clk/CU at the matrix-pipe rate).
struct AfiWkArgs {
    int ntile_m, ntile_n, nK;  // tiles of 32 x 32 (tile id = tile_n * ntile_m + tile_m), K stages per tile
    int M, HW;
    unsigned rcp_HW, rcp_W, rcp_taps, rcp_ntm;
    int kph_shift;
    int nj;                    // bf16x6 kernel: 32-column halves... 1 = 32 x 32 tiles, 2 = 32 x 64 (ntile_n counts tiles of 32 nj columns)
    unsigned long long* dbg;   // diagnostic build only
};
// Fused epilogue with every optional operand loaded UP FRONT (address select to the zero page when a term is off, its scale then 0):
// the generic afi_epilogue_store reads O_old, R1, R2 and Z one after the other behind uniform branches, i.e. up to four memory
// latencies in a row -- 3 us at the end of a kernel whose whole K loop takes 1.5 us.  Same arithmetic, same order of the adds.
typedef const __attribute__((address_space(1))) float afi_gfloat;
typedef const __attribute__((address_space(1))) f32x4 afi_gf32x4;
struct AfiEpiPre {                                          // the optional operands of one output float4, loaded ahead of the accumulator
    float* dst; f32x4 bv, ov, r1, r2, zv; int ch; bool live, use_old, use_r1, use_r2, use_z;
};
__device__ __forceinline__ AfiEpiPre afi_epilogue_prefetch(const AfiPixGemm& p, int img, int y, int x, int col, afi_gfloat* zpage) {
    AfiEpiPre e;
    int phase = 0, ch = col;
    if (p.o_up == 2) { phase = col / p.CoutPhase; ch = col - phase * p.CoutPhase; }
    const int yo = y * p.o_up + (phase >> 1), xo = x * p.o_up + (phase & 1);
    e.ch = ch;
    e.live = !(yo >= p.oH || xo >= p.oW);
    e.dst = p.O.p + (long long)img * p.O.sN + (long long)yo * p.O.sH + (long long)xo * p.O.sW + ch;
    e.use_old = e.live && p.beta != 0.f;
    e.use_r1 = e.live && p.R1.p && !p.r1_bilinear && ch >= p.r1_lo && ch < p.r1_hi;
    e.use_r2 = e.live && p.R2.p && ch >= p.r2_lo && ch < p.r2_hi;
    e.use_z = e.live && p.Z.p && ch >= p.z_lo && ch < p.z_hi;
    afi_gfloat* a_b = (e.live && p.bias) ? (afi_gfloat*)(p.bias + ch) : zpage;
    afi_gfloat* a_o = e.use_old ? (afi_gfloat*)e.dst : zpage;
    afi_gfloat* a_1 = e.use_r1 ? (afi_gfloat*)(p.R1.p + (long long)img * p.R1.sN + (long long)yo * p.R1.sH + (long long)xo * p.R1.sW + ch) : zpage;
    afi_gfloat* a_2 = e.use_r2 ? (afi_gfloat*)(p.R2.p + (long long)img * p.R2.sN + (long long)yo * p.R2.sH + (long long)xo * p.R2.sW + ch) : zpage;
    afi_gfloat* a_z = e.use_z ? (afi_gfloat*)(p.Z.p + (long long)img * p.Z.sN + (long long)yo * p.Z.sH + (long long)xo * p.Z.sW + ch) : zpage;
    e.bv = *(afi_gf32x4*)a_b; e.ov = *(afi_gf32x4*)a_o; e.r1 = *(afi_gf32x4*)a_1; e.r2 = *(afi_gf32x4*)a_2; e.zv = *(afi_gf32x4*)a_z;
    return e;
}
__device__ __forceinline__ void afi_epilogue_finish(const AfiPixGemm& p, const AfiEpiPre& e, int img, int y, int x, f32x4 accv) {
    if (!e.live) return;
    const int ch = e.ch;
    f32x4 v = p.alpha * accv;
    v += e.bv;
    if (e.use_old) v += p.beta * e.ov;
    if (e.use_r1) v += p.r1s * e.r1;
    if (p.R1.p && p.r1_bilinear && ch >= p.r1_lo && ch < p.r1_hi) {      // (uniform) only the interpolator's last conv
        int by0, by1, bx0, bx1; float ly, lx;
        afi_bil_idx(y, p.H >> 1, by0, by1, ly); afi_bil_idx(x, p.W >> 1, bx0, bx1, lx);
        const float* rb = p.R1.p + (long long)img * p.R1.sN + ch;
        const f32x4 x00 = *(const f32x4*)(rb + (long long)by0 * p.R1.sH + (long long)bx0 * p.R1.sW);
        const f32x4 x01 = *(const f32x4*)(rb + (long long)by0 * p.R1.sH + (long long)bx1 * p.R1.sW);
        const f32x4 x10 = *(const f32x4*)(rb + (long long)by1 * p.R1.sH + (long long)bx0 * p.R1.sW);
        const f32x4 x11 = *(const f32x4*)(rb + (long long)by1 * p.R1.sH + (long long)bx1 * p.R1.sW);
        const f32x4 top = x00 * (1.f - lx) + x01 * lx;
        const f32x4 bot = x10 * (1.f - lx) + x11 * lx;
        v += p.r1s * (top * (1.f - ly) + bot * ly);
    }
    if (e.use_r2) v += p.r2s * e.r2;
    if (p.lrelu) {
        const float slope = (p.lrelu == 1) ? AFI_LRELU_SLOPE : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
    }
    if (e.use_z) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= (e.zv[j] > 0.f ? 1.f : AFI_LRELU_SLOPE);
    }
    *(f32x4*)e.dst = v;
}
__device__ __forceinline__ void afi_epilogue_store_fast(const AfiPixGemm& p, int img, int y, int x, int col, f32x4 accv, afi_gfloat* zpage) {
    const AfiEpiPre e = afi_epilogue_prefetch(p, img, y, x, col, zpage);
    afi_epilogue_finish(p, e, img, y, x, accv);
}

// LEAN: one register set in flight and one fragment set (<= 128 registers): TWO blocks per CU = four waves per SIMD, whose interleaving
// covers the memory latency instead of the second register set; for grids with more blocks than CUs (N = 384 .. 1024 columns)
template <bool B_RC, bool LEAN, bool DIAG>
__device__ __forceinline__ void afi_wk_body(const AfiPixGemm& p, const AfiWkArgs& sk, const int lb) {
    constexpr int BK = AFI_BK, LDK = BK + 4, NW = 8, D = LEAN ? 1 : 2, FS = LEAN ? 1 : 2;
    constexpr int A_TILE = 32 * LDK, PATCH = 2 * 32 * LDK;   // per-wave LDS patch: A [32][LDK] + B ([32][LDK] or [32 k][32 n]); >= one 32 x LDK partial tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    if constexpr (DIAG) { if (threadIdx.x == 0) sk.dbg[(long long)blockIdx.x * 10 + 6] = __builtin_amdgcn_s_memrealtime(); }
    AFI_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int q8 = lane & 7, r8 = lane >> 3;               // staging: float4 column, first row (8 rows per pass, 4 passes)
    const int M = sk.M, HW = sk.HW;
    const int Ck4 = (p.Ck + 3) & ~3;
    // Gathers are raw BUFFER loads: 32-bit byte offsets (the launcher checks both operands stay below 2 GB), one v_add + one
    // v_cndmask per load, and a masked lane's offset (0xFFFFFFFF) is out of range: the hardware returns zeros without touching memory
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A.p, 0, 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0x7FFFFFF0, 0x00020000);
    const int tile_n = (int)afi_udiv((unsigned)lb, (unsigned)sk.ntile_m, sk.rcp_ntm), tile_m = lb - tile_n * sk.ntile_m;
    const int m0 = tile_m * 32, n0 = tile_n * 32;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int kc0 = wave_u * sk.nK / NW, kc1 = (wave_u + 1) * sk.nK / NW;   // this wave's K stages
    const int nK = kc1 - kc0;
    float* As = smem + wave_u * PATCH;
    float* Bs = As + A_TILE;

    // ---- loader state (per lane: 4 rows of A, 4 rows of B): byte offsets, validity as bit masks
    unsigned a_off[4], a_mask[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + r8 + 8 * i;
        const unsigned img = afi_udiv((unsigned)m, (unsigned)HW, sk.rcp_HW);
        const int rem = m - (int)img * HW;
        const int y = (int)afi_udiv((unsigned)rem, (unsigned)p.W, sk.rcp_W), x = rem - y * p.W;
        unsigned mk = 1u;
        if (p.ntaps == 9) {
            const unsigned cm = ((unsigned)(x - p.a_sgn) < (unsigned)p.W ? 1u : 0u) | 2u | ((unsigned)(x + p.a_sgn) < (unsigned)p.W ? 4u : 0u);
            mk = ((unsigned)(y - p.a_sgn) < (unsigned)p.H ? cm : 0u) | (cm << 3) | ((unsigned)(y + p.a_sgn) < (unsigned)p.H ? (cm << 6) : 0u);
        }
        a_mask[i] = m < M ? mk : 0u;
        a_off[i] = 4u * (unsigned)((m < M ? (int)img : 0) * (int)p.A.sN + (y * p.a_up) * (int)p.A.sH + (x * p.a_up) * (int)p.A.sW + 4 * q8);
    }
    const int tail_c0 = (p.Ck / BK) * BK;                  // first channel of a partial last chunk
    const unsigned a_tail = (tail_c0 + 4 * q8) < Ck4 ? 1u : 0u;
    unsigned b_off[4], b_okm[4], b_tail[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (!B_RC) {
            const int n = n0 + r8 + 8 * i;
            b_okm[i] = n < p.Ncols ? 1u : 0u; b_tail[i] = 1u;
            b_off[i] = 4u * (unsigned)(n * (int)p.b_sRow + 4 * q8);
        } else {
            const int n = n0 + 4 * q8;
            b_okm[i] = n < p.Ncols ? 1u : 0u;
            b_tail[i] = (tail_c0 + r8 + 8 * i) < p.Ck ? 1u : 0u;
            b_off[i] = 4u * (unsigned)((r8 + 8 * i) * (int)p.b_sRow + n);
        }
    }
    const int kq = (int)afi_udiv((unsigned)kc0, (unsigned)p.ntaps, sk.rcp_taps);
    int k_tap = kc0 - kq * p.ntaps, k_kph = kq & (p.nKphase - 1), k_c0 = (kq >> sk.kph_shift) * BK;   // NEXT stage to gather
    auto stage_advance = [&]() {
        if (++k_tap == p.ntaps) {
            k_tap = 0;
            if (++k_kph == p.nKphase) { k_kph = 0; k_c0 += BK; }
        }
    };
    u32x4 a_reg[D][4], b_reg[D][4];
    auto load_all = [&](int set, bool more) {              // the loads are issued UNCONDITIONALLY (counted vmcnt waits need a path-independent count)
        int dy = 0, dx = 0;
        if (p.ntaps == 9) { dy = k_tap / 3 - 1; dx = k_tap - (k_tap / 3) * 3 - 1; }
        const int a_delta = 4 * ((dy * p.a_sgn * p.a_up + (k_kph >> 1)) * (int)p.A.sH + (dx * p.a_sgn * p.a_up + (k_kph & 1)) * (int)p.A.sW + k_c0);
        const bool is_tail = k_c0 >= tail_c0;
        unsigned mm = more ? 1u : 0u;
        asm volatile("" : "+v"(mm));
        const unsigned ta = (is_tail ? a_tail : 1u) & mm;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned ok = (a_mask[i] >> k_tap) & ta;
            a_reg[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ok ? a_off[i] + (unsigned)a_delta : 0xFFFFFFFFu, 0, 0);
        }
        if constexpr (!B_RC) {
            const int b_delta = 4 * (k_tap * (int)p.b_sTap + k_c0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned ok = b_okm[i] & ta;
                b_reg[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, ok ? b_off[i] + (unsigned)b_delta : 0xFFFFFFFFu, 0, 0);
            }
        } else {
            const int b_delta = 4 * ((k_kph * p.Ck + k_c0) * (int)p.b_sRow + k_tap * (int)p.b_sTap);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned ok = b_okm[i] & (is_tail ? b_tail[i] : 1u) & mm;
                b_reg[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, ok ? b_off[i] + (unsigned)b_delta : 0xFFFFFFFFu, 0, 0);
            }
        }
    };
    auto stage_store = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(u32x4*)(As + (r8 + 8 * i) * LDK + 4 * q8) = a_reg[set][i];
        if constexpr (!B_RC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *(u32x4*)(Bs + (r8 + 8 * i) * LDK + 4 * q8) = b_reg[set][i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) *(u32x4*)(Bs + (r8 + 8 * i) * 32 + 4 * q8) = b_reg[set][i];
        }
    };
    f32x4 fa[FS][4], fb[FS][4];                            // all fragments of a stage (two sets: the next stage's are read under this stage's MFMAs)
    auto read_frags = [&](int set) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            fa[set][s] = *(const f32x4*)(As + lr * LDK + s * 8 + lh * 4);
            if constexpr (!B_RC) {
                fb[set][s] = *(const f32x4*)(Bs + lr * LDK + s * 8 + lh * 4);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[set][s][j] = Bs[(s * 8 + lh * 4 + j) * 32 + lr];
            }
        }
    };
    auto lds_order = [&]() {                               // (lgkmcnt only) a wave's LDS instructions execute in order; this pins the compiler's order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    AFI_STAMP(1);
#pragma unroll
    for (int d = 0; d < D; ++d) { load_all(d, d < nK); stage_advance(); }
    AFI_STAMP(2);
    stage_store(0);
    lds_order();
    read_frags(0);
    load_all(0, D < nK); stage_advance();
    AFI_STAMP(3);
    if constexpr (!LEAN) {
        for (int kb = 0; kb < nK; kb += 2) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int kc = kb + d;
                if (kc >= nK) break;                       // (uniform per wave)
                // stage kc's fragments are in registers (set d): the patch is free for stage kc+1, whose registers were gathered D stages ago
                stage_store((d + 1) % D);
                lds_order();
                read_frags((d + 1) & 1);
                load_all((d + 1) % D, kc + 1 + D < nK); stage_advance();
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[d][s][j], fb[d][s][j], acc, 0, 0, 0);
            }
        }
    } else {
        for (int kc = 0; kc < nK; ++kc) {
            stage_store(0);                                // stage kc+1 (gathered one stage ago; the other three waves of this SIMD ran meanwhile)
            load_all(0, kc + 2 < nK); stage_advance();
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][s][j], fb[0][s][j], acc, 0, 0, 0);
            lds_order();
            read_frags(0);
        }
    }
    AFI_STAMP(4);
    // ---- the eight partial blocks meet in LDS (each wave writes into its own patch), then 256 threads sum and run the fused epilogue
    lds_order();
    // the epilogue's own operands (old value, residuals, mask, bias) are requested BEFORE the partial tiles meet in LDS: their global
    // round trip (~1.5 us) runs under the LDS write, the barrier and the eight partial reads instead of after them
    const int e_rl = tid >> 3, e_c4 = tid & 7;
    const int e_m = m0 + e_rl, e_col = n0 + 4 * e_c4;
    const bool e_on = tid < 256 && e_m < M && e_col < p.Ncols;
    AfiEpiPre epre; int e_img = 0, e_y = 0, e_x = 0;
    if (e_on) {
        afi_gfloat* zpage = (afi_gfloat*)afi_zeros;
        asm volatile("" : "+v"(zpage));
        const unsigned img = afi_udiv((unsigned)e_m, (unsigned)HW, sk.rcp_HW);
        const int rem = e_m - (int)img * HW;
        e_y = (int)afi_udiv((unsigned)rem, (unsigned)p.W, sk.rcp_W); e_x = rem - e_y * p.W; e_img = (int)img;
        epre = afi_epilogue_prefetch(p, e_img, e_y, e_x, e_col, zpage);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) As[((r & 3) + 8 * (r >> 2) + 4 * lh) * LDK + lr] = acc[r];
    __syncthreads();
    if (e_on) {
        f32x4 v = *(const f32x4*)(smem + e_rl * LDK + 4 * e_c4);
#pragma unroll
        for (int w = 1; w < NW; ++w) v += *(const f32x4*)(smem + w * PATCH + e_rl * LDK + 4 * e_c4);     // fixed order: bit-reproducible
        afi_epilogue_finish(p, epre, e_img, e_y, e_x, v);
    }
    AFI_STAMP(5);
    AFI_STAMP(8);
    if constexpr (DIAG) { if (threadIdx.x == 0) sk.dbg[(long long)blockIdx.x * 10 + 7] = __builtin_amdgcn_s_memrealtime(); }
}

__device__ __forceinline__ int afi_xcd_logical_id() {       // XCD-contiguous logical block ids (bijective for any grid)
    const int nwg = gridDim.x, qq = nwg >> 3, rr = nwg & 7, xcd = blockIdx.x & 7;
    return (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (int)(blockIdx.x >> 3);
}
template <bool B_RC, bool LEAN, bool DIAG = false>
__global__ __launch_bounds__(512, LEAN ? 4 : 2) void afi_pix_gemm_wk_kernel(const AfiPixGemm p, const AfiWkArgs sk) {
    afi_wk_body<B_RC, LEAN, DIAG>(p, sk, afi_xcd_logical_id());
}
// Several GEMMs that are ready at the same time in ONE launch (the dense block in column-batched form: the contributions of one
// source slice to every later conv of the block).  Tiles are numbered problem-major; a block finds its problem in a 5-entry table.
#define AFI_WK_MAXP 5
struct AfiWkGroup {
    int nprob;
    int tile_start[AFI_WK_MAXP + 1];
    AfiWkArgs wk[AFI_WK_MAXP];
    AfiPixGemm p[AFI_WK_MAXP];
};
template <bool B_RC>
__global__ __launch_bounds__(512, 4) void afi_pix_gemm_wk_group_kernel(const AfiWkGroup grp) {
    const int t = afi_xcd_logical_id();
    int pi = 0;
    while (pi + 1 < grp.nprob && t >= grp.tile_start[pi + 1]) ++pi;     // (uniform)
    afi_wk_body<B_RC, true, false>(grp.p[pi], grp.wk[pi], t - grp.tile_start[pi]);
}

// ------------------------------------------------------------------------------------------------
// afi_pix_gemm_wk6: the same block structure (one 32 x 32 output tile, eight waves each multiplying an eighth of K, partial tiles meeting
// in LDS, fused epilogue) on the bf16 matrix cores in the six-product form of csrc/afi_gemm_bf16.h: every fp32 operand is x = hi + mid + lo
// exactly (three bf16), the six products of relative size >= 2^-16 are summed smallest first into fp32 accumulators.  fp32 results at
// 6/16 of the fp32-MFMA pipe time: 24 v_mfma_f32_16x16x32_bf16 (384 cycles) per 32-deep stage of a wave instead of 16
// v_mfma_f32_32x32x2_f32 (1024 cycles).
//   * B (weights) arrives PRE-SPLIT in fragment order (AfiPixGemm::Bimg, built by afi_wk6_image_kernel once per weight and pass, or once
//     per optimizer step when the caller registers a weight cache): a lane's operand of one (n half, part) is ONE 16-byte load straight
//     into the register the MFMA reads -- no LDS, no conversion, and the same image serves the forward (K-contiguous weights) and the
//     data gradient (row-contiguous weights: the builder does the transposition), so the kernel has no B_RC variant.
//   * A (activations / gradients, fp32 in memory as before) is gathered in FRAGMENT layout too: lane (l15, lq) loads the eight consecutive
//     channels 8 lq .. 8 lq + 7 of pixel row l15 (two 16-byte buffer loads per 16-row half) and splits them in registers
//     (afi_pack8_bf16 / afi_bf16_residual).  A wave's split work is amortised over its 32 columns only -- 16 elements per lane and stage
//     against 24 MFMAs -- which is why B must not be split here as well (vector issue, not the matrix pipe, would then set the pace).
//   * No LDS in the K loop at all; the 36 KB of the block hold only the eight partial tiles of the reduction.
// One register set: the next stage's A gather is issued as soon as the split has consumed the current one, the next B fragments behind the
// MFMAs that read the current ones; four waves per SIMD (two blocks per CU) cover what remains of the latency.
// ------------------------------------------------------------------------------------------------
#ifndef AFI_WK6_ABLATE
#define AFI_WK6_ABLATE 0                                   // tools/micro/wk6_bench.cpp only: 1 no MFMAs, 2 no A gather, 4 no B loads, 8 no split (wrong results)
#endif
// NJ = 2: a wave's tile is 32 x 64 -- the same A fragments (gather, LDS staging, split: 110 of the ~160 vector-issue slots of a 32 x 32
// stage) feed 48 MFMAs instead of 24; 150 registers, so one block per CU: for the grids that still cover the chip with 64-column tiles.
template <bool DIAG, int NW = 8, int NJ = 1>
__device__ __forceinline__ void afi_wk6_body(const AfiPixGemm& p, const AfiWkArgs& sk, const int lb) {
    constexpr int BK = AFI_BK, LDK = BK + 4, TN = 32 * NJ, LDC = TN + 4, PATCH = 32 * LDC, NH = 2 * NJ;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if constexpr (DIAG) { if (threadIdx.x == 0) sk.dbg[(long long)blockIdx.x * 10 + 6] = __builtin_amdgcn_s_memrealtime(); }
    AFI_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int M = sk.M, HW = sk.HW;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A.p, 0, 0x7FFFFFF0, 0x00020000);
    const int tile_n = (int)afi_udiv((unsigned)lb, (unsigned)sk.ntile_m, sk.rcp_ntm), tile_m = lb - tile_n * sk.ntile_m;
    const int m0 = tile_m * 32, n0 = tile_n * TN;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int kc0 = wave_u * sk.nK / NW, kc1 = (wave_u + 1) * sk.nK / NW;   // this wave's K stages
    const int nK = kc1 - kc0;
    // this tile's part of the weight image: stage s of the problem at + s * 6144 bytes; a lane's fragment (n half ni, part) at
    // + (3 ni + part) * 1024 + 16 lane.  Addressed as a buffer (32-bit offsets; the launcher checks the image stays below 2 GB)
    const unsigned char* img = p.Bimg + ((long long)tile_n * NJ * p.bimg_nstages + p.bimg_stage0) * AFI_WK6_STAGE_BYTES;
    // (NJ = 2: the tile's second 32 columns are the image's next N tile, bimg_nstages stages further; an odd tile count: the last block's
    //  second half re-reads the first -- its columns lie beyond Ncols and are never stored)
    const int sub1 = (NJ == 2 && (tile_n * 2 + 1) * 32 < p.Ncols) ? p.bimg_nstages * AFI_WK6_STAGE_BYTES : 0;
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, 0x7FFFFFF0, 0x00020000);
    const int last_stage = sk.nK - 1;

    // ---- A gather, COALESCED: lane (r8 = lane >> 3, q8 = lane & 7) loads the 16 bytes q8 of the 128-byte channel chunk of pixel rows
    //      m0 + r8 + 8 i (i = 0..3): eight adjacent lanes read one whole cache line.  (Gathering straight into the MFMA's fragment layout
    //      -- lane (l15, lq) = row l15, channels 8 lq .. + 7 -- makes every group of adjacent lanes touch 16 different lines: measured with
    //      tools/micro/wk6_bench.cpp, that gather alone cost 12.7 of the kernel's 27.9 us at 850 x 384 x 2304, against 5 us for the split
    //      and the MFMAs.)  The fragments are formed through the wave's own LDS patch, as in the fp32 kernel above: no block barrier.
    const int q8 = lane & 7, r8 = lane >> 3;
    unsigned a_off[4], a_mask[4];
#pragma unroll
    for (int i2 = 0; i2 < 4; ++i2) {
        const int m = m0 + r8 + 8 * i2;
        const unsigned im = afi_udiv((unsigned)m, (unsigned)HW, sk.rcp_HW);
        const int rem = m - (int)im * HW;
        const int y = (int)afi_udiv((unsigned)rem, (unsigned)p.W, sk.rcp_W), x = rem - y * p.W;
        unsigned mk = 1u;
        if (p.ntaps == 9) {
            const unsigned cm = ((unsigned)(x - p.a_sgn) < (unsigned)p.W ? 1u : 0u) | 2u | ((unsigned)(x + p.a_sgn) < (unsigned)p.W ? 4u : 0u);
            mk = ((unsigned)(y - p.a_sgn) < (unsigned)p.H ? cm : 0u) | (cm << 3) | ((unsigned)(y + p.a_sgn) < (unsigned)p.H ? (cm << 6) : 0u);
        }
        a_mask[i2] = m < M ? mk : 0u;
        a_off[i2] = 4u * (unsigned)((m < M ? (int)im : 0) * (int)p.A.sN + (y * p.a_up) * (int)p.A.sH + (x * p.a_up) * (int)p.A.sW + 4 * q8);
    }
    // STAGE ROTATION: wave w of M tile m walks its K stages starting at stage (m mod nK) and wraps, so the blocks of one N tile (27 M
    // tiles at config 1, most of them on one XCD) do not ask for the same weight-image stage at the same time.  Same products, another
    // (fixed) order of the fp32 sums per M tile.  (Measured neutral at config 1 -- the kernel is not bound by first touches of the image --
    //  and kept: it spreads the requests of an XCD over the L2's channels.)
    const int rot = nK > 1 ? (int)((unsigned)tile_m % (unsigned)nK) : 0;
    const int kq0 = (int)afi_udiv((unsigned)kc0, (unsigned)p.ntaps, sk.rcp_taps);
    const int w_tap = kc0 - kq0 * p.ntaps, w_kph = kq0 & (p.nKphase - 1), w_c0 = (kq0 >> sk.kph_shift) * BK;   // the wave's first stage (wrap target)
    const int kq = (int)afi_udiv((unsigned)(kc0 + rot), (unsigned)p.ntaps, sk.rcp_taps);
    int k_tap = kc0 + rot - kq * p.ntaps, k_kph = kq & (p.nKphase - 1), k_c0 = (kq >> sk.kph_shift) * BK;   // NEXT stage to gather
    int k_abs = kc0 + rot;                                 // ... and its index in the problem's stage order
    auto stage_advance = [&]() {
        if (++k_tap == p.ntaps) {
            k_tap = 0;
            if (++k_kph == p.nKphase) { k_kph = 0; k_c0 += BK; }
        }
        if (++k_abs == kc1) { k_abs = kc0; k_tap = w_tap; k_kph = w_kph; k_c0 = w_c0; }
    };
    float* As = smem + wave_u * PATCH;                     // this wave's patch: [32 rows][LDK] fp32 (later its partial output tile)
    u32x4 a_raw[4], b_raw[NH][3];
    // ONE register set per operand.  A: stage s + 2 is in flight in a_raw while stage s + 1 sits in the LDS patch and stage s is multiplied;
    // B: the weight fragments of an n half are re-requested (next stage) right behind the twelve MFMAs that read them.
    // (Loads are issued unconditionally -- counted vmcnt waits need a path-independent count: a masked A lane reads out of range = zeros,
    //  a B load past the problem's last stage re-reads that stage and is never used.)
    auto load_a = [&](bool more) {
        int dy = 0, dx = 0;
        if (p.ntaps == 9) { dy = k_tap / 3 - 1; dx = k_tap - (k_tap / 3) * 3 - 1; }
        const int a_delta = 4 * ((dy * p.a_sgn * p.a_up + (k_kph >> 1)) * (int)p.A.sH + (dx * p.a_sgn * p.a_up + (k_kph & 1)) * (int)p.A.sW + k_c0);
        unsigned mm = more ? 1u : 0u;
        asm volatile("" : "+v"(mm));
        const unsigned okc = (k_c0 + 4 * q8) < p.Ck ? mm : 0u;                        // (only the last channel chunk can be partial)
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
            const unsigned ok = (a_mask[i2] >> k_tap) & okc;
            if constexpr ((AFI_WK6_ABLATE & 2) == 0)
                a_raw[i2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ok ? a_off[i2] + (unsigned)a_delta : 0xFFFFFFFFu, 0, 0);
            else { const unsigned o = ok ? a_off[i2] + (unsigned)a_delta : 0xFFFFFFFFu; a_raw[i2] = u32x4{o, o, o, o}; }
        }
        stage_advance();
    };
    auto store_a = [&]() {
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) *(u32x4*)(As + (r8 + 8 * i2) * LDK + 4 * q8) = a_raw[i2];
    };
    u32x4 fa[2][2];                                        // fragment layout: row 16 mi + l15, channels 8 lq + 4 h .. + 3
    auto read_frags = [&]() {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int h = 0; h < 2; ++h) fa[mi][h] = *(const u32x4*)(As + (16 * mi + l15) * LDK + 8 * lq + 4 * h);
    };
    auto lds_order = [&]() {                               // (lgkmcnt only) a wave's LDS instructions execute in order; this pins the compiler's order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    };
    auto load_b = [&](int ni, int stage_abs) {             // (ni: compile-time at every call site)
        const int st = stage_abs < last_stage ? stage_abs : last_stage;
        const int soff = __builtin_amdgcn_readfirstlane(st) * AFI_WK6_STAGE_BYTES + ((ni >> 1) ? sub1 : 0);
#pragma unroll
        for (int pt = 0; pt < 3; ++pt) {
            if constexpr ((AFI_WK6_ABLATE & 4) == 0)
                b_raw[ni][pt] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (unsigned)(16 * lane + (3 * (ni & 1) + pt) * 1024), soff, 0);
            else { const unsigned o = (unsigned)soff + lane; b_raw[ni][pt] = u32x4{o, o, o, o}; }
        }
    };
    f32x4 acc[2][NH];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NH; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mfma = [](u32x4 x, u32x4 y, f32x4 c) -> f32x4 {
        if constexpr ((AFI_WK6_ABLATE & 1) != 0) { asm volatile("" :: "v"(x), "v"(y)); return c; }
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0);
    };

    AFI_STAMP(1);
    const int kb_first = k_abs;
    load_a(0 < nK);                                        // stage 0 (k_abs moves on to the second stage)
    AFI_STAMP(2);
    store_a();
    lds_order();
    read_frags();
    int kb_next = k_abs;                                   // the stage a_raw is about to receive: its weight fragments follow the current MFMAs
    load_a(1 < nK);                                        // stage 1
    // (the first weight fragments are requested BEHIND the second gather, the order every later stage has: the loop's counted waits are the
    //  merge of the entry and the back-edge states, and with the fragments requested first every stage waited for them before its stores)
#pragma unroll
    for (int ni = 0; ni < NH; ++ni) load_b(ni, kb_first);
    AFI_STAMP(3);
    for (int kc = 0; kc < nK; ++kc) {
        u32x4 ah[2], am[2], al[2];                          // packed bf16 pairs: element 2 j in the low half of word j
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 c = __builtin_bit_cast(f32x4, fa[mi][h]);
                unsigned h0, m0_, l0, h1, m1_, l1;
                if constexpr ((AFI_WK6_ABLATE & 8) == 0) {
                    afi_split3_pair(c[0], c[1], h0, m0_, l0);
                    afi_split3_pair(c[2], c[3], h1, m1_, l1);
                } else { h0 = fa[mi][h][0]; h1 = fa[mi][h][1]; m0_ = fa[mi][h][2]; m1_ = fa[mi][h][3]; l0 = h0 ^ m0_; l1 = h1 ^ m1_; }
                ah[mi][2 * h] = h0; ah[mi][2 * h + 1] = h1; am[mi][2 * h] = m0_; am[mi][2 * h + 1] = m1_; al[mi][2 * h] = l0; al[mi][2 * h + 1] = l1;
            }
        // the fragments of stage kc are split: the patch is free for stage kc + 1 (gathered one stage ago), a_raw for stage kc + 2
        store_a();
        const int kb_this = kb_next;
        kb_next = k_abs;
        load_a(kc + 2 < nK);
        lds_order();
        read_frags();                                      // stage kc + 1, in flight under the MFMAs below
        // per accumulator: smallest terms first; consecutive MFMAs alternate between the two accumulators of the n half
#pragma unroll
        for (int ni = 0; ni < NH; ++ni) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = mfma(al[mi], b_raw[ni][0], acc[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = mfma(ah[mi], b_raw[ni][2], acc[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = mfma(am[mi], b_raw[ni][1], acc[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = mfma(am[mi], b_raw[ni][0], acc[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = mfma(ah[mi], b_raw[ni][1], acc[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = mfma(ah[mi], b_raw[ni][0], acc[mi][ni]);
            __builtin_amdgcn_sched_barrier(0);
            load_b(ni, kb_this);                           // this half's fragments of the next stage
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    lds_order();                                           // (the patch becomes this wave's partial output tile below)
    AFI_STAMP(4);
    // ---- the eight partial tiles meet in LDS (each wave writes its own patch), then 256 threads sum them and run the fused epilogue;
    //      the epilogue's own operands are requested first (their global round trip runs under the LDS write, the barrier and the reads)
    constexpr int C4 = TN / 4;                              // float4 columns of the tile
    AfiEpiPre epre[NJ]; int e_img[NJ], e_y[NJ], e_x[NJ]; bool e_on[NJ];
#pragma unroll
    for (int it = 0; it < NJ; ++it) {
        const int item = tid + 256 * it, e_rl = item / C4, e_c4 = item - e_rl * C4;
        const int e_m = m0 + e_rl, e_col = n0 + 4 * e_c4;
        e_on[it] = tid < 256 && e_m < M && e_col < p.Ncols;
        e_img[it] = e_y[it] = e_x[it] = 0;
        if (e_on[it]) {
            afi_gfloat* zpage = (afi_gfloat*)afi_zeros;
            asm volatile("" : "+v"(zpage));
            const unsigned im = afi_udiv((unsigned)e_m, (unsigned)HW, sk.rcp_HW);
            const int rem = e_m - (int)im * HW;
            e_y[it] = (int)afi_udiv((unsigned)rem, (unsigned)p.W, sk.rcp_W); e_x[it] = rem - e_y[it] * p.W; e_img[it] = (int)im;
            epre[it] = afi_epilogue_prefetch(p, e_img[it], e_y[it], e_x[it], e_col, zpage);
        }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NH; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) As[(16 * mi + 4 * lq + r) * LDC + 16 * ni + l15] = acc[mi][ni][r];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NJ; ++it) {
        if (!e_on[it]) continue;
        const int item = tid + 256 * it, e_rl = item / C4, e_c4 = item - e_rl * C4;
        f32x4 v = *(const f32x4*)(smem + e_rl * LDC + 4 * e_c4);
#pragma unroll
        for (int w = 1; w < NW; ++w) v += *(const f32x4*)(smem + w * PATCH + e_rl * LDC + 4 * e_c4);     // fixed order: bit-reproducible
        afi_epilogue_finish(p, epre[it], e_img[it], e_y[it], e_x[it], v);
    }
    AFI_STAMP(5);
    AFI_STAMP(8);
    if constexpr (DIAG) { if (threadIdx.x == 0) sk.dbg[(long long)blockIdx.x * 10 + 7] = __builtin_amdgcn_s_memrealtime(); }
}
template <bool DIAG = false>
__global__ __launch_bounds__(512, 4) void afi_pix_gemm_wk6_kernel(const AfiPixGemm p, const AfiWkArgs sk) {
    afi_wk6_body<DIAG>(p, sk, afi_xcd_logical_id());
}
__global__ __launch_bounds__(512, 4) void afi_pix_gemm_wk6_group_kernel(const AfiWkGroup grp) {
    const int t = afi_xcd_logical_id();
    int pi = 0;
    while (pi + 1 < grp.nprob && t >= grp.tile_start[pi + 1]) ++pi;     // (uniform)
    afi_wk6_body<false>(grp.p[pi], grp.wk[pi], t - grp.tile_start[pi]);
}
// the 32 x 64 form (one block per CU): single problems, and groups whose wide problems take it (wk.nj == 2) beside 32-column ones
__global__ __launch_bounds__(512, 2) void afi_pix_gemm_wk6w_kernel(const AfiPixGemm p, const AfiWkArgs sk) {
    afi_wk6_body<false, 8, 2>(p, sk, afi_xcd_logical_id());
}
__global__ __launch_bounds__(512, 2) void afi_pix_gemm_wk6w_group_kernel(const AfiWkGroup grp) {
    const int t = afi_xcd_logical_id();
    int pi = 0;
    while (pi + 1 < grp.nprob && t >= grp.tile_start[pi + 1]) ++pi;     // (uniform)
    if (grp.wk[pi].nj == 2) afi_wk6_body<false, 8, 2>(grp.p[pi], grp.wk[pi], t - grp.tile_start[pi]);
    else afi_wk6_body<false, 8, 1>(grp.p[pi], grp.wk[pi], t - grp.tile_start[pi]);
}

// Weight images of afi_pix_gemm_wk6.  One thread per (job, N tile, K stage, n half, lane): the lane's eight consecutive k of column
// n = 32 tile + 16 half + (lane & 15) -- k = 8 (lane >> 4) .. + 7 of the stage's 32-channel chunk at its (K phase, tap) -- read through the
// job's B addressing (zeros beyond Ncols / Ck), split, and stored as the three 16-byte fragments the kernel loads.
#define AFI_WK6_MAXJOBS 40
struct AfiWk6ImgJobs {
    int njobs, pad_;
    int unit_start[AFI_WK6_MAXJOBS + 1];                   // prefix sums of (N tiles x stages) per job
    int nstages[AFI_WK6_MAXJOBS];
    AfiWk6ImgJob j[AFI_WK6_MAXJOBS];
    // blocks of image units, then of the conv-transpose image, of its packed fp32 form, of the zero fill (8 float4 per thread), of the skip gradient (one float4 per thread)
    int nb_img, nb_ct, nb_ctpack, nb_zero;
    AfiWk6ConvT ct;
    AfiWk6Side side;
};
#define AFI_WK6_CT_LDS (32 * (4 * 36 + 1))                  /* floats: 32 x 145 (mode 0) >= 16 x 289 (mode 1) >= the pack's 32 x 73 */
template <int MODE>
__device__ __forceinline__ void afi_wk6_convT_image_body(const AfiWk6ConvT& ct, const int b, float* T) {
    constexpr int CI = MODE ? 16 : 32, CO = MODE ? 8 : 4, LD = CO * 36 + 1, RQ = CO * 9;   // tile, LDS row, float4 per ci row
    const int nbx = ct.Cin / CI, bx = b % nbx, by = b / nbx;
    const int ci0 = CI * bx, co0 = CO * by, tid = threadIdx.x;
    for (int i = tid; i < CI * RQ; i += 256) {              // row r: the CO * 36 contiguous floats of (ci0 + r, co0 .. co0 + CO)
        const int r = i / RQ, q = i - r * RQ;
        const f32x4 v = *(const f32x4*)(ct.W + ((long long)(ci0 + r) * ct.Cout + co0) * 36 + 4 * q);
        float* d = T + r * LD + 4 * q;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    __syncthreads();
    const int grp = tid >> 4, l = tid & 15;                 // sixteen lanes per (phase, tap)
    for (int q = grp; q < 36; q += 16) {
        const int phase = q / 9, tap = q - 9 * phase;
        const int kk = ((phase >> 1) + 2 - 2 * (tap / 3 - 1)) * 6 + ((phase & 1) + 2 - 2 * (tap % 3 - 1));   // ky * 6 + kx
        f32x4 v0, v1;
        unsigned char* dst;
        if constexpr (MODE == 0) {                          // column (phase, co0 + co_l), k = ci: the lane's eight consecutive ci
            const int co_l = l & 3, lq = l >> 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) { v0[j] = T[(8 * lq + j) * LD + co_l * 36 + kk]; v1[j] = T[(8 * lq + 4 + j) * LD + co_l * 36 + kk]; }
            const int n = phase * ct.Cout + co0 + co_l;
            const int nst = (ct.Cin >> 5) * 9, stage = bx * 9 + tap;
            dst = ct.dst + ((long long)(n >> 5) * nst + stage) * AFI_WK6_STAGE_BYTES + (3 * ((n >> 4) & 1)) * 1024 + 16 * (16 * lq + (n & 15));
        } else {                                            // column ci0 + l, k = co: this block's eight co are one lane group of their chunk
#pragma unroll
            for (int j = 0; j < 4; ++j) { v0[j] = T[l * LD + j * 36 + kk]; v1[j] = T[l * LD + (4 + j) * 36 + kk]; }
            const int n = ci0 + l;
            const int nst = (ct.Cout >> 5) * 36, stage = ((co0 >> 5) * 4 + phase) * 9 + tap;
            dst = ct.dst + ((long long)(n >> 5) * nst + stage) * AFI_WK6_STAGE_BYTES + (3 * ((n >> 4) & 1)) * 1024 + 16 * (16 * ((co0 >> 3) & 3) + (n & 15));
        }
        const f32x4 r0 = afi_bf16_residual(v0), r1 = afi_bf16_residual(v1);
        *(bf16x8*)dst = afi_pack8_bf16(v0, v1);
        *(bf16x8*)(dst + 1024) = afi_pack8_bf16(r0, r1);
        *(bf16x8*)(dst + 2048) = afi_pack8_bf16(afi_bf16_residual(r0), afi_bf16_residual(r1));
    }
}
__global__ __launch_bounds__(256) void afi_wk6_image_kernel(const AfiWk6ImgJobs jobs) {
    if ((int)blockIdx.x >= jobs.nb_img) {                   // (uniform) what rides with the images: AfiWk6ConvT, AfiWk6Side
        __shared__ __attribute__((aligned(16))) float T[AFI_WK6_CT_LDS];
        int zb = (int)blockIdx.x - jobs.nb_img;
        if (zb < jobs.nb_ct) {
            if (jobs.ct.mode == 0) afi_wk6_convT_image_body<0>(jobs.ct, zb, T);
            else afi_wk6_convT_image_body<1>(jobs.ct, zb, T);
            return;
        }
        zb -= jobs.nb_ct;
        if (zb < jobs.nb_ctpack) {
            const int nbx = (jobs.ct.Cin + AFI_CT_CI - 1) / AFI_CT_CI;
            afi_convT_repack_body<false>(jobs.ct.W, jobs.ct.pack_dst, jobs.ct.Cin, jobs.ct.Cout, zb % nbx, zb / nbx, (float (*)[AFI_CT_LD])T);
            return;
        }
        zb -= jobs.nb_ctpack;
        if (zb < jobs.nb_zero) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long long i = (long long)zb * 2048 + j * 256 + threadIdx.x;
                if (i < jobs.side.zero_n4) ((f32x4*)jobs.side.zero_p)[i] = z;
            }
        } else {
            const long long i = (long long)(zb - jobs.nb_zero) * 256 + threadIdx.x;
            if (i < (long long)jobs.side.bl_N * jobs.side.bl_H * jobs.side.bl_W * (jobs.side.bl_C / 4))
                *(f32x4*)(jobs.side.bl_dx + i * 4) = afi_bilinear2x_bwd_elem(jobs.side.bl_dout, jobs.side.bl_H, jobs.side.bl_W, jobs.side.bl_C, i);
        }
        return;
    }
    const int u = (int)blockIdx.x * 2 + (int)(threadIdx.x >> 7);
    if (u >= jobs.unit_start[jobs.njobs]) return;
    int ji = 0;
    while (ji + 1 < jobs.njobs && u >= jobs.unit_start[ji + 1]) ++ji;
    const AfiWk6ImgJob& jb = jobs.j[ji];
    const int nst = jobs.nstages[ji];
    const int local = u - jobs.unit_start[ji];
    const int tile_n = local / nst, stage = local - tile_n * nst;
    const int tap = stage % jb.ntaps, kq = stage / jb.ntaps;
    const int kph = kq % jb.nKphase, chunk = kq / jb.nKphase;
    const int t = threadIdx.x & 127, ni = t >> 6, lane = t & 63, l15 = lane & 15, lq = lane >> 4;
    const int n = tile_n * 32 + 16 * ni + l15, c = chunk * 32 + 8 * lq;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
    if (n < jb.Ncols) {
        if (!jb.b_rc) {
            const float* src = jb.B + (long long)n * jb.b_sRow + (long long)tap * jb.b_sTap + c;
            if (c + 4 <= jb.Ck) v0 = *(const f32x4*)src;
            if (c + 8 <= jb.Ck) v1 = *(const f32x4*)(src + 4);
        } else {
            const float* src = jb.B + ((long long)kph * jb.Ck + c) * jb.b_sRow + (long long)tap * jb.b_sTap + n;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (c + j < jb.Ck) v0[j] = src[(long long)j * jb.b_sRow];
                if (c + 4 + j < jb.Ck) v1[j] = src[(long long)(4 + j) * jb.b_sRow];
            }
        }
    }
    const int nst_img = jb.nstages_img > 0 ? jb.nstages_img : nst;
    unsigned char* dst = jb.dst + ((long long)tile_n * nst_img + jb.stage_off + stage) * AFI_WK6_STAGE_BYTES + (3 * ni) * 1024 + 16 * lane;
    const f32x4 r0 = afi_bf16_residual(v0), r1 = afi_bf16_residual(v1);
    *(bf16x8*)dst = afi_pack8_bf16(v0, v1);
    *(bf16x8*)(dst + 1024) = afi_pack8_bf16(r0, r1);
    *(bf16x8*)(dst + 2048) = afi_pack8_bf16(afi_bf16_residual(r0), afi_bf16_residual(r1));
}
long long afi_wk6_image_bytes(int Ncols, int Ck, int ntaps, int nKphase) {
    return (long long)afi_cdiv(Ncols, 32) * afi_cdiv(Ck, AFI_BK) * ntaps * nKphase * AFI_WK6_STAGE_BYTES;
}
int afi_launch_wk6_images(const AfiWk6ImgJob* jobs, int n, hipStream_t st, const AfiWk6Side* side = nullptr, const AfiWk6ConvT* ct = nullptr) {
    static_assert(AFI_WK6_CT_LDS >= 16 * (8 * 36 + 1) && AFI_WK6_CT_LDS >= AFI_CT_CI * AFI_CT_LD, "one LDS tile for the three conv-transpose bodies");
    if (ct) {
        if (!ct->W || !ct->dst || ct->Cin <= 0 || ct->Cout <= 0 || (ct->mode != 0 && ct->mode != 1)) return AFI_ERR_BAD_ARG;
        if ((ct->Cin & 31) || (ct->Cout & 31) || (((uintptr_t)ct->W | (uintptr_t)ct->dst) & 15)) return AFI_ERR_UNSUPPORTED;
    }
    if (side) {
        if ((side->zero_p && ((((uintptr_t)side->zero_p) & 15) || side->zero_n4 < 0)) ||
            (side->bl_dx && (!side->bl_dout || side->bl_N <= 0 || side->bl_H <= 0 || side->bl_W <= 0 || side->bl_C <= 0 || (side->bl_C & 3)))) return AFI_ERR_BAD_ARG;
    }
    for (int done = 0; done < n || side || ct;) {
        const int cnt = (n - done) < AFI_WK6_MAXJOBS ? (n - done) : AFI_WK6_MAXJOBS;
        AfiWk6ImgJobs tb;
        tb.njobs = cnt; tb.pad_ = 0;
        long long units = 0;
        for (int i = 0; i < cnt; ++i) {
            const AfiWk6ImgJob& j = jobs[done + i];
            if (!j.B || !j.dst || j.Ncols <= 0 || j.Ck <= 0 || (j.ntaps != 1 && j.ntaps != 9) || (j.nKphase != 1 && j.nKphase != 4)) return AFI_ERR_BAD_ARG;
            if (j.stage_off < 0 || (j.nstages_img > 0 && j.stage_off + afi_cdiv(j.Ck, AFI_BK) * j.ntaps * j.nKphase > j.nstages_img)) return AFI_ERR_BAD_ARG;
            if ((j.Ck & 3) || (!j.b_rc && ((j.b_sRow | j.b_sTap) & 3)) || (((uintptr_t)j.B | (uintptr_t)j.dst) & 15)) return AFI_ERR_UNSUPPORTED;   // float4 reads, 16-byte stores
            tb.j[i] = j;
            tb.nstages[i] = afi_cdiv(j.Ck, AFI_BK) * j.ntaps * j.nKphase;
            tb.unit_start[i] = (int)units;
            units += (long long)afi_cdiv(j.Ncols, 32) * tb.nstages[i];
            if (units > 0x3fffffffLL) return AFI_ERR_UNSUPPORTED;
        }
        for (int i = cnt; i <= AFI_WK6_MAXJOBS; ++i) tb.unit_start[i] = (int)units;
        tb.nb_img = (int)((units + 1) / 2); tb.nb_zero = tb.nb_ct = tb.nb_ctpack = 0;
        tb.side = AfiWk6Side{nullptr, 0, nullptr, nullptr, 0, 0, 0, 0};
        tb.ct = AfiWk6ConvT{nullptr, nullptr, nullptr, 0, 0, 0, 0};
        long long blocks = tb.nb_img;
        if (ct) {                                           // (with the first batch of jobs, or alone)
            tb.ct = *ct;
            tb.nb_ct = ct->mode ? (ct->Cin / 16) * (ct->Cout / 8) : (ct->Cin / 32) * (ct->Cout / 4);
            if (ct->pack_dst) tb.nb_ctpack = ((ct->Cin + AFI_CT_CI - 1) / AFI_CT_CI) * ((ct->Cout + AFI_CT_CO - 1) / AFI_CT_CO);
            blocks += tb.nb_ct + tb.nb_ctpack;
            ct = nullptr;
        }
        if (side) {
            tb.side = *side;
            if (side->zero_p) { tb.nb_zero = afi_cdiv(side->zero_n4, 2048); blocks += tb.nb_zero; }
            if (side->bl_dx) blocks += afi_cdiv((long long)side->bl_N * side->bl_H * side->bl_W * (side->bl_C / 4), 256);
            side = nullptr;
        }
        if (blocks > 0x3fffffffLL) return AFI_ERR_UNSUPPORTED;
        if (blocks > 0) hipLaunchKernelGGL(afi_wk6_image_kernel, dim3((unsigned)blocks), dim3(256), 0, st, tb);
        done += cnt;
    }
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// fills the kernel's argument block; AFI_ERR_UNSUPPORTED when the problem does not fit the kernel's 32-bit index math
static int wk_prepare(const AfiPixGemm& p, bool b_rc, AfiWkArgs& wk) {
    const long long M = (long long)p.N * p.H * p.W;
    if (p.gtap || p.b_sImg != 0 || (p.ntaps != 1 && p.ntaps != 9)) return AFI_ERR_UNSUPPORTED;
    if (p.nKphase != 1 && p.nKphase != 4) return AFI_ERR_UNSUPPORTED;
    if (M * (long long)p.H * p.W >= (1LL << 32) || M >= (1LL << 30)) return AFI_ERR_UNSUPPORTED;   // 32-bit index math, reciprocal division
    {   // buffer loads address both operands with 32-bit byte offsets below 2 GB
        auto ab = [](long long v) { return v < 0 ? -v : v; };
        const long long a_ext = (ab(p.A.sN) * p.N + ab(p.A.sH) * ((long long)p.H * p.a_up + 2) + ab(p.A.sW) * ((long long)p.W * p.a_up + 2) + p.Ck + 64) * 4;
        const long long b_rows = b_rc ? (long long)p.nKphase * p.Ck : p.Ncols;
        const long long b_ext = (ab(p.b_sRow) * (b_rows + 32) + ab(p.b_sTap) * p.ntaps + (b_rc ? p.Ncols : p.Ck) + 64) * 4;
        if (p.A.sN < 0 || p.A.sH < 0 || p.A.sW < 0 || p.b_sRow < 0 || p.b_sTap < 0 || a_ext >= 0x7FFFFFF0LL || b_ext >= 0x7FFFFFF0LL) return AFI_ERR_UNSUPPORTED;
    }
    wk.ntile_m = afi_cdiv(M, 32); wk.ntile_n = afi_cdiv(p.Ncols, 32);
    wk.nK = p.ntaps * p.nKphase * afi_cdiv(p.Ck, AFI_BK);
    wk.M = (int)M; wk.HW = p.H * p.W;
    wk.rcp_HW = sk_rcp((unsigned)wk.HW); wk.rcp_W = sk_rcp((unsigned)p.W); wk.rcp_taps = sk_rcp((unsigned)p.ntaps); wk.rcp_ntm = sk_rcp((unsigned)wk.ntile_m);
    wk.kph_shift = p.nKphase == 4 ? 2 : 0;
    wk.nj = 1;
    wk.dbg = nullptr;
    if ((long long)wk.ntile_m * wk.ntile_n * wk.ntile_m >= (1LL << 32)) return AFI_ERR_UNSUPPORTED;
    return AFI_OK;
}
// the bf16x6 kernel's 32 x 64 form for this problem: half as many (twice as wide) tiles
static inline void wk6_widen(const AfiPixGemm& p, AfiWkArgs& wk) { wk.nj = 2; wk.ntile_n = afi_cdiv(p.Ncols, 64); }
static inline bool wk6_wide_candidate(const AfiPixGemm& p, const AfiWkArgs& wk) { return p.Ncols >= 64 && wk.nK >= 16; }
// ------------------------------------------------------------------------------------------------
// afi_rdb_chain6_kernel: the dense block's chain of 32-channel convs (AfiChain6, afi_common.h) in one launch.  At config 1 each link was
// a launch of its own whose 1 us of matrix work sat in 9 us of fixed cost (launch, first touches, reduction, epilogue), four per block
// and direction; here a link is a phase of a resident block: its inputs are in LDS, its weights come from the same bf16x6 images the
// small-map GEMMs read, and what a neighbour tile would have provided is recomputed on the halo (144 + 100 + 64 pixels per 64 owned, on
// 32-column GEMMs: noise next to the launches saved).
//   Sixteen waves.  A phase's work items are (PAIR of 16-pixel row groups of its region, K slice): 5 x 2, 4 x 3 and 2 x 5 items for the
//   three phases, i.e. 5 / 6 / 6 stages in a row per wave instead of 9 / 18 / 27 -- the kernel is a latency chain, so its length counts --
//   and a stage's six 1-KB weight fragments (from L2, prefetched one stage ahead) feed 24 MFMAs on two row groups.
//   An item leaves its raw partial tiles in an LDS slot of its K slice; behind a barrier all threads sum the slots in FIXED order
//   (bit-reproducible), add `partial`, apply the LeakyReLU / LeakyReLU' factor, zero what lies outside the map, and store float4 rows to
//   the next region and -- the owner's 8 x 8 pixels -- to `out`.
// ------------------------------------------------------------------------------------------------
// LDS: the three regions hold their 32 channels PRE-SPLIT -- per pixel 240 bytes: hi | mid | lo, each 32 bf16 (64 bytes) + 16 bytes of pad,
// so that a fragment (lane = pixel l15, channels 8 lq .. + 7) is ONE 16-byte read per part and the 16 lanes of a read group fall on
// distinct banks (pixel stride 60 banks) -- because a region element is read by nine taps of up to three links: split at every read
// (v3 of this kernel), the 36 vector instructions per fragment against 12 MFMAs on a 32-column GEMM made the block's 171 wave-stages
// a vector-issue queue on ONE CU: 16 of the launch's 23 us (tools/micro/ch6_bench.cpp: the same time with the weight loads or the MFMAs
// removed).  Split once where the element is produced, a stage is six weight loads, six LDS reads and 24 MFMAs.
#define AFI_CH6_PIX 240                                    // bytes per region pixel
#define AFI_CH6_R1 (196 * AFI_CH6_PIX)
#define AFI_CH6_R2 ((196 + 144) * AFI_CH6_PIX)
#define AFI_CH6_SLOTS ((196 + 144 + 100) * AFI_CH6_PIX)    // K-slice partial tiles (fp32, 36 floats per row): 2 slots of 144 rows / 3 of 112 / 5 of 64
#define AFI_CH6_LDP 36
#define AFI_CH6_SLOT_ROWS 336
#define AFI_CH6_LDS_BYTES (AFI_CH6_SLOTS + AFI_CH6_SLOT_ROWS * AFI_CH6_LDP * 4)
#ifndef AFI_CH6_ABLATE
#define AFI_CH6_ABLATE 0                                   // tools/micro/ch6_bench.cpp only: 1 no stages (loads + MFMAs), 2 no weight-fragment loads, 4 no MFMAs
#endif
__device__ __forceinline__ void afi_ch6_store_split(unsigned char* pix_base, int c4, f32x4 v) {      // channels c4 .. c4 + 3 of one pixel -> its three parts
    unsigned h0, m0_, l0, h1, m1_, l1;
    afi_split3_pair(v[0], v[1], h0, m0_, l0);
    afi_split3_pair(v[2], v[3], h1, m1_, l1);
    *(u32x2*)(pix_base + 2 * c4) = u32x2{h0, h1};
    *(u32x2*)(pix_base + 80 + 2 * c4) = u32x2{m0_, m1_};
    *(u32x2*)(pix_base + 160 + 2 * c4) = u32x2{l0, l1};
}
template <int P, int MODE>
__device__ __forceinline__ void afi_chain6_phase(const AfiChain6& c, unsigned char* lds, int n, int y0, int x0) {
    constexpr int LDP = AFI_CH6_LDP;
    constexpr int S = 12 - 2 * P, NPIX = S * S, NSUB = (NPIX + 15) / 16, HP = 2 - P, NST = 9 * (P + 1);
    constexpr int KS = P == 0 ? 2 : (P == 1 ? 3 : 5), NPAIR = (NSUB + 1) / 2, NITEM = NPAIR * KS, ROWS = NSUB * 16;
    static_assert(KS * ROWS <= AFI_CH6_SLOT_ROWS, "slot area");
    auto R = [&](int ci) -> unsigned char* { return lds + (ci == 0 ? 0 : (ci == 1 ? AFI_CH6_R1 : AFI_CH6_R2)); };   // regions 0 (14 x 14), 1 (12 x 12), 2 (10 x 10)
    float* slots = (float*)(lds + AFI_CH6_SLOTS);           // [KS][ROWS][LDP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    const AfiChain6Phase& ph = c.ph[P];
    typedef const __attribute__((address_space(1))) u32x4 gu32x4;
    auto mfma = [](u32x4 x, u32x4 y, f32x4 a) -> f32x4 {
        if constexpr ((AFI_CH6_ABLATE & 4) != 0) { asm volatile("" :: "v"(x), "v"(y)); return a; }
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), a, 0, 0, 0);
    };
    // the epilogue's own operands (`partial`, Z: first touches of lines the previous kernel wrote) are requested BEFORE the K loop
    constexpr int NEPI = (NPIX * 8 + 1023) / 1024;
    f32x4 e_part[NEPI], e_z[MODE == 1 ? NEPI : 1];
#pragma unroll
    for (int it = 0; it < NEPI; ++it) {
        const int e = tid + 1024 * it, q2 = e >> 3, c4 = (e & 7) * 4;
        const int py_ = q2 / S, px_ = q2 - py_ * S;
        const int gy = y0 - HP + py_, gx = x0 - HP + px_;
        const bool inside = e < NPIX * 8 && (unsigned)gy < (unsigned)c.H && (unsigned)gx < (unsigned)c.W;
        e_part[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (MODE == 1) e_z[it] = f32x4{1.f, 1.f, 1.f, 1.f};
        if (inside) {
            e_part[it] = *(const f32x4*)(ph.partial.p + (long long)n * ph.partial.sN + (long long)gy * ph.partial.sH + (long long)gx * ph.partial.sW + c4);
            if constexpr (MODE == 1) e_z[it] = *(const f32x4*)(ph.Z.p + (long long)n * ph.Z.sN + (long long)gy * ph.Z.sH + (long long)gx * ph.Z.sW + c4);
        }
    }
    for (int item = wave; item < NITEM; item += 16) {       // (uniform per wave)
        const int pr = item / KS, ks = item - pr * KS;
        const int s_lo = ks * NST / KS, s_hi = (ks + 1) * NST / KS;
        const bool two = 2 * pr + 1 < NSUB;                 // (uniform) an odd group count: the last pair's second group repeats the first, unused
        int qy[2], qx[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int u = two ? 2 * pr + g : 2 * pr;
            const int q = 16 * u + l15, qq = q < NPIX ? q : NPIX - 1;
            qy[g] = qq / S; qx[g] = qq - qy[g] * S;
        }
        f32x4 acc[2][2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[g][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 bw[2][6];                                     // weight fragments of two stages in flight: [n half][hi | mid | lo]
        auto load_b = [&](int set, int st) {                // stage st = 9 chunk + tap (clamped: a prefetch past the slice re-reads its last stage)
            st = st < s_hi ? st : s_hi - 1;
            const int ci = st / 9, tap = st - 9 * ci;
            const unsigned char* base = ph.img[ci] + (long long)(ph.stage0[ci] + tap) * AFI_WK6_STAGE_BYTES + 16 * lane;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                if constexpr ((AFI_CH6_ABLATE & 3) == 0) bw[set][k] = *(gu32x4*)(base + k * 1024);
                else { const unsigned o = (unsigned)(size_t)base + k; bw[set][k] = u32x4{o, o, o, o}; }
            }
        };
        auto stage = [&](int set, int st) {
            if constexpr ((AFI_CH6_ABLATE & 1) != 0) { acc[0][0][0] += __uint_as_float(bw[set][0][0]) * (float)st; return; }
            const int ci = st / 9, tap = st - 9 * ci;
            const int Sc = 14 - 2 * ci, off = 1 + P - ci;
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            u32x4 ah[2], am[2], al[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const unsigned char* a = R(ci) + ((qy[g] + off + c.a_sgn * dy) * Sc + (qx[g] + off + c.a_sgn * dx)) * AFI_CH6_PIX + 16 * lq;
                ah[g] = *(const u32x4*)a; am[g] = *(const u32x4*)(a + 80); al[g] = *(const u32x4*)(a + 160);
            }
            // per accumulator smallest terms first; consecutive MFMAs go to the four different accumulators (a dependent MFMA waits for its
            // predecessor's result: six in a row on one accumulator were most of a stage's time)
#define AFI_CH6_TERM(A, B)                                                                                  \
            _Pragma("unroll") for (int g = 0; g < 2; ++g)                                                   \
                _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) acc[g][ni] = mfma(A[g], bw[set][3 * ni + B], acc[g][ni]);
            AFI_CH6_TERM(al, 0)
            AFI_CH6_TERM(ah, 2)
            AFI_CH6_TERM(am, 1)
            AFI_CH6_TERM(am, 0)
            AFI_CH6_TERM(ah, 1)
            AFI_CH6_TERM(ah, 0)
#undef AFI_CH6_TERM
        };
        load_b(0, s_lo);
        int st = s_lo;
        for (; st + 2 <= s_hi; st += 2) {                   // pairs of stages, then (odd count) the last one
            load_b(1, st + 1);
            stage(0, st);
            load_b(0, st + 2);
            stage(1, st + 1);
        }
        if (st < s_hi) stage(0, st);
        // the item's raw partial tiles -> its K slice's slot: lane (l15, lq) holds channels l15, 16 + l15 of rows 4 lq .. 4 lq + 3 of a group
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (g == 1 && !two) break;
            float* sl = slots + ((long long)ks * ROWS + 16 * (2 * pr + g)) * LDP;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) sl[(4 * lq + r) * LDP + 16 * ni + l15] = acc[g][ni][r];
        }
    }
    __syncthreads();
    // ---- the link's epilogue, by all threads: one float4 (four channels of one pixel) per item
#pragma unroll
    for (int it = 0; it < NEPI; ++it) {
        const int e = tid + 1024 * it;
        if (e >= NPIX * 8) break;
        const int q2 = e >> 3, c4 = (e & 7) * 4;
        f32x4 v = *(const f32x4*)(slots + q2 * LDP + c4);
#pragma unroll
        for (int k = 1; k < KS; ++k) v += *(const f32x4*)(slots + ((long long)k * ROWS + q2) * LDP + c4);          // fixed order
        const int py_ = q2 / S, px_ = q2 - py_ * S;
        const int gy = y0 - HP + py_, gx = x0 - HP + px_;
        const bool inside = (unsigned)gy < (unsigned)c.H && (unsigned)gx < (unsigned)c.W;
        if (inside) {
            v += e_part[it];
            if constexpr (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : v[k] * AFI_LRELU_SLOPE;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] *= e_z[it][k] > 0.f ? 1.f : AFI_LRELU_SLOPE;
            }
        } else {
            v = f32x4{0.f, 0.f, 0.f, 0.f};                  // the convs' zero padding
        }
        if constexpr (P < 2) afi_ch6_store_split(R(P + 1) + q2 * AFI_CH6_PIX, c4, v);
        if (inside && (unsigned)(py_ - HP) < 8u && (unsigned)(px_ - HP) < 8u)
            *(f32x4*)(ph.out.p + (long long)n * ph.out.sN + (long long)gy * ph.out.sH + (long long)gx * ph.out.sW + c4) = v;
    }
}
template <int MODE>
__global__ __launch_bounds__(1024, 4) void afi_rdb_chain6_kernel(const AfiChain6 c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_c[];
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int tx = b % c.tiles_x; b /= c.tiles_x;
    const int ty = b % c.tiles_y; const int n = b / c.tiles_y;
    const int y0 = ty * 8, x0 = tx * 8;
    // (measured and dropped: touching the 2592 lines of weight fragments the block will read, up front, to pay one HBM round trip instead
    //  of one per stage: 26 against 24 us per launch -- the stages are not waiting for first touches)
    // region 0 (14 x 14 pixels x 32 channels): eight lanes read one pixel's 128 bytes; zeros outside the map; split on the way into LDS
    for (int idx = tid; idx < 196 * 8; idx += 1024) {
        const int pix = idx >> 3, seg = idx & 7;
        const int ry = pix / 14, rx = pix - ry * 14;
        const int gy = y0 - 3 + ry, gx = x0 - 3 + rx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const bool inside = (unsigned)gy < (unsigned)c.H && (unsigned)gx < (unsigned)c.W;
        if (inside) v = *(const f32x4*)(c.src0.p + (long long)n * c.src0.sN + (long long)gy * c.src0.sH + (long long)gx * c.src0.sW + 4 * seg);
        afi_ch6_store_split(smem_c + pix * AFI_CH6_PIX, 4 * seg, v);
        if (c.copy0.p && inside && (unsigned)(ry - 3) < 8u && (unsigned)(rx - 3) < 8u)
            *(f32x4*)(c.copy0.p + (long long)n * c.copy0.sN + (long long)gy * c.copy0.sH + (long long)gx * c.copy0.sW + 4 * seg) = v;
    }
    __syncthreads();
    afi_chain6_phase<0, MODE>(c, smem_c, n, y0, x0);
    __syncthreads();
    afi_chain6_phase<1, MODE>(c, smem_c, n, y0, x0);
    __syncthreads();
    afi_chain6_phase<2, MODE>(c, smem_c, n, y0, x0);
}
static bool chain6_opt_in() {                               // 100 KB of dynamic LDS: an opt-in per kernel and per device
    static std::mutex mu;
    static std::set<int> done;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lk(mu);
    if (done.count(dev)) return true;
    if (hipFuncSetAttribute((const void*)afi_rdb_chain6_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    if (hipFuncSetAttribute((const void*)afi_rdb_chain6_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    done.insert(dev);
    return true;
}
int afi_launch_rdb_chain6(const AfiChain6& c_in, hipStream_t st) {
    AfiChain6 c = c_in;
    if (c.N <= 0 || c.H <= 0 || c.W <= 0 || (c.a_sgn != 1 && c.a_sgn != -1) || (c.mode != 0 && c.mode != 1) || !c.src0.p) return AFI_ERR_BAD_ARG;
    auto al16 = [](const AfiView& v) { return !(((uintptr_t)v.p) & 15) && !((v.sN | v.sH | v.sW) & 3); };
    if (!al16(c.src0) || (c.copy0.p && !al16(c.copy0))) return AFI_ERR_UNSUPPORTED;
    for (int p = 0; p < 3; ++p) {
        if (!c.ph[p].partial.p || !c.ph[p].out.p || (c.mode == 1 && !c.ph[p].Z.p)) return AFI_ERR_BAD_ARG;
        for (int ci = 0; ci <= p; ++ci) if (!c.ph[p].img[ci] || (((uintptr_t)c.ph[p].img[ci]) & 15) || c.ph[p].stage0[ci] < 0) return AFI_ERR_BAD_ARG;
    }
    c.tiles_y = afi_cdiv(c.H, 8); c.tiles_x = afi_cdiv(c.W, 8);
    const long long blocks = (long long)c.N * c.tiles_y * c.tiles_x;
    if (blocks > 0x7fffffffLL) return AFI_ERR_UNSUPPORTED;
    for (int p = 0; p < 3; ++p) {                          // float4 epilogue accesses
        if (!al16(c.ph[p].partial) || !al16(c.ph[p].out) || (c.mode == 1 && !al16(c.ph[p].Z))) return AFI_ERR_UNSUPPORTED;
    }
    if (!chain6_opt_in()) return AFI_ERR_LAUNCH;
    if (c.mode == 0) hipLaunchKernelGGL(afi_rdb_chain6_kernel<0>, dim3((unsigned)blocks), dim3(1024), (size_t)AFI_CH6_LDS_BYTES, st, c);
    else hipLaunchKernelGGL(afi_rdb_chain6_kernel<1>, dim3((unsigned)blocks), dim3(1024), (size_t)AFI_CH6_LDS_BYTES, st, c);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// the bf16x6 kernel takes a problem when it carries a weight image whose tile range stays inside the kernel's 32-bit buffer offsets
static bool wk6_ok(const AfiPixGemm& p, const AfiWkArgs& wk) {
    if (!p.Bimg || p.bimg_nstages <= 0 || p.bimg_stage0 < 0 || (((uintptr_t)p.Bimg) & 15)) return false;
    if (p.bimg_stage0 + wk.nK > p.bimg_nstages) return false;
    if (p.Ck & 3) return false;
    return 2LL * p.bimg_nstages * AFI_WK6_STAGE_BYTES < 0x7FFFFFF0LL;      // (the 32 x 64 form reaches into the next N tile of the image)
}
static bool wk6_opt_in_wide() {                             // the 32 x 64 form keeps eight 32 x 68 partial tiles: 69.6 KB of dynamic LDS, an opt-in per kernel and device
    static std::mutex mu;
    static std::set<int> done;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lk(mu);
    if (done.count(dev)) return true;
    if (hipFuncSetAttribute((const void*)afi_pix_gemm_wk6w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    if (hipFuncSetAttribute((const void*)afi_pix_gemm_wk6w_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    done.insert(dev);
    return true;
}
// up to AFI_WK_MAXP simultaneous small-map GEMMs in one launch; validates everything before it launches anything
int afi_launch_pix_gemm_wk_group(const AfiPixGemm* probs, int n, int b_rc, hipStream_t st) {
    if (n < 1 || n > AFI_WK_MAXP) return AFI_ERR_UNSUPPORTED;
    AfiWkGroup grp;
    grp.nprob = n;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        const int rc = wk_prepare(probs[i], b_rc != 0, grp.wk[i]);
        if (rc != AFI_OK) return rc;
        grp.p[i] = probs[i];
        grp.tile_start[i] = tiles;
        tiles += grp.wk[i].ntile_m * grp.wk[i].ntile_n;
    }
    for (int i = n; i <= AFI_WK_MAXP; ++i) grp.tile_start[i] = tiles;
    bool all6 = true;
    for (int i = 0; i < n; ++i) all6 = all6 && wk6_ok(probs[i], grp.wk[i]);
    if (all6) {
        // 32 x 64 tiles for the group's wide, long-K problems when the launch still has a block for at least half the CUs that way
        // (the dense block's first step: four 32-column problems + conv5's 256 columns = 108 + 108 blocks instead of 108 + 216)
        int wide_tiles = 0;
        for (int i = 0; i < n; ++i)
            wide_tiles += wk6_wide_candidate(probs[i], grp.wk[i]) ? grp.wk[i].ntile_m * afi_cdiv(probs[i].Ncols, 64) : grp.wk[i].ntile_m * grp.wk[i].ntile_n;
        bool any_wide = false;
        for (int i = 0; i < n; ++i) any_wide = any_wide || wk6_wide_candidate(probs[i], grp.wk[i]);
        if (any_wide && wide_tiles >= 128 && wk6_opt_in_wide()) {
            tiles = 0;
            for (int i = 0; i < n; ++i) {
                if (wk6_wide_candidate(probs[i], grp.wk[i])) wk6_widen(probs[i], grp.wk[i]);
                grp.tile_start[i] = tiles;
                tiles += grp.wk[i].ntile_m * grp.wk[i].ntile_n;
            }
            for (int i = n; i <= AFI_WK_MAXP; ++i) grp.tile_start[i] = tiles;
            hipLaunchKernelGGL(afi_pix_gemm_wk6w_group_kernel, dim3((unsigned)tiles), dim3(512), sizeof(float) * 8 * 32 * (64 + 4), st, grp);
            return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(afi_pix_gemm_wk6_group_kernel, dim3((unsigned)tiles), dim3(512), sizeof(float) * 8 * 32 * (AFI_BK + 4), st, grp);
        return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
    }
    const size_t lds = sizeof(float) * 8 * (2 * 32 * (AFI_BK + 4));
    if (b_rc) hipLaunchKernelGGL((afi_pix_gemm_wk_group_kernel<true>), dim3((unsigned)tiles), dim3(512), lds, st, grp);
    else hipLaunchKernelGGL((afi_pix_gemm_wk_group_kernel<false>), dim3((unsigned)tiles), dim3(512), lds, st, grp);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

template <bool B_RC>
static int launch_wk(const AfiPixGemm& p, hipStream_t st) {
    const long long M = (long long)p.N * p.H * p.W;
    AfiWkArgs wk;
    { const int rc = wk_prepare(p, B_RC, wk); if (rc != AFI_OK) return rc; }
    const long long G = (long long)wk.ntile_m * wk.ntile_n;
    constexpr int LDK = AFI_BK + 4;
    if (wk6_ok(p, wk)) {                                   // bf16x6 on the pre-split weight image (no B_RC distinction: the image holds the k-contiguous fragments)
        // (measured and dropped: sixteen waves per tile -- 1024-thread blocks, four waves per SIMD -- for the grids of at most one block per CU
        //  (216 tiles, K = 2304 / 9216): 13.4 against 12.3 us and 37.5 against 37.5; the loop is bound by vector issue, not by latency)
        if (wk6_wide_candidate(p, wk) && (long long)wk.ntile_m * afi_cdiv(p.Ncols, 64) >= 128 && wk6_opt_in_wide()) {   // 32 x 64 tiles while they cover half the chip
            wk6_widen(p, wk);
            hipLaunchKernelGGL(afi_pix_gemm_wk6w_kernel, dim3((unsigned)(wk.ntile_m * wk.ntile_n)), dim3(512), sizeof(float) * 8 * 32 * (64 + 4), st, p, wk);
            return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
        }
        hipLaunchKernelGGL((afi_pix_gemm_wk6_kernel<false>), dim3((unsigned)G), dim3(512), sizeof(float) * 8 * 32 * LDK, st, p, wk);
        return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
    }
    const size_t lds = sizeof(float) * 8 * (2 * 32 * LDK);
    // one register set, two blocks per CU: measured faster than the two-set variant at every grid size (1.06 -> 1.01 ms at config 1)
    hipLaunchKernelGGL((afi_pix_gemm_wk_kernel<B_RC, true>), dim3((unsigned)G), dim3(512), lds, st, p, wk);
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------------
// grouped weight gradients: one launch, a table of problems, each block owns a whole dW tile (no split over pixels)
// ------------------------------------------------------------------------------------------------
#define AFI_WG_MAXP 20
// Stream-K form: the (tile, 32-pixel stage) units of all problems are laid end to end and cut into EQUAL runs, one per block, with as
// many blocks as the chip holds at once.  A block works through its run tile by tile -- at most a partial tile at each end -- and adds
// a tile it shares with a neighbour by fp32 atomics; tiles it owns alone are stored.  Against "one tile slice per block" (above) this
// removes the partly filled last round: 1044 slices on 768 slots ran as two rounds (191 us for config 1's backward); equal runs need
// 14058 stage times / 256 CUs.
struct AfiWgradGroupSK {
    int nprob, units_per_block, total_units;
    int unit_start[AFI_WG_MAXP + 1];                       // prefix sums of units (tiles * stages) per problem
    short ntile_m[AFI_WG_MAXP], ntile_n[AFI_WG_MAXP];
    int nst[AFI_WG_MAXP];                                  // 32-pixel stages per tile
    AfiWgradGemm g[AFI_WG_MAXP];
};
// The walk of one block's run [u, u_end) over the unit order, shared by the kernel and by afi_debug_wgrad_sk_plan (host): calls
// f(problem, tile, first stage, end stage, shared) once per tile the run touches.  `shared` is false exactly when the run covers every
// stage of the tile -- then no other run touches that tile and its result may be stored; otherwise every run that touches the tile is
// partial and all of them add by atomics.  (Ownership is a property of the run boundaries alone, so it is checked on the host for every
// size the tests and the bench use: tests/test_cabi.py::test_wgrad_stream_k_plan_is_a_partition.)
template <class F>
__host__ __device__ __forceinline__ void afi_sk_walk(const int* unit_start, const int* nst_of, int nprob, int u, int u_end, F&& f) {
    int pi = 0;
    while (u < u_end) {                                    // (uniform) a run crosses a few tile boundaries at most
        while (pi + 1 < nprob && u >= unit_start[pi + 1]) ++pi;
        const int nst = nst_of[pi];
        const int lu = u - unit_start[pi];
        const int tile = lu / nst, s0 = lu - tile * nst;
        int s1 = s0 + (u_end - u);
        if (s1 > nst) s1 = nst;
        f(pi, tile, s0, s1, !(s0 == 0 && s1 == nst));
        u += s1 - s0;
    }
}
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void afi_wgrad_group_sk_kernel(const AfiWgradGroupSK grp) {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
    const int b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);      // neighbours in the unit order share an XCD's L2
    const int u = b * grp.units_per_block;
    int u_end = u + grp.units_per_block;
    if (u_end > grp.total_units) u_end = grp.total_units;
    afi_sk_walk(grp.unit_start, grp.nst, grp.nprob, u, u_end, [&](int pi, int tile, int s0, int s1, bool shared) {
        const AfiWgradGemm& p = grp.g[pi];
        const long long P = (long long)p.N * p.H * p.W;
        const long long k0 = (long long)s0 * AFI_BK, k1 = (long long)s1 * AFI_BK < P ? (long long)s1 * AFI_BK : P;
        afi_wgrad_gemm_range<BM, BN, WM, WN>(p, grp.ntile_m[pi], grp.ntile_n[pi], tile, k0, k1, shared);
    });
}
// run length and grid of a stream-K group launch (one place: the launcher and the debug plan use it)
static inline void sk_cut(long long units, int bpc, int& upb, int& blocks) {
    blocks = 256 * bpc;
    upb = (int)((units + blocks - 1) / blocks);
    if (upb < 4) upb = 4;                                  // tiny groups: no run shorter than four stages
    blocks = (int)((units + upb - 1) / upb);
}
template <int BM, int BN, int WM, int WN>
static int launch_wgrad_group_sk(const AfiWgradGemm* probs, int n, hipStream_t st, int bpc) {
    AfiWgradGroupSK grp;
    int done = 0;
    while (done < n) {
        const int cnt = (n - done) < AFI_WG_MAXP ? (n - done) : AFI_WG_MAXP;
        grp.nprob = cnt;
        long long units = 0;
        for (int i = 0; i < cnt; ++i) {
            const AfiWgradGemm& g = probs[done + i];
            if ((g.Ncols & 3) || (g.dy_up == 2 && (g.CoutPhase & 3))) return AFI_ERR_UNSUPPORTED;
            grp.g[i] = g;
            const long long P = (long long)g.N * g.H * g.W;
            grp.nst[i] = afi_cdiv(P, AFI_BK);
            grp.ntile_m[i] = (short)afi_cdiv(g.Mrows, BM); grp.ntile_n[i] = (short)afi_cdiv(g.Ncols, BN);
            grp.unit_start[i] = (int)units;
            units += (long long)grp.ntile_m[i] * grp.ntile_n[i] * g.ntaps * grp.nst[i];
            if (units > 0x7fffffffLL) return AFI_ERR_UNSUPPORTED;
        }
        for (int i = cnt; i <= AFI_WG_MAXP; ++i) grp.unit_start[i] = (int)units;
        int blocks, upb;
        sk_cut(units, bpc, upb, blocks);
        grp.units_per_block = upb; grp.total_units = (int)units;
        hipLaunchKernelGGL((afi_wgrad_group_sk_kernel<BM, BN, WM, WN>), dim3((unsigned)blocks), dim3(64 * WM * WN), sizeof(float) * AFI_BK * (BM + BN), st, grp);
        done += cnt;
    }
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------------
// bf16x6 form of the grouped weight gradients (128 x 128 tiles): the gather of afi_wgrad_gemm_range (both operands pixel-major, the X
// rows shifted by the tap and masked at the map border) around the LDS images and transposed fragment reads of afi_gemm_tn_bf16_kernel
// (csrc/afi_gemm_bf16.h): 16 pixels x 128 channels of dY and of X at a time (a HALF stage: one MFMA k-step) are split into three bf16 parts
// on their way from the prefetch registers into LDS ([16 k][128 columns] per part, 16-byte chunks swizzled), fragments come out k-fast
// through ds_read_b64_tr_b16, and the six products of relative size >= 2^-16 run on v_mfma_f32_32x32x16_bf16, smallest first, into fp32
// accumulators -- fp32-grade sums at 6/16 of the fp32-MFMA pipe time.  Same stream-K walk as the fp32 kernel (units of 32 pixels).
// The loop is SOFTWARE-PIPELINED inside the wave: two LDS buffers of 24 KB, ONE barrier per half stage; while a wave's 24 MFMAs of half stage
// h run, the same wave splits half stage h + 1 into the other buffer and requests half stage h + 3 into the register set that frees (two
// sets: a gather is requested two half stages before it is split).  The first version (stages of 32 pixels, split between two barriers,
// MFMAs behind the second: every wave of a block in the same phase at the same time) ran as the SUM of its MFMA time and of everything
// else -- 153 us on config 1's ten problems, 95 without the MFMAs (tools/micro/wg6_bench.cpp, profiles/r04/wgrad6_*.txt); this one 121.
// Measured on the way with the in-kernel stamps below: 12.8 us per range in the epilogue (fixed, see there); packed against plain
// subtractions in the split: 1.82 against 1.79 us per half stage (kept plain); 5 / 7 / 9 vector instructions asked behind each MFMA: no difference.
// ------------------------------------------------------------------------------------------------
#ifdef AFI_WG6_DIAG                                        // tools/micro/wg6_bench.cpp only: per block, 10 ns ticks spent in (range prologue, K loop, epilogue), ranges, kernel span
__device__ unsigned long long afi_wg6_stamp[4096][6];
#define AFI_WG6_TICK(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); afi_wg6_stamp[blockIdx.x][i] += t_ - wg6_t_; wg6_t_ = t_; } } while (0)
#else
#define AFI_WG6_TICK(i) do { } while (0)
#endif
#ifndef AFI_WG6_SPLIT
#define AFI_WG6_SPLIT afi_split3_pair_np
#endif
#ifndef AFI_WG6_VALU_PER_MFMA
#define AFI_WG6_VALU_PER_MFMA 7                            // vector instructions the scheduler is asked to place behind each MFMA of a half stage
#endif
__device__ __forceinline__ void afi_wgrad6_gemm_range(const AfiWgradGemm& p, int ntile_m, int ntile_n, int t, long long k_begin, long long k_end, bool use_atomic) {
    constexpr int BM = 128, HK = 16, WN = 2, MI = 2, NI = 2;
    constexpr int PART = HK * BM * 2;                      // one bf16 image of a half stage: [16 k][128 columns], 4 KB
    constexpr int BUF = 6 * PART;                          // [dY: hi | mid | lo][X: hi | mid | lo]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];   // two buffers
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    int tap, tile_n, tile_m;
    tap = t % p.ntaps; t /= p.ntaps;
    tile_n = t % ntile_n; tile_m = t / ntile_n;
    const int m0 = tile_m * BM, n0 = tile_n * 128;
    int dy = 0, dx = 0;
    if (p.ntaps == 9) { dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }
    const int HW = p.H * p.W;
    if (k_begin >= k_end) return;
    const int nH = (int)((k_end - k_begin + HK - 1) / HK);
#ifdef AFI_WG6_DIAG
    unsigned long long wg6_t_ = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { afi_wg6_stamp[blockIdx.x][3] += 1; afi_wg6_stamp[blockIdx.x][4] += nH; }
#endif

    // ---- loader: thread = the eight channels 8 c8 .. + 7 of pixel row kr of a half stage, for both operands: two 16-byte buffer loads per
    //      operand (16 adjacent lanes read 512 contiguous bytes of one pixel), and per part ONE 16-byte LDS store of a whole chunk.
    //      Everything is branch-free -- validity is an out-of-range offset (the buffer returns zeros), the walk over the pixels advances
    //      (y, x) and the two offsets by selects -- and the descriptor fields the loop needs are copied out of the kernel argument block first.
    const int c8 = tid & 15, kr = tid >> 4;
    const int pW = p.W, pH = p.H, xs = p.x_stride, pxH = p.xH, pxW = p.xW;
    const int a_col = m0 + 8 * c8;
    int a_ph = 0, a_ch = a_col;
    if (p.dy_up == 2) { a_ph = a_col / p.CoutPhase; a_ch = a_col - a_ph * p.CoutPhase; }
    const unsigned a_ok0 = a_col < p.Mrows ? 1u : 0u, a_ok1 = a_col + 4 < p.Mrows ? 1u : 0u;
    const int b_col = n0 + 8 * c8;
    const unsigned b_ok0 = b_col < p.Ncols ? 1u : 0u, b_ok1 = b_col + 4 < p.Ncols ? 1u : 0u;
    int a_half = 4;
    if (p.dy_up == 2 && a_ch + 4 >= p.CoutPhase) {
        const int ph1 = a_ph + 1;
        a_half = ((ph1 >> 1) - (a_ph >> 1)) * (int)p.DY.sH + ((ph1 & 1) - (a_ph & 1)) * (int)p.DY.sW - a_ch;
    }
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.DY.p, 0, 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.X.p, 0, 0x7FFFFFF0, 0x00020000);
    const int a_eH = p.dy_up * (int)p.DY.sH, a_eW = p.dy_up * (int)p.DY.sW;
    const int adv_y = HK / pW, adv_x = HK - adv_y * pW;
    const int a_adv = adv_y * a_eH + adv_x * a_eW, a_wrapx = a_eH - pW * a_eW, a_wrapy = (int)p.DY.sN - pH * a_eH;
    const int b_eH = xs * (int)p.X.sH, b_eW = xs * (int)p.X.sW;
    const int b_adv = adv_y * b_eH + adv_x * b_eW, b_wrapx = b_eH - pW * b_eW, b_wrapy = (int)p.X.sN - pH * b_eH;
    const bool single_wrap = adv_y + 1 <= pH;
    int a_off, b_off, py, px;
    {
        const long long pix = k_begin + kr;
        const int n = (int)(pix / HW); const int rem = (int)(pix - (long long)n * HW);
        py = rem / pW; px = rem - py * pW;
        a_off = n * (int)p.DY.sN + py * a_eH + px * a_eW + (a_ph >> 1) * (int)p.DY.sH + (a_ph & 1) * (int)p.DY.sW + a_ch;
        b_off = n * (int)p.X.sN + py * b_eH + dy * (int)p.X.sH + px * b_eW + dx * (int)p.X.sW + b_col;
    }
    int k_left = (int)(k_end - k_begin) - kr;
    u32x4 a_reg[2][2], b_reg[2][2];                        // [set: parity of the half stage][half of the 8-channel group]
    auto prefetch = [&](auto SET, bool more) {
        constexpr int S = decltype(SET)::value;
        unsigned mmv = more ? 1u : 0u;
        asm volatile("" : "+v"(mmv));
        const unsigned in = (k_left > 0) ? mmv : 0u;
        const int yy = py * xs + dy, xx = px * xs + dx;
        const unsigned inb = ((unsigned)yy < (unsigned)pxH && (unsigned)xx < (unsigned)pxW) ? in : 0u;
        const unsigned ao = 4u * (unsigned)a_off, bo = 4u * (unsigned)b_off;
        a_reg[S][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (in & a_ok0) ? ao : 0xFFFFFFFFu, 0, 0);
        a_reg[S][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (in & a_ok1) ? ao + 4u * (unsigned)a_half : 0xFFFFFFFFu, 0, 0);
        b_reg[S][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (inb & b_ok0) ? bo : 0xFFFFFFFFu, 0, 0);
        b_reg[S][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (inb & b_ok1) ? bo + 16u : 0xFFFFFFFFu, 0, 0);
        px += adv_x; py += adv_y; a_off += a_adv; b_off += b_adv;
        if (single_wrap) {
            const bool wx = px >= pW;
            px -= wx ? pW : 0; py += wx ? 1 : 0; a_off += wx ? a_wrapx : 0; b_off += wx ? b_wrapx : 0;
            const bool wy = py >= pH;
            py -= wy ? pH : 0; a_off += wy ? a_wrapy : 0; b_off += wy ? b_wrapy : 0;
        } else {
            if (px >= pW) { px -= pW; ++py; a_off += a_wrapx; b_off += b_wrapx; }
            while (py >= pH) { py -= pH; a_off += a_wrapy; b_off += b_wrapy; }
        }
        k_left -= HK;
    };
    const int st_off = 256 * kr + 16 * (c8 ^ (((kr & 3) << 2) | ((kr >> 2) & 3)));
    auto split_store = [&](auto SET, unsigned char* buf) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int op = 0; op < 2; ++op) {                   // dY, then X
            const f32x4 v0 = __builtin_bit_cast(f32x4, op ? b_reg[S][0] : a_reg[S][0]), v1 = __builtin_bit_cast(f32x4, op ? b_reg[S][1] : a_reg[S][1]);
            unsigned char* base = buf + op * 3 * PART + st_off;
            u32x4 h, m, l;
            unsigned hh, mm_, ll;
            AFI_WG6_SPLIT(v0[0], v0[1], hh, mm_, ll); h[0] = hh; m[0] = mm_; l[0] = ll;
            AFI_WG6_SPLIT(v0[2], v0[3], hh, mm_, ll); h[1] = hh; m[1] = mm_; l[1] = ll;
            AFI_WG6_SPLIT(v1[0], v1[1], hh, mm_, ll); h[2] = hh; m[2] = mm_; l[2] = ll;
            AFI_WG6_SPLIT(v1[2], v1[3], hh, mm_, ll); h[3] = hh; m[3] = mm_; l[3] = ll;
            *(u32x4*)base = h; *(u32x4*)(base + PART) = m; *(u32x4*)(base + 2 * PART) = l;
        }
    };
    int fa_off[MI][2], fb_off[NI][2];
    {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int r = 8 * (g >> 1) + 4 * rd + q;
            const int swz = ((r & 3) << 2) | ((r >> 2) & 3);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) fa_off[mi][rd] = 256 * r + 16 * ((4 * (wm * MI + mi) + 2 * (g & 1) + (pp >> 1)) ^ swz) + 8 * (pp & 1);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) fb_off[ni][rd] = 3 * PART + 256 * r + 16 * ((4 * (wn * NI + ni) + 2 * (g & 1) + (pp >> 1)) ^ swz) + 8 * (pp & 1);
        }
    }
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    auto mm = [](bf16x8 x, bf16x8 y, f32x16 c) -> f32x16 { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0); };

    AFI_WG6_TICK(0);
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    // half stage h lives in register set h & 1 from two iterations before it is split: the gather of h + 3 is requested in iteration h,
    // behind the split of h + 1 that frees its set (a request only one iteration ahead was waited for at the top of every iteration)
    prefetch(S0(), true);                                  // half stage 0
    prefetch(S1(), 1 < nH);                                // half stage 1
    split_store(S0(), smem_b);
    prefetch(S0(), 2 < nH);                                // half stage 2
    __syncthreads();
    auto half_stage = [&](auto NEXT, int h) {              // NEXT: the register set of half stage h + 1
        const unsigned char* cur = smem_b + (h & 1) * BUF;
        unsigned char* nxt = smem_b + ((h + 1) & 1) * BUF;
        bf16x8 ah[MI], am[MI], al[MI], bh[NI], bm[NI], bl[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            ah[mi] = afi_tr_frag(cur, fa_off[mi][0], fa_off[mi][1]);
            am[mi] = afi_tr_frag(cur + PART, fa_off[mi][0], fa_off[mi][1]);
            al[mi] = afi_tr_frag(cur + 2 * PART, fa_off[mi][0], fa_off[mi][1]);
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            bh[ni] = afi_tr_frag(cur, fb_off[ni][0], fb_off[ni][1]);
            bm[ni] = afi_tr_frag(cur + PART, fb_off[ni][0], fb_off[ni][1]);
            bl[ni] = afi_tr_frag(cur + 2 * PART, fb_off[ni][0], fb_off[ni][1]);
        }
        // per accumulator smallest terms first; consecutive MFMAs go to different accumulators (four of them)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = mm(al[mi], bh[ni], acc[mi][ni]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = mm(ah[mi], bl[ni], acc[mi][ni]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = mm(am[mi], bm[ni], acc[mi][ni]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = mm(am[mi], bh[ni], acc[mi][ni]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = mm(ah[mi], bm[ni], acc[mi][ni]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = mm(ah[mi], bh[ni], acc[mi][ni]);
        // the next half stage goes from its registers into the other buffer (every wave left that one at the last barrier); its set then
        // takes the half stage two further on
        split_store(NEXT, nxt);
        prefetch(NEXT, h + 3 < nH);
        // the scheduler's pipeline hint: one MFMA, then a share of the split's vector work
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, AFI_WG6_VALU_PER_MFMA, 0);    // VALU
        }
        __syncthreads();
    };
    for (int h = 0; h < nH; h += 2) {
        half_stage(S1(), h);
        if (h + 1 < nH) half_stage(S0(), h + 1);           // (uniform)
    }
    AFI_WG6_TICK(1);
    // ---- dW += alpha * acc.  Every element goes out as ONE no-return fp32 atomic add, shared tile or not: on a tile this run owns alone
    //      it is the only addend (same bits as load / add / store), and nothing waits for a round trip -- the load / add / store form
    //      the compiler produced for `*dst += v` (a load, s_waitcnt vmcnt(0), a store, and the descriptor re-read from the argument block,
    //      per element: 64 serial round trips per thread) cost 12.8 us per range (in-kernel stamps, tools/micro/wg6_bench.cpp -DAFI_WG6_DIAG)
    {
        (void)use_atomic;
        const int pM = p.Mrows, pN = p.Ncols;
        const float alpha = p.alpha;
        float* const base = p.DW + (long long)tap * p.dw_sTap;
        const long long sRow = p.dw_sRow;
        int cols[NI]; bool cok[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) { cols[ni] = n0 + (wn * NI + ni) * 32 + lr; cok[ni] = cols[ni] < pN; }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float* const rp = base + (long long)row * sRow;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    if (row < pM && cok[ni]) __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(rp + cols[ni]), alpha * acc[mi][ni][r]);
            }
    }
    AFI_WG6_TICK(2);
}
__global__ __launch_bounds__(256, 3) void afi_wgrad6_group_sk_kernel(const AfiWgradGroupSK grp) {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
    const int b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    const int u = b * grp.units_per_block;
    int u_end = u + grp.units_per_block;
    if (u_end > grp.total_units) u_end = grp.total_units;
    afi_sk_walk(grp.unit_start, grp.nst, grp.nprob, u, u_end, [&](int pi, int tile, int s0, int s1, bool shared) {
        const AfiWgradGemm& p = grp.g[pi];
        const long long P = (long long)p.N * p.H * p.W;
        const long long k0 = (long long)s0 * AFI_BK, k1 = (long long)s1 * AFI_BK < P ? (long long)s1 * AFI_BK : P;
        afi_wgrad6_gemm_range(p, grp.ntile_m[pi], grp.ntile_n[pi], tile, k0, k1, shared);
        __syncthreads();                                   // (the next range's first split goes into buffer 0, which a slower wave may still read)
    });
}
// the grouped weight gradients of one small-map backward pass on the bf16 matrix cores (six-product form); 3x3 / 1x1 problems only
int afi_launch_wgrad6_group(const AfiWgradGemm* probs, int n, hipStream_t st) {
    if (n <= 0) return AFI_OK;
    for (int i = 0; i < n; ++i) {                          // 3x3 / 1x1 problems whose operands the kernel's 32-bit element offsets reach
        const AfiWgradGemm& g = probs[i];
        auto span = [&](const AfiView& v, int up, int hh, int ww) { return v.sN * g.N + v.sH * ((long long)hh * up + 2) + v.sW * ((long long)ww * up + 2) + 4096; };
        if (g.ntaps > 9 || g.dy_sTap != 0 || g.x_sTap != 0) return AFI_ERR_UNSUPPORTED;
        if (g.DY.sN < 0 || g.DY.sH < 0 || g.DY.sW < 0 || g.X.sN < 0 || g.X.sH < 0 || g.X.sW < 0) return AFI_ERR_UNSUPPORTED;
        if (span(g.DY, g.dy_up, g.H, g.W) >= (1LL << 28) || span(g.X, g.x_stride, g.xH, g.xW) >= (1LL << 28)) return AFI_ERR_UNSUPPORTED;   // buffer offsets in bytes
        if ((((uintptr_t)g.DY.p | (uintptr_t)g.X.p) & 15) || ((g.DY.sN | g.DY.sH | g.DY.sW | g.X.sN | g.X.sH | g.X.sW) & 3)) return AFI_ERR_UNSUPPORTED;       // 16-byte loads
    }
    constexpr int bpc = 3;
    AfiWgradGroupSK grp;
    int done = 0;
    while (done < n) {
        const int cnt = (n - done) < AFI_WG_MAXP ? (n - done) : AFI_WG_MAXP;
        grp.nprob = cnt;
        // Equal runs over the resident slots, as the fp32 group kernel cuts them.  Measured with tools/micro/wg6_bench.cpp (config 1, run
        // length in stages: time per launch): 16: 163 us, 22 (three blocks per CU, this rule): 143, 27 (= one low-res tile: every tile of a
        // low-res conv owned whole by one run and STORED instead of added by fp32 atomics; 602 blocks): 158, 32 (two per CU): 163, 54: 202
        // -- the atomics of shared tiles are not what the kernel waits for; three resident blocks per CU are what it needs.
        long long units = 0;
        for (int i = 0; i < cnt; ++i) {
            const AfiWgradGemm& g = probs[done + i];
            if ((g.Ncols & 3) || (g.Mrows & 3) || (g.dy_up == 2 && (g.CoutPhase & 3))) return AFI_ERR_UNSUPPORTED;
            grp.g[i] = g;
            grp.nst[i] = afi_cdiv((long long)g.N * g.H * g.W, AFI_BK);
            grp.ntile_m[i] = (short)afi_cdiv(g.Mrows, 128); grp.ntile_n[i] = (short)afi_cdiv(g.Ncols, 128);
            grp.unit_start[i] = (int)units;
            units += (long long)grp.ntile_m[i] * grp.ntile_n[i] * g.ntaps * grp.nst[i];
            if (units > 0x7fffffffLL) return AFI_ERR_UNSUPPORTED;
        }
        for (int i = cnt; i <= AFI_WG_MAXP; ++i) grp.unit_start[i] = (int)units;
        int blocks, upb;
#ifdef AFI_WG6_UPB_OVERRIDE
        if (AFI_WG6_UPB_OVERRIDE > 0) { upb = AFI_WG6_UPB_OVERRIDE; blocks = (int)((units + upb - 1) / upb); } else
#endif
        sk_cut(units, bpc, upb, blocks);
        grp.units_per_block = upb; grp.total_units = (int)units;
        hipLaunchKernelGGL(afi_wgrad6_group_sk_kernel, dim3((unsigned)blocks), dim3(256), 6 * 8192, st, grp);
        done += cnt;
    }
    return hipGetLastError() == hipSuccess ? AFI_OK : AFI_ERR_LAUNCH;
}

// wide = 0: problems with <= 32 rows (the RDB growth convs), 32 x 128 tiles; wide = 1: 128 x 128 tiles.  Problems are sorted by
// the caller, longest pixel range first (the dispatcher hands blocks out in order: long tiles early, short ones fill the tail).
int afi_launch_wgrad_group(const AfiWgradGemm* probs, int n, int wide, hipStream_t st) {
    if (n <= 0) return AFI_OK;
    constexpr int bpc = 3;                                 // resident blocks per CU the stream-K runs are cut for
    return wide ? launch_wgrad_group_sk<128, 128, 1, 4>(probs, n, st, bpc) : launch_wgrad_group_sk<32, 128, 1, 4>(probs, n, st, bpc);
}

// (grouped bias gradients: elementwise.hip, afi_launch_colsum_group / afi_launch_g_bwd_tail)

// ------------------------------------------------------------------------------------------------
// Host-only: the ownership the stream-K cut gives each dW tile of a group of weight-gradient problems (same cut, same walk as the
// kernel above).  Problem i has tiles[i] tiles (M tiles x N tiles x taps) of ceil(pixels[i] / 32) stages.  For tile t (tiles numbered
// problem-major) the outputs are: stored[t] = runs that STORE it (plain read-modify-write), added[t] = runs that ADD to it atomically,
// stages[t] = stages covered by all of them.  A correct plan has (stored, added) = (1, 0) or (0, >= 2) and stages[t] = the tile's stage
// count for every tile: a tile is either owned whole by one run or every run that touches it uses atomics.
// ------------------------------------------------------------------------------------------------
extern "C" int afi_debug_wgrad_sk_plan(const long long* pixels, const int* tiles, int nprob, int bpc, int* stored, int* added, int* stages) {
    if (!pixels || !tiles || nprob <= 0 || bpc <= 0 || !stored || !added || !stages) return AFI_ERR_BAD_ARG;
    long long tile0 = 0;
    for (int done = 0; done < nprob;) {
        const int cnt = (nprob - done) < AFI_WG_MAXP ? (nprob - done) : AFI_WG_MAXP;
        int unit_start[AFI_WG_MAXP + 1], nst[AFI_WG_MAXP];
        long long tile_start[AFI_WG_MAXP + 1];
        long long units = 0, tl = tile0;
        for (int i = 0; i < cnt; ++i) {
            if (pixels[done + i] <= 0 || tiles[done + i] <= 0) return AFI_ERR_BAD_ARG;
            nst[i] = afi_cdiv(pixels[done + i], AFI_BK);
            unit_start[i] = (int)units; tile_start[i] = tl;
            units += (long long)tiles[done + i] * nst[i]; tl += tiles[done + i];
            if (units > 0x7fffffffLL) return AFI_ERR_UNSUPPORTED;
        }
        for (int i = cnt; i <= AFI_WG_MAXP; ++i) unit_start[i] = (int)units;
        for (long long t = tile0; t < tl; ++t) stored[t] = added[t] = stages[t] = 0;
        int upb, blocks;
        sk_cut(units, bpc, upb, blocks);
        for (int b = 0; b < blocks; ++b) {
            const int u = b * upb;
            int u_end = u + upb;
            if (u_end > (int)units) u_end = (int)units;
            afi_sk_walk(unit_start, nst, cnt, u, u_end, [&](int pi, int tile, int s0, int s1, bool shared) {
                const long long t = tile_start[pi] + tile;
                if (shared) ++added[t]; else ++stored[t];
                stages[t] += s1 - s0;
            });
        }
        tile0 = tl;
        done += cnt;
    }
    return AFI_OK;
}
