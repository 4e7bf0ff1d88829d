// bf16-MFMA variants of the two batched Winograd GEMMs (included by igemm.hip after AfiGemmNT / AfiGemmTN / ProfScope).
//
// The planes stay fp32 in HBM (the transforms, the epilogues and every caller are unchanged); an operand is converted to bf16 parts on
// chip and multiplied on the bf16 matrix cores into fp32 accumulators.
//   SPLIT = 1  "bf16":    x -> hi = bf16(x); one MFMA per k-step.  Operand error 2^-9: use on F(2x2) planes only.
//   SPLIT = 3  "bf16x3":  x -> hi + lo, lo = bf16(x - hi); hi*hi + hi*lo + lo*hi (three MFMAs per k-step, the lo*lo term dropped):
//                         operand error 2^-17, so the F(4x4) planes stay usable.  Three bf16 MFMAs cost 3/16 of the fp32 MFMA work.
//   SPLIT = 6  "bf16x6":  x = hi + mid + lo exactly (three bf16 carry all 24 mantissa bits); the six products of order <= 2^-16
//                         (hh, hm, mh, mm, hl, lh), smallest first: what is dropped (ml, lm, ll) is below fp32's own rounding of a
//                         product, so the result is fp32-grade (measured 1e-6, like the fp32 MFMA) at 6/16 of its matrix-core work.
// NT (forward / data gradient): afi_gemm_nt_bf16_dma_kernel further down (LDS-DMA staging, v_mfma_f32_16x16x32_bf16).
// TN (weight gradient): afi_gemm_tn_bf16_kernel (register staging, transposed LDS reads, v_mfma_f32_32x32x16_bf16; three blocks per CU in the
// six-product form).
#pragma once

#include "afi_bf16_split.h"

// ------------------------------------------------------------------------------------------------
// Weight-gradient GEMM  dU[g][m][n] += sum_k Q[g][k][m] * V[g][k][n]: both operands are k-slow in memory, and the bf16 MFMA wants eight
// consecutive k per lane.  The tiles go to LDS as they come -- [32 k][128 columns] bf16, 256-byte rows, the 16-byte chunk ch of row r at
// chunk ch ^ (((r & 3) << 2) | ((r >> 2) & 3)) -- and the fragments are read TRANSPOSED with ds_read_b64_tr_b16: per 16-lane group one
// 4-row x 16-column block, column i of the four rows delivered to lane i.  Two such reads (k 0..3, k 4..7 of the lane's k-group) make
// one operand.  Stores (8 bytes per lane, 32 lanes = one 256-byte row) and the transposed reads are conflict-free on this image.
// Split-K over blockIdx.y with fp32 atomics into dU, exactly like afi_gemm_tn_kernel.
// Round 3, measured against this kernel in one process and rejected (profiles/r03/gemm_tn_variants_ab.log, counters beside it): (1) both
// operands copied verbatim (fp32) by LDS-DMA, eight ds_read_b32 per fragment, split in registers, 128 x 256 tile: the split is then done
// by both waves that share a fragment, 5.8 VALU instructions per MFMA, -6 %; (2) this kernel on v_mfma_f32_16x16x32_bf16: +-3 %;
// (3) operands pre-split into these LDS images by their producers, pure LDS-DMA + transposed reads, no VALU at all, at 2 / 3 / 4 blocks
// per CU: -1 % .. +6 % on the large shapes, -12 .. -27 % on the small ones, before the producers' 1.5x write bytes are counted.  Every
// form holds 0.2 .. 0.27 of MFMA duty per resident wave; none reaches the NT kernel's four waves at 0.2 each.
// ------------------------------------------------------------------------------------------------
template <int SPLIT, bool DB>
__global__ __launch_bounds__(256, 2) void afi_gemm_tn_bf16_kernel(const AfiGemmTN p, int ntile_m, int ntile_n, int kper) {
    constexpr int BM = 128, BN = 128, BK = 32, WN = 2, MI = 2, NI = 2;
    constexpr int NPART = SPLIT == 6 ? 3 : (SPLIT == 3 ? 2 : 1);
    constexpr int TILE = BK * BM * 2;                        // 8 KB: [32 k][128 columns] bf16
    constexpr int BUF = 2 * NPART * TILE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    int t;
    {   // contiguous run of logical ids per XCD; planes slowest, N tiles fastest (as afi_gemm_tn_kernel)
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int tile_n = t % ntile_n; t /= ntile_n;
    const int tile_m = t % ntile_m; const int plane = t / ntile_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const long long k_begin = (long long)blockIdx.y * kper;
    const long long k_end = (k_begin + kper < p.rows_per_plane) ? k_begin + kper : p.rows_per_plane;
    if (k_begin >= k_end) return;                            // (whole block: EXEC stays full for the transposed reads below)
    const int nK = (int)((k_end - k_begin) / BK);
    const int cq = tid & 31, kr = tid >> 5;                  // float4 column, first k row (8 rows per pass, 4 passes)
    const float* a_base = p.Q + ((long long)plane * p.rows_per_plane + k_begin + kr) * p.M + m0 + 4 * cq;
    const float* b_base = p.V + ((long long)plane * p.rows_per_plane + k_begin + kr) * p.N + n0 + 4 * cq;
    const long long a_pass = 8LL * p.M, b_pass = 8LL * p.N, a_stage = (long long)BK * p.M, b_stage = (long long)BK * p.N;

    f32x4 a_reg[4], b_reg[4];
    auto issue = [&](int kc) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a_reg[i] = *(const f32x4*)(a_base + kc * a_stage + i * a_pass);
#pragma unroll
        for (int i = 0; i < 4; ++i) b_reg[i] = *(const f32x4*)(b_base + kc * b_stage + i * b_pass);
    };
    int st_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = kr + 8 * i;
        st_off[i] = 256 * r + 16 * ((cq >> 1) ^ (((r & 3) << 2) | ((r >> 2) & 3))) + 8 * (cq & 1);
    }
    auto stage_store = [&](int buf) {
        unsigned char* base = smem_b + buf * BUF;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(u32x2*)(base + st_off[i]) = afi_pack_bf16(a_reg[i]);
            *(u32x2*)(base + NPART * TILE + st_off[i]) = afi_pack_bf16(b_reg[i]);
            if (SPLIT >= 3) {
                const f32x4 ra = afi_bf16_residual(a_reg[i]), rb = afi_bf16_residual(b_reg[i]);
                *(u32x2*)(base + TILE + st_off[i]) = afi_pack_bf16(ra);
                *(u32x2*)(base + NPART * TILE + TILE + st_off[i]) = afi_pack_bf16(rb);
                if (SPLIT == 6) {
                    *(u32x2*)(base + 2 * TILE + st_off[i]) = afi_pack_bf16(afi_bf16_residual(ra));
                    *(u32x2*)(base + NPART * TILE + 2 * TILE + st_off[i]) = afi_pack_bf16(afi_bf16_residual(rb));
                }
            }
        }
    };
    // transposed-read addresses of k-step 0 (k-step 1: + 16 rows = + 4096 bytes): 32-column tile T, read rd (k 0..3 / 4..7 of the lane's group)
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
            for (int ni = 0; ni < NI; ++ni) fb_off[ni][rd] = NPART * TILE + 256 * r + 16 * ((4 * (wn * NI + ni) + 2 * (g & 1) + (pp >> 1)) ^ swz) + 8 * (pp & 1);
        }
    }
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    issue(0);
    if (DB) { stage_store(0); __syncthreads(); }
    for (int kc = 0; kc < nK; ++kc) {
        const bool more = kc + 1 < nK;
        if (!DB) { stage_store(0); __syncthreads(); }
        if (more) issue(kc + 1);
        const unsigned char* base = smem_b + (DB ? (kc & 1) * BUF : 0);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            // rows 16 s + ...: (r >> 2) & 3 of the swizzle is unchanged by + 16 rows, so k-step 1 is a constant + 4096 bytes
            bf16x8 ah[MI], bh[NI], al[MI], bl[NI], am[MI], bm[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                ah[mi] = afi_tr_frag(base + 4096 * s, fa_off[mi][0], fa_off[mi][1]);
                if (SPLIT == 3) al[mi] = afi_tr_frag(base + 4096 * s + TILE, fa_off[mi][0], fa_off[mi][1]);
                if (SPLIT == 6) { am[mi] = afi_tr_frag(base + 4096 * s + TILE, fa_off[mi][0], fa_off[mi][1]); al[mi] = afi_tr_frag(base + 4096 * s + 2 * TILE, fa_off[mi][0], fa_off[mi][1]); }
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                bh[ni] = afi_tr_frag(base + 4096 * s, fb_off[ni][0], fb_off[ni][1]);
                if (SPLIT == 3) bl[ni] = afi_tr_frag(base + 4096 * s + TILE, fb_off[ni][0], fb_off[ni][1]);
                if (SPLIT == 6) { bm[ni] = afi_tr_frag(base + 4096 * s + TILE, fb_off[ni][0], fb_off[ni][1]); bl[ni] = afi_tr_frag(base + 4096 * s + 2 * TILE, fb_off[ni][0], fb_off[ni][1]); }
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if (SPLIT >= 3) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
                    }
                    if (SPLIT == 6) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[mi], bm[ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bm[ni], acc[mi][ni], 0, 0, 0);
                    }
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                }
        }
        if (DB && more) stage_store((kc + 1) & 1);
        __syncthreads();
    }
    const bool use_atomic = gridDim.y > 1;
    float* out = p.dU + (long long)plane * p.M * p.N;
    const long long ldn = p.N;
    if (use_atomic) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    atomicAdd(out + (long long)row * ldn + n0 + (wn * NI + ni) * 32 + lr, acc[mi][ni][r]);
                }
    } else {
        // dU += acc on a tile this block owns: ALL old values first, then add and store.  Written as `*dst += acc` per element the compiler
        // emitted load / s_waitcnt vmcnt(0) / store 64 times in a row (a later load may alias an earlier store): 64 serial round trips
        // per thread at the end of every tile (found with in-kernel stamps on the small-map weight-gradient kernel, smallmap.hip)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            float old[NI][16];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    old[ni][r] = out[(long long)row * ldn + n0 + (wn * NI + ni) * 32 + lr];
                }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    out[(long long)row * ldn + n0 + (wn * NI + ni) * 32 + lr] = old[ni][r] + acc[mi][ni][r];
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Round 4: the six-product weight-gradient GEMM, SOFTWARE-PIPELINED inside the wave (the scheme of the small-map weight-gradient kernel,
// smallmap.hip: afi_wgrad6_gemm_range, where in-kernel stamps showed the block-phased loop above running as the SUM of its MFMA time and
// of its split / staging time -- every wave of a block splits between the barriers and multiplies behind them, and three resident blocks
// overlap those phases poorly).  Half stages of 16 k rows (one MFMA k-step): two LDS buffers of 24 KB ([dY hi | mid | lo][X hi | mid | lo],
// [16 k][128 columns] bf16 each, the same swizzle and transposed fragment reads as above), ONE barrier per half stage; while a wave's 24
// MFMAs of half stage h run, the same wave splits half stage h + 1 from its registers into the other buffer (pair-wise split, one
// v_cvt_pk_bf16_f32 per part and pair, one 16-byte LDS store per part) and requests half stage h + 3 into the register set that frees.
// Thread = the eight columns 8 c8 .. + 7 of k row kr of a half stage, for both operands (16 adjacent lanes read 512 contiguous bytes).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void afi_gemm_tn_bf16x6_pipe_kernel(const AfiGemmTN p, int ntile_m, int ntile_n, int kper) {
    constexpr int BM = 128, BN = 128, HK = 16, WN = 2, MI = 2, NI = 2;
    constexpr int PART = HK * BM * 2;                        // 4 KB: [16 k][128 columns] bf16
    constexpr int BUF = 6 * PART;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    int t;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int tile_n = t % ntile_n; t /= ntile_n;
    const int tile_m = t % ntile_m; const int plane = t / ntile_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const long long k_begin = (long long)blockIdx.y * kper;
    const long long k_end = (k_begin + kper < p.rows_per_plane) ? k_begin + kper : p.rows_per_plane;
    if (k_begin >= k_end) return;                            // (whole block: EXEC stays full for the transposed reads below)
    const int nH = (int)((k_end - k_begin) / HK);           // (the launcher keeps every K range a multiple of 32 rows)
    const int c8 = tid & 15, kr = tid >> 4;
    const float* a_ptr = p.Q + ((long long)plane * p.rows_per_plane + k_begin + kr) * p.M + m0 + 8 * c8;
    const float* b_ptr = p.V + ((long long)plane * p.rows_per_plane + k_begin + kr) * p.N + n0 + 8 * c8;
    const long long a_step = (long long)HK * p.M, b_step = (long long)HK * p.N;

    f32x4 a_reg[2][2], b_reg[2][2];                          // [set: parity of the half stage][half of the 8-column group]
    auto prefetch = [&](auto SET, bool more) {
        constexpr int S = decltype(SET)::value;
        if (more) {                                          // (uniform; the last two requests of a range are not made)
            a_reg[S][0] = *(const f32x4*)a_ptr; a_reg[S][1] = *(const f32x4*)(a_ptr + 4);
            b_reg[S][0] = *(const f32x4*)b_ptr; b_reg[S][1] = *(const f32x4*)(b_ptr + 4);
        }
        a_ptr += a_step; b_ptr += b_step;
    };
    const int st_off = 256 * kr + 16 * (c8 ^ (((kr & 3) << 2) | ((kr >> 2) & 3)));
    auto split_store = [&](auto SET, unsigned char* buf) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int op = 0; op < 2; ++op) {                     // Q, then V
            const f32x4 v0 = op ? b_reg[S][0] : a_reg[S][0], v1 = op ? b_reg[S][1] : a_reg[S][1];
            unsigned char* base = buf + op * 3 * PART + st_off;
            u32x4 h, m, l;
            unsigned hh, mm_, ll;
            afi_split3_pair_np(v0[0], v0[1], hh, mm_, ll); h[0] = hh; m[0] = mm_; l[0] = ll;
            afi_split3_pair_np(v0[2], v0[3], hh, mm_, ll); h[1] = hh; m[1] = mm_; l[1] = ll;
            afi_split3_pair_np(v1[0], v1[1], hh, mm_, ll); h[2] = hh; m[2] = mm_; l[2] = ll;
            afi_split3_pair_np(v1[2], v1[3], hh, mm_, ll); h[3] = hh; m[3] = mm_; l[3] = ll;
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

    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    prefetch(S0(), true);                                    // half stage 0
    prefetch(S1(), 1 < nH);                                  // half stage 1
    split_store(S0(), smem_b);
    prefetch(S0(), 2 < nH);                                  // half stage 2
    __syncthreads();
    auto half_stage = [&](auto NEXT, int h) {                // NEXT: the register set of half stage h + 1
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
        // per accumulator smallest terms first (the order of the kernel above); consecutive MFMAs go to different accumulators
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
        if (h + 1 < nH) split_store(NEXT, nxt);              // (uniform) every wave left that buffer at the last barrier
        prefetch(NEXT, h + 3 < nH);
        // the scheduler's pipeline hint: one MFMA, then a share of the split's vector work
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
        }
        __syncthreads();
    };
    for (int h = 0; h < nH; h += 2) {
        half_stage(S1(), h);
        if (h + 1 < nH) half_stage(S0(), h + 1);             // (uniform)
    }
    const bool use_atomic = gridDim.y > 1;
    float* out = p.dU + (long long)plane * p.M * p.N;
    const long long ldn = p.N;
    if (use_atomic) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(out + (long long)row * ldn + n0 + (wn * NI + ni) * 32 + lr), acc[mi][ni][r]);
                }
    } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            float old[NI][16];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    old[ni][r] = out[(long long)row * ldn + n0 + (wn * NI + ni) * 32 + lr];
                }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    out[(long long)row * ldn + n0 + (wn * NI + ni) * 32 + lr] = old[ni][r] + acc[mi][ni][r];
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Round 3: the NT GEMM with both operands staged by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no LDS store instructions) on
// v_mfma_f32_16x16x32_bf16.  Measured against the register-staged kernel above in one process (tools/gemm_ab.py, random operands):
// removing the conversion VALU alone changes nothing; removing the VGPR round trip and the ds_write stream is worth +17..20 %.
//   A (the Winograd-domain activations V, fp32 in HBM as before -- the transforms and their traffic are unchanged) is copied VERBATIM into
//   an fp32 LDS image [128 rows][32 floats] and split into bf16 parts when a wave reads its fragment (a wave owns its 32 rows, so each
//   element is split once per block, as before).  128-byte rows, the 16-byte chunk c of row r at chunk c ^ ((r >> 1) & 5): a lane's
//   fragment is chunks 2q, 2q + 1 of row l15 (q = lane >> 4), and both ds_read_b128 are conflict-free (every b128 lane group hits 16
//   distinct slots of the 256-byte bank row; found by exhaustive search over the linear swizzles).  The DMA writes LDS linearly
//   (wave base + 16 * lane), so the swizzle is applied to each lane's SOURCE address.
//   B (the transformed weights U, shared by every M tile of a plane) arrives PRE-SPLIT into bf16 parts in LDS-image order --
//   [plane][N / 128][K / 32][part][128 x 64 bytes], chunk ch of row r at ch ^ ((-(r >> 2)) & 3) (conflict-free for the 16x16x32 fragment
//   reads) -- written once per weight transform by afi_split_bf16_tiles_kernel, so a stage of it is a linear copy.
//   One LDS buffer (16 KB + NPART x 8 KB = 40 KB for the six-product form), two barriers per stage, the next stage's DMA issued behind the
//   second barrier: its latency is covered by the other resident blocks (four per CU at <= 128 registers).  Measured and rejected: two
//   buffers with one barrier per stage and the next stage's DMA in flight under the MFMAs, at two blocks per CU: 8-10 % slower on every
//   shape (profiles/r03/gemm_nt_dma_single_vs_double_buffer.log) -- the fourth and third resident block are worth more than the overlap
//   inside one block, as with the register-staged kernels of rounds 1 and 2.  Also measured: the A split pair-wise (v_cvt_pk + two masks + one
//   packed subtract per pair and part: 9 instead of 15 VALU instructions per pair), bit-identical, +-1 % -- the kernel is not VALU-bound;
//   eight waves of 32 x 64 per block at six waves per SIMD (80 registers): -18 % (half the MFMAs per B fragment read and per barrier).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int afi_bf16_tile16_off(int row, int kq /* float4 column 0..7 */) {
    return row * 64 + ((((kq >> 1) ^ (-(row >> 2))) & 3) << 4) + ((kq & 1) << 3);
}

template <int SPLIT, int MINW>
__global__ __launch_bounds__(256, MINW) void afi_gemm_nt_bf16_dma_kernel(const AfiGemmNT p, int ntile_n, int ntile_m, int chunk) {
    constexpr int BM = 128, BN = 128, BK = 32;
    constexpr int MI = 2, NI = 8;                            // 4 x 1 waves of 32 x 128: a wave's A rows are its own
    constexpr int NPART = SPLIT == 6 ? 3 : (SPLIT == 3 ? 2 : 1);
    constexpr int TILE_A = BM * 128;                         // fp32 image, 16 KB
    constexpr int TILE_B = BN * 64;                          // one bf16 image, 8 KB
    constexpr int OFF_B = TILE_A;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int tile_n = jb % ntile_n, tile_m = xcd * chunk + jb / ntile_n;
    if (tile_m >= ntile_m) return;
    const long long m0 = (long long)tile_m * BM;
    const int n0 = tile_n * BN;
    const int plane = (int)(m0 / p.rows_per_plane);
    const int nK = p.K / BK;
    typedef const __attribute__((address_space(1))) void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
    // A: DMA instruction i of wave w fills rows 8 (4 i + w) .. + 7 (lane >> 3 = row, lane & 7 = physical chunk); source chunk = physical ^ swizzle
    const int a_row0 = 8 * wave + (lane >> 3);
    const int a_swz = ((lane >> 4) & 1) | ((wave & 1) << 2);                    // ((row >> 1) & 5) of every row this lane fills
    const float* a_src = p.A + (m0 + a_row0) * p.K + 4 * ((lane & 7) ^ a_swz);
    const long long a_step = 32LL * p.K;                     // 32 rows per DMA instruction
    const unsigned char* b_src = (const unsigned char*)p.B + (((long long)plane * ntile_n + tile_n) * nK) * (long long)(NPART * TILE_B) + 16 * tid;
    auto issue = [&](int kc) {
        unsigned char* dst = smem_b;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr)(a_src + i * a_step + kc * BK), (lptr)(dst + (4 * i + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2 * NPART; ++i)
            __builtin_amdgcn_global_load_lds((gptr)(b_src + (long long)kc * (NPART * TILE_B) + i * 4096), (lptr)(dst + OFF_B + (4 * i + wave) * 1024), 16, 0, 0);
    };
    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    int fa_off[MI], fb_off[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) { const int row = (wave * MI + mi) * 16 + l15; fa_off[mi] = row * 128 + (((2 * lq) ^ ((row >> 1) & 5)) << 4); }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) { const int row = ni * 16 + l15; fb_off[ni] = OFF_B + row * 64 + (((lq ^ (-(row >> 2))) & 3) << 4); }
    auto mfma = [](bf16x8 x, bf16x8 y, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0); };

    issue(0);
    for (int kc = 0; kc < nK; ++kc) {
        __syncthreads();                                     // (vmcnt(0) in front of the barrier: the stage has landed)
        const unsigned char* sm = smem_b;
        bf16x8 ah[MI], am[MI], al[MI];                       // (SPLIT 3: "am" is the second part)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            // chunks 2q and 2q + 1 of the row: the swizzle never touches bit 0 ... of the PAIR index; bit 0 of the chunk may flip
            const f32x4 c0 = *(const f32x4*)(sm + fa_off[mi]);
            const f32x4 c1 = *(const f32x4*)(sm + (fa_off[mi] ^ 16));
            ah[mi] = afi_pack8_bf16(c0, c1);
            if (SPLIT >= 3) {
                const f32x4 r0 = afi_bf16_residual(c0), r1 = afi_bf16_residual(c1);
                am[mi] = afi_pack8_bf16(r0, r1);
                if (SPLIT == 6) al[mi] = afi_pack8_bf16(afi_bf16_residual(r0), afi_bf16_residual(r1));
            }
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            bf16x8 bh, bm, bl;
            bh = *(const bf16x8*)(sm + fb_off[ni]);
            if (SPLIT >= 3) bm = *(const bf16x8*)(sm + TILE_B + fb_off[ni]);
            if (SPLIT == 6) bl = *(const bf16x8*)(sm + 2 * TILE_B + fb_off[ni]);
            if (SPLIT == 6) {                                // smallest terms first; consecutive MFMAs go to different accumulators
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = mfma(al[mi], bh, acc[mi][ni]);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = mfma(ah[mi], bl, acc[mi][ni]);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = mfma(am[mi], bm, acc[mi][ni]);
            }
            if (SPLIT >= 3) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = mfma(am[mi], bh, acc[mi][ni]);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = mfma(ah[mi], bm, acc[mi][ni]);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = mfma(ah[mi], bh, acc[mi][ni]);
        }
        __syncthreads();
        if (kc + 1 < nK) issue(kc + 1);                      // the buffer is free again
    }
    // epilogue: accumulators -> LDS -> float4 rows of C, 16 rows of every wave per pass
    constexpr int LDC = BN + 4, C_F4 = BN / 4;
    // (the launcher sizes the dynamic LDS as max(stage buffer, 4 * 16 * LDC floats))
    float* Cs = (float*)smem_b;
    float* c_base = p.C + m0 * p.N + n0;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(wave * 16 + lq * 4 + r) * LDC + ni * 16 + l15] = acc[mi][ni][r];
        __syncthreads();
        for (int item = tid; item < 4 * 16 * C_F4; item += 256) {
            const int rloc = item / C_F4, c4 = item - rloc * C_F4;
            const int rl = (rloc >> 4) * 32 + mi * 16 + (rloc & 15);
            // C is written once and read next by another kernel, far beyond L2's reach: nontemporal, so it does not push the A / B tiles that
            // the neighbouring blocks are about to re-read out of the XCD's L2 (FETCH_SIZE -12 %, +1..3 % on the large shapes)
            __builtin_nontemporal_store(*(const f32x4*)(Cs + rloc * LDC + 4 * c4), (f32x4*)(c_base + (long long)rl * p.N + 4 * c4));
        }
        if (mi + 1 < MI) __syncthreads();
    }
}

// B[plane][n][k] fp32 -> the pre-split LDS-image order of afi_gemm_nt_bf16_m16_kernel<.., BPRE = true>: one thread per float4.
template <int SPLIT, int BN>
__global__ __launch_bounds__(256) void afi_split_bf16_tiles_kernel(const float* __restrict__ B, unsigned char* __restrict__ out, int planes, int N, int K) {
    constexpr int NPART = SPLIT == 6 ? 3 : (SPLIT == 3 ? 2 : 1);
    constexpr int TILE_B = BN * 64;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int kq4 = K / 4;
    const long long total = (long long)planes * N * kq4;
    if (i >= total) return;
    const int kq = (int)(i % kq4);
    const long long rowg = i / kq4;                         // plane * N + n
    const int n = (int)(rowg % N), plane = (int)(rowg / N);
    const f32x4 v = *(const f32x4*)(B + rowg * K + 4 * kq);
    const int tile_n = n / BN, row = n - tile_n * BN, kc = kq >> 3;
    unsigned char* img = out + ((((long long)plane * (N / BN) + tile_n) * (K / 32)) + kc) * (long long)(NPART * TILE_B);
    const int off = afi_bf16_tile16_off(row, kq & 7);
    *(u32x2*)(img + off) = afi_pack_bf16(v);
    if (SPLIT >= 3) {
        const f32x4 r1 = afi_bf16_residual(v);
        *(u32x2*)(img + TILE_B + off) = afi_pack_bf16(r1);
        if (SPLIT == 6) *(u32x2*)(img + 2 * TILE_B + off) = afi_pack_bf16(afi_bf16_residual(r1));
    }
}

