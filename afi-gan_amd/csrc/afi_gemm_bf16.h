// bf16-MFMA variants of the two batched Winograd GEMMs (included by igemm.hip after AfiGemmNT / AfiGemmTN / ProfScope).
//
// The planes stay fp32 in HBM (the transforms, the epilogues and every caller are unchanged); a tile is converted to bf16 once, on
// its way from the prefetch registers into LDS, and multiplied with v_mfma_f32_32x32x16_bf16 into fp32 accumulators.
//   SPLIT = 1  "bf16":    x -> hi = bf16(x); one MFMA per k-step.  Operand error 2^-9: use on F(2x2) planes only.
//   SPLIT = 3  "bf16x3":  x -> hi + lo, lo = bf16(x - hi); hi*hi + hi*lo + lo*hi (three MFMAs per k-step, the lo*lo term dropped):
//                         operand error 2^-17, so the F(4x4) planes stay usable.  Three bf16 MFMAs cost 3/16 of the fp32 MFMA work.
//   SPLIT = 6  "bf16x6":  x = hi + mid + lo exactly (three bf16 carry all 24 mantissa bits); the six products of order <= 2^-16
//                         (hh, hm, mh, mm, hl, lh), smallest first: what is dropped (ml, lm, ll) is below fp32's own rounding of a
//                         product, so the result is fp32-grade (measured 1e-6, like the fp32 MFMA) at 6/16 of its matrix-core work.
// At bf16 rate a 128x128x32 stage is 256 (x3: 768) MFMA cycles per wave while its operands are 32 KB of fp32 from L2, so both forms
// are bound by the L2 -> CU stream, not by the matrix cores: the roofline for this kernel is L2 bandwidth x 32 FLOP/B per tile pair.
//
// LDS image of a tile: [128 rows][32 bf16] = 64-byte rows, 16-byte chunk ch of row r stored at chunk ch ^ ((r >> 2) & 3): the
// ds_read_b128 fragment reads (16 lanes = 16 consecutive rows, one chunk) and the ds_write_b64 staging writes are conflict-free.
#pragma once

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32x2 afi_pack_bf16(f32x4 v) {
    const bf16x4 h = __builtin_convertvector(v, bf16x4);
    return __builtin_bit_cast(u32x2, h);
}
__device__ __forceinline__ f32x4 afi_bf16_residual(f32x4 v) {          // v - float(bf16(v)), exact in fp32
    const bf16x4 h = __builtin_convertvector(v, bf16x4);
    return v - __builtin_convertvector(h, f32x4);
}
__device__ __forceinline__ int afi_bf16_tile_off(int row, int kq /* float4 column 0..7 */) {      // byte offset inside an 8 KB tile
    return row * 64 + ((((kq >> 1) ^ (row >> 2)) & 3) << 4) + ((kq & 1) << 3);
}

// DB: two LDS buffers and one barrier per stage (two blocks per CU); !DB: one buffer, two barriers, three blocks per CU whose stages
// interleave -- the form the six-product variant needs (its three images per operand are 48 KB per buffer).
template <int SPLIT, bool DB>
__global__ __launch_bounds__(256, DB ? 2 : 3) void afi_gemm_nt_bf16_kernel(const AfiGemmNT p, int ntile_n, int ntile_m, int chunk) {
    constexpr int BM = 128, BN = 128, BK = 32, WN = 2, MI = 2, NI = 2;
    constexpr int NPART = SPLIT == 6 ? 3 : (SPLIT == 3 ? 2 : 1);   // hi (, mid) (, lo) images per operand
    constexpr int TILE = BM * BK * 2;                        // bytes of one bf16 tile image
    constexpr int BUF = 2 * NPART * TILE;                    // A parts then B parts
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int tile_n = jb % ntile_n, tile_m = xcd * chunk + jb / ntile_n;
    if (tile_m >= ntile_m) return;
    const long long m0 = (long long)tile_m * BM;
    const int n0 = tile_n * BN;
    const int plane = (int)(m0 / p.rows_per_plane);
    const int aq = tid & 7, ar = tid >> 3;                   // float4 column, first row (32 rows per pass, 4 passes)
    const float* a_base = p.A + (m0 + ar) * p.K + 4 * aq;
    const float* b_base = p.B + ((long long)plane * p.N + n0 + ar) * p.K + 4 * aq;
    const long long pass = 32LL * p.K;
    const int nK = p.K / BK;

    f32x4 a_reg[4], b_reg[4];
    auto issue = [&](int kc) {
        const int k0 = kc * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) a_reg[i] = *(const f32x4*)(a_base + i * pass + k0);
#pragma unroll
        for (int i = 0; i < 4; ++i) b_reg[i] = *(const f32x4*)(b_base + i * pass + k0);
    };
    auto stage_store = [&](int buf) {
        unsigned char* base = smem_b + buf * BUF;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = afi_bf16_tile_off(ar + 32 * i, aq);
            *(u32x2*)(base + off) = afi_pack_bf16(a_reg[i]);
            *(u32x2*)(base + NPART * TILE + off) = afi_pack_bf16(b_reg[i]);
            if (SPLIT >= 3) {
                const f32x4 ra = afi_bf16_residual(a_reg[i]), rb = afi_bf16_residual(b_reg[i]);
                *(u32x2*)(base + TILE + off) = afi_pack_bf16(ra);
                *(u32x2*)(base + NPART * TILE + TILE + off) = afi_pack_bf16(rb);
                if (SPLIT == 6) {
                    *(u32x2*)(base + 2 * TILE + off) = afi_pack_bf16(afi_bf16_residual(ra));
                    *(u32x2*)(base + NPART * TILE + 2 * TILE + off) = afi_pack_bf16(afi_bf16_residual(rb));
                }
            }
        }
    };
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // fragment addresses: lane (lr, lh) takes row (tile32 * 32 + lr), k = 16 s + 8 lh .. +7 = chunk 2 s + lh
    int fa_off[MI][2], fb_off[NI][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) { const int row = (wm * MI + mi) * 32 + lr; fa_off[mi][s] = row * 64 + ((((2 * s + lh) ^ (row >> 2)) & 3) << 4); }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) { const int row = (wn * NI + ni) * 32 + lr; fb_off[ni][s] = NPART * TILE + row * 64 + ((((2 * s + lh) ^ (row >> 2)) & 3) << 4); }
    }

    issue(0);
    if (DB) { stage_store(0); __syncthreads(); }
    for (int kc = 0; kc < nK; ++kc) {
        const bool more = kc + 1 < nK;
        if (!DB) { stage_store(0); __syncthreads(); }
        if (more) issue(kc + 1);                             // in flight behind this stage's MFMAs (and the other blocks of the CU)
        const unsigned char* base = smem_b + (DB ? (kc & 1) * BUF : 0);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 ah[MI], bh[NI], al[MI], bl[NI], am[MI], bm[NI];      // (SPLIT 3: "l" is the second part; SPLIT 6: h, m, l)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                ah[mi] = *(const bf16x8*)(base + fa_off[mi][s]);
                if (SPLIT == 3) al[mi] = *(const bf16x8*)(base + TILE + fa_off[mi][s]);
                if (SPLIT == 6) { am[mi] = *(const bf16x8*)(base + TILE + fa_off[mi][s]); al[mi] = *(const bf16x8*)(base + 2 * TILE + fa_off[mi][s]); }
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                bh[ni] = *(const bf16x8*)(base + fb_off[ni][s]);
                if (SPLIT == 3) bl[ni] = *(const bf16x8*)(base + TILE + fb_off[ni][s]);
                if (SPLIT == 6) { bm[ni] = *(const bf16x8*)(base + TILE + fb_off[ni][s]); bl[ni] = *(const bf16x8*)(base + 2 * TILE + fb_off[ni][s]); }
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if (SPLIT >= 3) {                        // small terms first
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
        if (DB && more) stage_store((kc + 1) & 1);           // the other buffer: its readers passed the barrier of the previous stage
        __syncthreads();
    }
    // epilogue: accumulators -> LDS -> float4 rows of C   (same staging as afi_gemm_nt_kernel)
    constexpr int LDC = BN + 4, C_F4 = BN / 4;
    float* Cs = (float*)smem_b;
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

static size_t afi_gemm_nt_bf16_lds(int split, bool db) {
    const size_t ring = (db ? 2u : 1u) * 2u * (split == 6 ? 3u : (split == 3 ? 2u : 1u)) * 128u * 32u * 2u;      // buffers x (A, B) x parts x tile
    const size_t cst = sizeof(float) * 64u * (128u + 4u);
    return ring > cst ? ring : cst;
}

// ------------------------------------------------------------------------------------------------
// Weight-gradient GEMM  dU[g][m][n] += sum_k Q[g][k][m] * V[g][k][n]: both operands are k-slow in memory, and the bf16 MFMA wants eight
// consecutive k per lane.  The tiles go to LDS as they come -- [32 k][128 columns] bf16, 256-byte rows, the 16-byte chunk ch of row r at
// chunk ch ^ (((r & 3) << 2) | ((r >> 2) & 3)) -- and the fragments are read TRANSPOSED with ds_read_b64_tr_b16: per 16-lane group one
// 4-row x 16-column block, column i of the four rows delivered to lane i.  Two such reads (k 0..3, k 4..7 of the lane's k-group) make
// one operand.  Stores (8 bytes per lane, 32 lanes = one 256-byte row) and the transposed reads are conflict-free on this image.
// Split-K over blockIdx.y with fp32 atomics into dU, exactly like afi_gemm_tn_kernel.
// ------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 afi_tr_frag(const unsigned char* base, int off_lo, int off_hi) {
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + off_lo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + off_hi));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

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
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float* dst = out + (long long)row * p.N + n0 + (wn * NI + ni) * 32 + lr;
                if (use_atomic) atomicAdd(dst, acc[mi][ni][r]); else *dst += acc[mi][ni][r];
            }
}
