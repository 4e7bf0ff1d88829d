// "f16x3": the two batched Winograd GEMMs with each fp32 operand split into TWO fp16 pieces and three products (included by igemm.hip
// after afi_gemm_bf16.h).
//
//   x' = x * s                    s a power of two chosen per operand and Winograd plane so that |x'| < 2^15 (fp16's largest finite value
//                                 is 65504); the multiplication is exact
//   hi = f16(x'), lo = f16(x' - hi)     round to nearest; the residual is exact in fp32 and has at most 13 significant bits, of which lo
//                                 keeps 11: |x' - hi - lo| <= 2^-23 |x'| wherever lo is a normal fp16 number, <= 2^-25 absolute below
//   a.b ~ (lo_a.hi_b + hi_a.lo_b + hi_a.hi_b) / (s_a s_b)      three v_mfma_f32_*_f16 into one fp32 accumulator, smallest terms first;
//                                 fp16 x fp16 products are exact in the accumulator's fp32; the dropped lo.lo term is <= 2^-22 of the
//                                 product (random sign, rms 2^-23.6: below the accumulator's own rounding of a sum of such products)
//
// against bf16x6 (afi_gemm_bf16.h: three bf16 pieces, six products): half the matrix-core work, two thirds of the split's vector work,
// two thirds of the pre-split weight bytes.  What bf16 had for free and fp16 does not is RANGE: an fp16 piece is normal only between
// 2^-14 and 2^16, so the operand is scaled first.  The scale needs the operand's largest magnitude, which is not known when its tiles
// stream by, so it comes from an upper BOUND that is known before the GEMM starts:
//   * activations (A of the NT GEMM; both operands of the TN GEMM) are Winograd transforms of a tensor whose largest magnitude `amax` the
//     transform kernel computes as a by-product of the loads it does anyway (one atomic max per block, winograd.hip); every plane a of
//     the transform satisfies |V[a]| <= c_a amax with c_a the product of the absolute row sums of the transform matrix (F(2x2): 4;
//     F(4x4): 36 .. 100; the dY transforms: <= 1), and s_a = 2^(14 - floor(log2(c_a amax))).  A bound that is loose by a factor L costs
//     log2(L) binades of the range below the largest element, never precision of the elements that matter: lo stays a normal number for
//     every element within 2^-18 / L of the bound, and below that the ABSOLUTE error of an element is 2^-25 / s = 2^-40 L of the bound --
//     fp32's own rounding of the plane's large elements is 2^-24 of them.
//   * weights (B of the NT GEMM) are split once per weight transform by afi_split_f16_tiles_kernel with the exact per-plane maximum.
// The accumulators hold s_a s_b C; the epilogue multiplies by 1 / s_a and 1 / s_b (exact; two steps so that neither factor leaves
// fp32's range).
#pragma once

#include "afi_f16_split.h"
__device__ __forceinline__ f16x8 afi_tr_frag_f16(const unsigned char* base, int off_lo, int off_hi) {
    return __builtin_bit_cast(f16x8, afi_tr_frag(base, off_lo, off_hi));
}

// ------------------------------------------------------------------------------------------------
// largest magnitude per plane of X [planes][per_plane] (per_plane a multiple of 4), as the bit pattern of a non-negative float under an
// unsigned atomic max (monotonic): out[plane] must be zero-filled before the launch.  grid = (blocks per plane, planes).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float afi_wave_max(float m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return m;
}
// every thread brings its own maximum; thread 0 publishes the block's, and only if it can still raise the slot (a relaxed device-scope
// read first: after the first blocks most of a launch's blocks add nothing, and thousands of atomics on ONE word would serialise at the
// memory side, ~12 ns each)
__device__ __forceinline__ void afi_block_amax_publish(float m, float* slot) {
    __shared__ float red[16];
    m = afi_wave_max(m);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if (lane == 0) red[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w) m = fmaxf(m, red[w]);
        const unsigned bits = __float_as_uint(m);
        const unsigned cur = __hip_atomic_load((const unsigned*)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bits > cur) (void)atomicMax((unsigned*)slot, bits);
    }
}
__global__ __launch_bounds__(256) void afi_absmax_planes_kernel(const float* __restrict__ X, long long per_plane, float* __restrict__ out) {
    const float* x = X + (long long)blockIdx.y * per_plane;
    const long long n4 = per_plane >> 2;
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = *(const f32x4*)(x + 4 * i);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    afi_block_amax_publish(m, out + blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// B[plane][n][k] fp32 -> [header: the plane scales s_b][fp16 pieces in the LDS-image order of the NT kernel below]:
// [plane][N / 128][K / 32][hi | lo][128 rows x 64 bytes], 16-byte chunk ch of row r at ch ^ ((-(r >> 2)) & 3) (afi_bf16_tile16_off).
// bb: the planes' bounds -- their exact maxima (afi_absmax_planes_kernel), or the weight tensor's largest magnitude times the transform's constants.
// One thread per float4.
// ------------------------------------------------------------------------------------------------
#define AFI_F16_HDR_BYTES 512                               // floats [0, 64): the plane scales; [64, 128): the plane maxima they were made from
template <int BN>
__global__ __launch_bounds__(256) void afi_split_f16_tiles_kernel(const float* __restrict__ B, unsigned char* __restrict__ out, const AfiF16Bound bb,
                                                                  int planes, int N, int K) {
    constexpr int TILE_B = BN * 64;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int kq4 = K / 4;
    const long long total = (long long)planes * N * kq4;
    if (i >= total) return;
    const int kq = (int)(i % kq4);
    const long long rowg = i / kq4;                         // plane * N + n
    const int n = (int)(rowg % N), plane = (int)(rowg / N);
    const float s = afi_f16_scale(bb.amax[(long long)plane * bb.stride] * bb.cmul[plane]);
    if (n == 0 && kq == 0) ((float*)out)[plane] = s;
    const f32x4 v = *(const f32x4*)(B + rowg * K + 4 * kq);
    const int tile_n = n / BN, row = n - tile_n * BN, kc = kq >> 3;
    unsigned char* img = out + AFI_F16_HDR_BYTES + ((((long long)plane * (N / BN) + tile_n) * (K / 32)) + kc) * (long long)(2 * TILE_B);
    const int off = afi_bf16_tile16_off(row, kq & 7);
    unsigned h0, h1, l0, l1;
    afi_split2_f16_pair(v[0], v[1], s, h0, l0);
    afi_split2_f16_pair(v[2], v[3], s, h1, l1);
    *(u32x2*)(img + off) = u32x2{h0, h1};
    *(u32x2*)(img + TILE_B + off) = u32x2{l0, l1};
}

// ------------------------------------------------------------------------------------------------
// NT GEMM  C[g][m][n] = sum_k A[g][m][k] B[g][n][k].  Tile, LDS images, fragment reads and epilogue of afi_gemm_nt_bf16_dma_kernel
// (afi_gemm_bf16.h: both operands staged by LDS-DMA, A verbatim as fp32 with the swizzle on the source address and split when a wave
// reads its fragment, B pre-split in LDS-image order; v_mfma_f32_16x16x32, four blocks per CU, one 32 KB stage buffer: 16 A + 2 x 8 B)
// with two fp16 pieces: 48 MFMAs per wave and stage instead of 96, the scale in the split and its inverse in the epilogue.
//
// With half the MFMAs per stage the stage's fixed chain shows: ablations of the first version (the bf16 kernel's loop: wait for the whole
// stage, barrier, read + split + multiply, barrier, request the next stage; tools/micro/nt_f16_ablate.py, 36 x 8448 x 1024 x 1024: 1792 us)
// ran 1274 us without the DMA behind the first stage, 1368 without the split, 1101 without the MFMAs and 1544 with L2-resident operands:
// a block's DMA wait, its split and its MFMAs ran one after the other, and four blocks per CU did not interleave them away.  So the loop
// is ordered for overlap INSIDE the block, with the same buffer and the same two barriers:
//   * a wave's A rows are its own (4 x 1 waves of 32 x 128), so the wave that reads them also fetches them (DMA instruction i of wave w
//     fills rows 32 w + 8 i ..): no barrier guards A, the wave's own vmcnt does, and A of stage k + 1 is requested as soon as the wave
//     holds stage k's fragments in registers -- it arrives under the stage's 48 MFMAs;
//   * B of stage k + 1 is requested behind the second barrier as before, and what waits for it is only the multiply phase: A's fragment
//     reads and the split of stage k + 1 run first, under a counted vmcnt (A was requested earlier, requests complete in order).
// The split itself: nine full-rate instructions per pair (v_cvt_pk_f16_f32 and friends) beat four v_fma_mix{lo,hi}_f16 (1648 against
// 1792 us): the mix forms do not issue at the full rate.
// ABL (micro-benchmark ablations, a bit mask; results wrong but for bit 4): 1 every block reads the operands of tile (0, 0) (L2-resident),
// 2 no DMA behind the first stage, 4 no MFMAs, 8 no split (the fragment registers are reinterpreted), 16 the split from v_fma_mix instructions
// ------------------------------------------------------------------------------------------------
// APRE: A arrives already split by its producer (winograd.hip, afi_store_split4): a row's 128 bytes per stage are [hi: 32 x fp16 | lo: 32 x fp16]
// instead of 32 fp32, staged by the same DMA pattern into the same 16 KB image (16-byte chunk c of row r at c ^ ((r >> 1) & 7): conflict-free
// for the two fragment reads of a lane, chunk q of hi and chunk 4 + q of lo; exhaustive search over the linear swizzles), and the loop
// has no conversion instruction left.
template <int MINW, int ABL = 0, bool APRE = false>
__global__ __launch_bounds__(256, MINW) void afi_gemm_nt_f16x3_kernel(const AfiGemmNT p, const AfiF16Bound ab, int ntile_n, int ntile_m, int chunk) {
    constexpr int BM = 128, BN = 128, BK = 32;
    constexpr int MI = 2, NI = 8;                            // 4 x 1 waves of 32 x 128: a wave's A rows are its own
    constexpr int TILE_A = BM * 128;                         // fp32 image, 16 KB
    constexpr int TILE_B = BN * 64;                          // one fp16 image, 8 KB
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
    // A: DMA instruction i of wave w fills rows 32 w + 8 i .. + 7 -- the wave's own rows (lane >> 3 = row, lane & 7 = physical chunk; source
    // chunk = physical ^ swizzle, ((row >> 1) & 5) of row 32 w + 8 i + (lane >> 3): bit 0 = lane bit 4, bit 2 = i bit 0)
    const float* a_src = p.A + (((ABL & 1) ? 0 : m0) + 32 * wave + (lane >> 3)) * p.K;
    // (APRE: ((row >> 1) & 7) = lane bits 4..5 | i bit 0 << 2)
    const int a_c0 = 4 * ((lane & 7) ^ (APRE ? (lane >> 4) : ((lane >> 4) & 1))), a_c1 = a_c0 ^ 16;                   // even / odd i (in floats)
    const long long a_step = 8LL * p.K;
    const unsigned char* b_hdr = (const unsigned char*)p.B;
    const unsigned char* b_src = b_hdr + AFI_F16_HDR_BYTES + ((ABL & 1) ? 0 : (((long long)plane * ntile_n + tile_n) * nK)) * (long long)(2 * TILE_B) + 16 * tid;
    auto issue_a = [&](int kc) {
        if ((ABL & 2) && kc > 0) return;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr)(a_src + i * a_step + kc * BK + ((i & 1) ? a_c1 : a_c0)), (lptr)(smem_b + (4 * wave + i) * 1024), 16, 0, 0);
    };
    auto issue_b = [&](int kc) {
        if ((ABL & 2) && kc > 0) return;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr)(b_src + (long long)kc * (2 * TILE_B) + i * 4096), (lptr)(smem_b + OFF_B + (4 * i + wave) * 1024), 16, 0, 0);
    };
    issue_a(0);
    issue_b(0);
    // the plane's scales (block-uniform: scalar loads, under the first stage's flight)
    const float s_a = afi_f16_scale(ab.amax[(long long)plane * ab.stride] * ab.cmul[plane]);
    const float inv_a = afi_pow2_inverse(s_a), inv_b = afi_pow2_inverse(((const float*)b_hdr)[plane]);
    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    int fa_off[MI], fb_off[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int row = (wave * MI + mi) * 16 + l15;
        fa_off[mi] = APRE ? row * 128 + ((lq ^ ((row >> 1) & 7)) << 4) : row * 128 + (((2 * lq) ^ ((row >> 1) & 5)) << 4);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) { const int row = ni * 16 + l15; fb_off[ni] = OFF_B + row * 64 + (((lq ^ (-(row >> 2))) & 3) << 4); }
    auto mfma = [](f16x8 x, f16x8 y, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c, 0, 0, 0); };

    // vmcnt bookkeeping (requests complete in order; 4 DMA instructions per operand and stage): at the top of stage k the wave has at most
    // A(k), B(k) outstanding -> vmcnt(4) = A(k) has landed; behind the request of A(k + 1): B(k), A(k + 1) -> vmcnt(4) = B(k) has landed
    // (vmcnt(0) in the last stage, where nothing is requested behind it).  Nothing but these waits orders a ds_read behind an LDS-DMA.
    for (int kc = 0; kc < nK; ++kc) {
        const bool more = kc + 1 < nK;
        if (ABL & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        const unsigned char* sm = smem_b;
        f16x8 ah[MI], al[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            if (APRE) {
                ah[mi] = *(const f16x8*)(sm + fa_off[mi]);
                al[mi] = *(const f16x8*)(sm + (fa_off[mi] ^ 64));
                continue;
            }
            const f32x4 c0 = *(const f32x4*)(sm + fa_off[mi]);
            const f32x4 c1 = *(const f32x4*)(sm + (fa_off[mi] ^ 16));
            u32x4 h, l;
            if (ABL & 8) { h = __builtin_bit_cast(u32x4, c0); l = __builtin_bit_cast(u32x4, c1); }
            else if (ABL & 16) afi_split2_f16_x8(c0, c1, s_a, h, l);
            else afi_split2_f16_x8_cvt(c0, c1, s_a, h, l);
            ah[mi] = __builtin_bit_cast(f16x8, h);
            al[mi] = __builtin_bit_cast(f16x8, l);
        }
        if (more) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave's A rows are in registers: their LDS rows are free
            issue_a(kc + 1);
        }
        // (pin the split in front of the wait for B: left alone, hipcc sinks it behind the wait and the barrier, where nothing covers it)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) asm volatile("" :: "v"(ah[mi]), "v"(al[mi]));
        if (more && !(ABL & 2)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // every wave's share of B(k) has landed
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const f16x8 bh = *(const f16x8*)(sm + fb_off[ni]);
            const f16x8 bl = *(const f16x8*)(sm + TILE_B + fb_off[ni]);
            if (ABL & 4) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) asm volatile("" :: "v"(al[mi]), "v"(ah[mi]), "v"(bh), "v"(bl));
                continue;
            }
            // smallest terms first; consecutive MFMAs go to different accumulators
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = mfma(al[mi], bh, acc[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = mfma(ah[mi], bl, acc[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = mfma(ah[mi], bh, acc[mi][ni]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's B reads have returned
        __builtin_amdgcn_s_barrier();                        // ... and every wave's: B's buffer is free
        if (more) issue_b(kc + 1);
    }
    // epilogue: accumulators / (s_a s_b) -> LDS -> float4 rows of C, 16 rows of every wave per pass
    constexpr int LDC = BN + 4, C_F4 = BN / 4;
    float* Cs = (float*)smem_b;
    float* c_base = p.C + m0 * p.N + n0;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(wave * 16 + lq * 4 + r) * LDC + ni * 16 + l15] = (acc[mi][ni][r] * inv_a) * inv_b;
        __syncthreads();
        for (int item = tid; item < 4 * 16 * C_F4; item += 256) {
            const int rloc = item / C_F4, c4 = item - rloc * C_F4;
            const int rl = (rloc >> 4) * 32 + mi * 16 + (rloc & 15);
            __builtin_nontemporal_store(*(const f32x4*)(Cs + rloc * LDC + 4 * c4), (f32x4*)(c_base + (long long)rl * p.N + 4 * c4));
        }
        if (mi + 1 < MI) __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// The same GEMM on a 256 x 256 tile: SIXTEEN waves of the kernel above in one 1024-thread block (8 x 2 waves of 32 x 128; the per-wave
// work -- fragment reads, split, 48 MFMAs per stage, 64 accumulator registers, 4 waves per SIMD -- is the same).  Why: the ablations of the
// 128 x 128 kernel (tools/micro/nt_f16_ablate.py, 36 x 8448 x 1024 x 1024, 1790 us) say that with three products per k-step the kernel
// has TWO resources that each need ~1100 us: the DMA transfers alone (no MFMAs, no split) take 1085 us -- 20 GB from L2 into LDS at
// 18.4 TB/s, the chip's LDS-DMA ceiling for L2-resident rows --, the MFMAs with their fragment reads and barriers alone 1138 us.  Four
// independent 128 x 128 blocks per CU fetch each operand tile twice into the same CU; one 256 x 256 block fetches it once: half the bytes
// per product.  Two 64 KB stage buffers ([A fp32: 256 rows x 128 B][B hi: 256 rows x 64 B][B lo]), ONE barrier per stage: it says that
// stage k has landed for every wave and that every wave has left stage k - 1's buffer, so stage k + 1 is requested right behind it (four
// DMA instructions per wave) and lands under the 48 MFMAs of stage k.  The epilogue stages a 16-row strip per wave (16 x 8448 B).
// One block per CU.  Rows beyond a plane's end (rows_per_plane is a multiple of 128, not 256) are read from the plane's last row and never
// stored.  B comes from the two 128-column images 2 tile_n, 2 tile_n + 1 ([hi 8 KB | lo 8 KB] per stage each).
// ------------------------------------------------------------------------------------------------
// LOCAL (AFI_OPT_F16_LOCAL_SUMS): the three products of a k-step are summed in a FRESH fragment (the first MFMA takes a zero C) and that
// fragment is added to the accumulator by one fp32 vector add, instead of three MFMAs accumulating into it.  Same products; what changes
// is how often the large accumulator is rounded: an fp32 accumulate rounds at the magnitude of the accumulator, whatever the addend, so
// three accumulating MFMAs per k-step (each with its own internal passes) round it 3+ times per k-step and the local form ONCE -- the
// fresh fragment holds one k-step's 96 products, 1 / sqrt(K / 32) of the final magnitude, where roundings cost nothing.  Measured on the
// discriminator's gradients (their LeakyReLU masks are decided by the forward convs' rounding): DESIGN.md 4b.  8 more registers, 64 vector
// adds per wave and stage beside 48 MFMAs, no extra staging (a two-walk form -- all cross products first, then all hi x hi -- bought the
// same accuracy for 1.5x the staging and was 1.7x slower: profiles/r06/two_walk_gemm_rejected.txt).
template <bool APRE, bool LOCAL = false>
__global__ __launch_bounds__(1024, 4) void afi_gemm_nt_f16x3_w16_kernel(const AfiGemmNT p, const AfiF16Bound ab, int ntile_n, int ntile_m, int chunk, int tiles_per_plane) {
    constexpr int BM = 256, BK = 32;
    constexpr int MI = 2, NI = 8;
    constexpr int TILE_A = BM * 128;                         // fp32 image, 32 KB
    constexpr int HALF_B = 128 * 64;                         // one fp16 piece of one 128-column image, 8 KB
    constexpr int PART_B = 256 * 64;                         // one piece of the block's 256 columns, 16 KB
    constexpr int OFF_B = TILE_A;
    constexpr int STAGE = TILE_A + 2 * PART_B;               // 64 KB
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lq = lane >> 4;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int tile_n = jb % ntile_n, tile_m = xcd * chunk + jb / ntile_n;
    if (tile_m >= ntile_m) return;
    const int plane = tile_m / tiles_per_plane, tp = tile_m - plane * tiles_per_plane;
    const long long row0 = (long long)plane * p.rows_per_plane + (long long)tp * BM;
    const int valid = (int)(p.rows_per_plane - (long long)tp * BM < BM ? p.rows_per_plane - (long long)tp * BM : BM);      // 128 or 256 rows of this tile exist
    const int n0 = tile_n * 256;
    const int nK = p.K / BK;
    typedef const __attribute__((address_space(1))) void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
    // A: DMA instruction i (0, 1) of wave w fills KB 2 w + i of the image = rows 8 (2 w + i) .. + 7 (lane >> 3 = row, lane & 7 = physical chunk;
    // source chunk = physical ^ ((row >> 1) & 5): bit 0 = lane bit 4, bit 2 = i)
    int ar0 = 16 * wave + (lane >> 3), ar1 = ar0 + 8;
    ar0 = ar0 < valid ? ar0 : valid - 1; ar1 = ar1 < valid ? ar1 : valid - 1;
    // (APRE, afi_gemm_nt_f16x3_kernel: swizzle (row >> 1) & 7 = lane bits 4..5 | i << 2)
    const float* a_src0 = p.A + (row0 + ar0) * p.K + 4 * ((lane & 7) ^ (APRE ? (lane >> 4) : ((lane >> 4) & 1)));
    const float* a_src1 = p.A + (row0 + ar1) * p.K + 4 * ((lane & 7) ^ (APRE ? (lane >> 4) : ((lane >> 4) & 1)) ^ 4);
    // B: DMA instruction j (0, 1) of wave w fills KB 16 j + w of the stage's B region: piece j, 128-column half w >> 3, KB w & 7 of that half
    const unsigned char* b_hdr = (const unsigned char*)p.B;
    const long long b_tile = (long long)nK * (2 * HALF_B);   // bytes of one 128-column image of one plane
    const unsigned char* b_src = b_hdr + AFI_F16_HDR_BYTES + ((long long)plane * (2 * ntile_n) + 2 * tile_n + (wave >> 3)) * b_tile + (wave & 7) * 1024 + 16 * lane;
    auto issue = [&](int kc) {
        unsigned char* dst = smem_b + (kc & 1) * STAGE;
        __builtin_amdgcn_global_load_lds((gptr)(a_src0 + kc * BK), (lptr)(dst + (2 * wave) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr)(a_src1 + kc * BK), (lptr)(dst + (2 * wave + 1) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr)(b_src + (long long)kc * (2 * HALF_B)), (lptr)(dst + OFF_B + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr)(b_src + (long long)kc * (2 * HALF_B) + HALF_B), (lptr)(dst + OFF_B + PART_B + wave * 1024), 16, 0, 0);
    };
    issue(0);
    const float s_a = afi_f16_scale(ab.amax[(long long)plane * ab.stride] * ab.cmul[plane]);
    const float inv_a = afi_pow2_inverse(s_a), inv_b = afi_pow2_inverse(((const float*)b_hdr)[plane]);
    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    int fa_off[MI], fb_off[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int row = (wm * MI + mi) * 16 + l15;
        fa_off[mi] = APRE ? row * 128 + ((lq ^ ((row >> 1) & 7)) << 4) : row * 128 + (((2 * lq) ^ ((row >> 1) & 5)) << 4);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) { const int row = (wn * NI + ni) * 16 + l15; fb_off[ni] = OFF_B + row * 64 + (((lq ^ (-(row >> 2))) & 3) << 4); }
    auto mfma = [](f16x8 x, f16x8 y, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c, 0, 0, 0); };

    for (int kc = 0; kc < nK; ++kc) {
        // this wave's share of stage k has landed (the wait), every wave's has, and every wave has left stage k - 1's buffer (the barrier).
        // Nothing but a wave's own vmcnt orders a ds_read behind an LDS-DMA.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kc + 1 < nK) issue(kc + 1);
        const unsigned char* sm = smem_b + (kc & 1) * STAGE;
        f16x8 ah[MI], al[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            if (APRE) {
                ah[mi] = *(const f16x8*)(sm + fa_off[mi]);
                al[mi] = *(const f16x8*)(sm + (fa_off[mi] ^ 64));
                continue;
            }
            const f32x4 c0 = *(const f32x4*)(sm + fa_off[mi]);
            const f32x4 c1 = *(const f32x4*)(sm + (fa_off[mi] ^ 16));
            u32x4 h, l;
            afi_split2_f16_x8_cvt(c0, c1, s_a, h, l);
            ah[mi] = __builtin_bit_cast(f16x8, h);
            al[mi] = __builtin_bit_cast(f16x8, l);
        }
        // B fragments one column group ahead: the reads of group ni + 1 are in flight under the six MFMAs of group ni (all sixteen waves
        // leave the barrier together, so nothing else covers an LDS round trip in front of every group)
        f16x8 bh = *(const f16x8*)(sm + fb_off[0]), bl = *(const f16x8*)(sm + PART_B + fb_off[0]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            f16x8 nh = bh, nl = bl;
            if (ni + 1 < NI) { nh = *(const f16x8*)(sm + fb_off[ni + 1]); nl = *(const f16x8*)(sm + PART_B + fb_off[ni + 1]); }
            __builtin_amdgcn_sched_barrier(0);               // (the reads are ISSUED here; left alone hipcc sinks them to just in front of their use)
            if (LOCAL) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                f32x4 t[MI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) t[mi] = mfma(al[mi], bh, z);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) t[mi] = mfma(ah[mi], bl, t[mi]);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) t[mi] = mfma(ah[mi], bh, t[mi]);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[mi][ni] += t[mi];
            } else {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = mfma(al[mi], bh, acc[mi][ni]);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = mfma(ah[mi], bl, acc[mi][ni]);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = mfma(ah[mi], bh, acc[mi][ni]);
            }
            __builtin_amdgcn_sched_barrier(0);
            bh = nh; bl = nl;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (this wave's reads of the stage have returned before it can reach the next barrier)
    }
    __builtin_amdgcn_s_barrier();                            // every wave has left the stage buffers: they become the C staging area
    // epilogue: a wave stages 16 of its rows x its 128 columns at a time in its own LDS strip and stores them as float4 rows
    constexpr int LDC = 128 + 4;
    float* Cs = (float*)smem_b + wave * 16 * LDC;
    float* c_base = p.C + row0 * p.N + n0 + wn * 128;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(lq * 4 + r) * LDC + ni * 16 + l15] = (acc[mi][ni][r] * inv_a) * inv_b;
        const int rbase = (wm * MI + mi) * 16;
        if (rbase < valid) {                                 // (valid is 128 or 256: whole 16-row groups)
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int item = it * 64 + lane, rloc = item >> 5, c4 = item & 31;
                __builtin_nontemporal_store(*(const f32x4*)(Cs + rloc * LDC + 4 * c4), (f32x4*)(c_base + (long long)(rbase + rloc) * p.N + 4 * c4));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// TN (weight-gradient) GEMM  dU[g][m][n] += sum_k Q[g][k][m] V[g][k][n]: the in-wave pipeline of afi_gemm_tn_bf16x6_pipe_kernel (half stages
// of 16 k rows, two LDS buffers, one barrier per half stage, a wave splits half stage h + 1 and requests h + 3 beside its own MFMAs of
// h; [16 k][128 columns] 16-bit images read transposed with ds_read_b64_tr_b16) with two fp16 pieces per operand: 16 KB per buffer
// ([Q hi | Q lo | V hi | V lo]), 12 MFMAs (32x32x16) per wave and half stage instead of 24.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void afi_gemm_tn_f16x3_kernel(const AfiGemmTN p, const AfiF16Bound qb, const AfiF16Bound vb, int ntile_m, int ntile_n, int kper) {
    constexpr int BM = 128, BN = 128, HK = 16, WN = 2, MI = 2, NI = 2;
    constexpr int PART = HK * BM * 2;                        // 4 KB: [16 k][128 columns] fp16
    constexpr int BUF = 4 * PART;
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
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    prefetch(S0(), true);                                    // half stage 0
    prefetch(S1(), 1 < nH);                                  // half stage 1
    const float s_q = afi_f16_scale(qb.amax[(long long)plane * qb.stride] * qb.cmul[plane]);
    const float s_v = afi_f16_scale(vb.amax[(long long)plane * vb.stride] * vb.cmul[plane]);
    const float inv_q = afi_pow2_inverse(s_q), inv_v = afi_pow2_inverse(s_v);
    const int st_off = 256 * kr + 16 * (c8 ^ (((kr & 3) << 2) | ((kr >> 2) & 3)));
    auto split_store = [&](auto SET, unsigned char* buf) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int op = 0; op < 2; ++op) {                     // Q, then V
            const float s = op ? s_v : s_q;
            const f32x4 v0 = op ? b_reg[S][0] : a_reg[S][0], v1 = op ? b_reg[S][1] : a_reg[S][1];
            unsigned char* base = buf + op * 2 * PART + st_off;
            u32x4 h, l;
            afi_split2_f16_x8_cvt(v0, v1, s, h, l);
            *(u32x4*)base = h; *(u32x4*)(base + PART) = l;
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
            for (int ni = 0; ni < NI; ++ni) fb_off[ni][rd] = 2 * PART + 256 * r + 16 * ((4 * (wn * NI + ni) + 2 * (g & 1) + (pp >> 1)) ^ swz) + 8 * (pp & 1);
        }
    }
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    auto mm = [](f16x8 x, f16x8 y, f32x16 c) -> f32x16 { return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0); };

    split_store(S0(), smem_b);
    prefetch(S0(), 2 < nH);                                  // half stage 2
    __syncthreads();
    auto half_stage = [&](auto NEXT, int h) {                // NEXT: the register set of half stage h + 1
        const unsigned char* cur = smem_b + (h & 1) * BUF;
        unsigned char* nxt = smem_b + ((h + 1) & 1) * BUF;
        f16x8 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            ah[mi] = afi_tr_frag_f16(cur, fa_off[mi][0], fa_off[mi][1]);
            al[mi] = afi_tr_frag_f16(cur + PART, fa_off[mi][0], fa_off[mi][1]);
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            bh[ni] = afi_tr_frag_f16(cur, fb_off[ni][0], fb_off[ni][1]);
            bl[ni] = afi_tr_frag_f16(cur + PART, fb_off[ni][0], fb_off[ni][1]);
        }
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
            for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = mm(ah[mi], bh[ni], acc[mi][ni]);
        if (h + 1 < nH) split_store(NEXT, nxt);              // (uniform) every wave left that buffer at the last barrier
        prefetch(NEXT, h + 3 < nH);
        // the scheduler's pipeline hint: one MFMA, then a share of the split's vector work
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
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
                    __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(out + (long long)row * ldn + n0 + (wn * NI + ni) * 32 + lr),
                                                            (acc[mi][ni][r] * inv_q) * inv_v);
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
                    out[(long long)row * ldn + n0 + (wn * NI + ni) * 32 + lr] = old[ni][r] + (acc[mi][ni][r] * inv_q) * inv_v;
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The weight-gradient GEMM on operands that arrive ALREADY SPLIT (winograd.hip, afi_store_split4: a row of a plane is
// [32-channel block][hi: 32 x fp16 | lo: 32 x fp16]): nothing is converted or staged through registers -- four LDS-DMA instructions per
// wave and half stage copy the [16 k][128 columns] fp16 images of Q hi, Q lo, V hi, V lo (wave w copies image w) with the transposed-read
// swizzle on the SOURCE address, AFI_TN_RING 16 KB buffers go round (half stage h + RING - 1 is requested behind the barrier of h, which also says
// that every wave has left h - 1), fragments by ds_read_b64_tr_b16, 12 MFMAs (32x32x16) per wave and half stage.  One barrier per half stage.
// ------------------------------------------------------------------------------------------------
#ifndef AFI_TN_RING
// LDS buffers of the ring = half stages in flight + 1.  Round 6 A/B (same box, stage-1 step, three alternating rounds): 4 buffers / two blocks per CU
// 72.6-73.9 against 71.3-72.4 ms for 3 buffers / three blocks (slower: occupancy matters more than prefetch depth); 2 buffers / FOUR blocks per CU
// (128 registers, 7 of them spilled outside the loop) 72.6-73.3 against 73.4-73.9, one-stream 82.0-82.6 against 82.9: kept.
#define AFI_TN_RING 2
#endif
#ifndef AFI_TN_WAVES
// waves per 128 x 128 block: 4 (64 x 64 per wave) or, A/B, 8 (32 x 64 per wave: half the accumulators, six waves per SIMD at 80 registers with 34
// spilled): round 6, same box, 72.2-73.2 against 72.2-72.5 ms per step, one-stream 82.1-82.7 against 82.0-82.7: no gain, 4 kept
#define AFI_TN_WAVES 4
#endif
#ifndef AFI_TN_MINW
#define AFI_TN_MINW (AFI_TN_WAVES == 8 ? 6 : (AFI_TN_RING == 3 ? 3 : (AFI_TN_RING == 2 ? 4 : 2)))
#endif
__global__ __launch_bounds__(64 * AFI_TN_WAVES, AFI_TN_MINW) void afi_gemm_tn_f16x3_pre_kernel(const AfiGemmTN p, const AfiF16Bound qb, const AfiF16Bound vb, int ntile_m, int ntile_n, int kper) {
    constexpr int BM = 128, BN = 128, HK = 16, WN = 2, MI = AFI_TN_WAVES == 8 ? 1 : 2, NI = 2;
    constexpr int IPW = 16 / AFI_TN_WAVES;                   // DMA instructions per wave and half stage (16 in all: four 4 KB images of four 1 KB instructions)
    constexpr int PART = HK * BM * 2;                        // 4 KB: [16 k][128 columns] fp16
    constexpr int BUF = 4 * PART;                            // [Q hi | Q lo | V hi | V lo]
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
    const int nH = (int)((k_end - k_begin) / HK);
    typedef const __attribute__((address_space(1))) void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
    // wave w copies image w: operand w >> 1 (Q, V), piece w & 1 (hi, lo).  DMA instruction i fills k rows 4 i .. 4 i + 3 of the image
    // (lane >> 4 = row, lane & 15 = physical 16-byte chunk); the image's chunk ch of row r sits at ch ^ (((r & 3) << 2) | ((r >> 2) & 3)),
    // so the lane fetches logical chunk (lane & 15) ^ ((lane >> 4) << 2 | i): columns 8 ch .. 8 ch + 7 of the tile = 32-channel block ch >> 2,
    // 16 bytes (ch & 3) of its hi or lo half
    const int img = AFI_TN_WAVES == 8 ? wave >> 1 : wave, i0 = AFI_TN_WAVES == 8 ? 2 * (wave & 1) : 0;     // (eight waves: two per image, instructions i0, i0 + 1)
    const int op = img >> 1, piece = img & 1;
    const long long ld = op ? p.N : p.M;                     // floats (= 4-byte units) per row of the operand
    const unsigned char* src_row = (const unsigned char*)((op ? p.V : p.Q) + ((long long)plane * p.rows_per_plane + k_begin + (lane >> 4)) * ld) +
                                   (long long)((op ? n0 : m0) >> 5) * 128 + piece * 64;
    int ch_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int ch = (lane & 15) ^ (((lane >> 4) << 2) | i); ch_off[i] = (ch >> 2) * 128 + (ch & 3) * 16; }
    const long long row_bytes = ld * 4;
    auto issue = [&](int h) {
        unsigned char* dst = smem_b + (h % AFI_TN_RING) * BUF + img * PART;
        const unsigned char* s0 = src_row + (long long)h * HK * row_bytes;
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const int i = i0 + j;
            __builtin_amdgcn_global_load_lds((gptr)(s0 + (long long)(4 * i) * row_bytes + ch_off[i]), (lptr)(dst + i * 1024), 16, 0, 0);
        }
    };
    issue(0);
    if (AFI_TN_RING > 2 && 1 < nH) issue(1);
    if (AFI_TN_RING > 3 && 2 < nH) issue(2);
    const float s_q = afi_f16_scale(qb.amax[(long long)plane * qb.stride] * qb.cmul[plane]);
    const float s_v = afi_f16_scale(vb.amax[(long long)plane * vb.stride] * vb.cmul[plane]);
    const float inv_q = afi_pow2_inverse(s_q), inv_v = afi_pow2_inverse(s_v);
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
            for (int ni = 0; ni < NI; ++ni) fb_off[ni][rd] = 2 * PART + 256 * r + 16 * ((4 * (wn * NI + ni) + 2 * (g & 1) + (pp >> 1)) ^ swz) + 8 * (pp & 1);
        }
    }
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    auto mm = [](f16x8 x, f16x8 y, f32x16 c) -> f32x16 { return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0); };

    for (int h = 0; h < nH; ++h) {
        // requests complete in order, four per half stage: with h + 1 requested too, vmcnt(4) = half stage h has landed
        if (AFI_TN_RING > 3 && h + 2 < nH) { if (IPW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else if (AFI_TN_RING > 2 && h + 1 < nH) { if (IPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // ... for every wave's image; and every wave has left half stage h - 1
        if (h + AFI_TN_RING - 1 < nH) issue(h + AFI_TN_RING - 1);
        const unsigned char* cur = smem_b + (h % AFI_TN_RING) * BUF;
        f16x8 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            ah[mi] = afi_tr_frag_f16(cur, fa_off[mi][0], fa_off[mi][1]);
            al[mi] = afi_tr_frag_f16(cur + PART, fa_off[mi][0], fa_off[mi][1]);
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            bh[ni] = afi_tr_frag_f16(cur, fb_off[ni][0], fb_off[ni][1]);
            bl[ni] = afi_tr_frag_f16(cur + PART, fb_off[ni][0], fb_off[ni][1]);
        }
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
            for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = mm(ah[mi], bh[ni], acc[mi][ni]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the wave's fragment reads have returned before it reaches the next barrier)
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
                    __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(out + (long long)row * ldn + n0 + (wn * NI + ni) * 32 + lr),
                                                            (acc[mi][ni][r] * inv_q) * inv_v);
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
                    out[(long long)row * ldn + n0 + (wn * NI + ni) * 32 + lr] = old[ni][r] + (acc[mi][ni][r] * inv_q) * inv_v;
                }
        }
    }
}
