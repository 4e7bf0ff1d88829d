// Micro-benchmark of the small-map pixel GEMM kernels (csrc/smallmap.hip) on one conv shape, outside the library:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DAFI_WK6_ABLATE=n] tools/micro/wk6_bench.cpp -o tools/micro/wk6_bench[_n]
//   ./wk6_bench [Ncols=384] [Ck=256] [H=25] [W=34] [rc=0]
// Prints the time per launch of the bf16x6 kernel (afi_pix_gemm_wk6, weights pre-split into an image) and of the fp32-MFMA kernel
// (afi_pix_gemm_wk) on the same problem, and the largest difference between their outputs.  AFI_WK6_ABLATE removes parts of the bf16x6
// kernel's stage (1 MFMAs, 2 A gather, 4 B loads, 8 split; results are wrong then) to see what the stage time is made of.
#include "../../afigan_amd/csrc/smallmap.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int Ncols = argc > 1 ? atoi(argv[1]) : 384, Ck = argc > 2 ? atoi(argv[2]) : 256, H = argc > 3 ? atoi(argv[3]) : 25, W = argc > 4 ? atoi(argv[4]) : 34;
    const int rc = argc > 5 ? atoi(argv[5]) : 0;
    const int N = 1, P = N * H * W;
    // forward: A [P][Ck], weights [Ncols][9][Ck]; data gradient (rc): A [P][Ck] (Ck = Cout), weights [Ck][9][Ncols]
    const long long wn = (long long)Ncols * 9 * Ck;
    std::vector<float> ha((size_t)P * Ck), hw((size_t)wn);
    srand(1);
    for (auto& v : ha) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto& v : hw) v = ((float)rand() / RAND_MAX - 0.5f) * 0.05f;
    float *da, *dw, *o6, *o32; unsigned char* img;
    CK(hipMalloc(&da, ha.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4));
    CK(hipMalloc(&o6, (size_t)P * Ncols * 4)); CK(hipMalloc(&o32, (size_t)P * Ncols * 4));
    const long long ib = afi_wk6_image_bytes(Ncols, Ck, 9, 1);
    CK(hipMalloc(&img, ib));
    CK(hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    AfiPixGemm g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.H = H; g.W = W; g.ntaps = 9; g.nKphase = 1; g.a_sgn = rc ? -1 : 1; g.a_up = 1; g.o_up = 1;
    g.alpha = 1.f; g.r1s = g.r2s = 1.f; g.a_stride = 1; g.aH = H; g.aW = W; g.oH = g.oW = 1 << 30; g.post_scale = 1.f;
    g.Ck = Ck; g.Ncols = Ncols; g.CoutPhase = Ncols;
    g.A = AfiView{da, (long long)H * W * Ck, (long long)W * Ck, Ck};
    g.B = dw;
    if (!rc) { g.b_sRow = 9LL * Ck; g.b_sTap = Ck; } else { g.b_sRow = 9LL * Ncols; g.b_sTap = Ncols; }
    g.O = AfiView{o32, (long long)H * W * Ncols, (long long)W * Ncols, Ncols};
    hipStream_t st; CK(hipStreamCreate(&st));
    AfiWk6ImgJob job{dw, g.b_sRow, g.b_sTap, Ncols, Ck, 9, 1, rc, 0, img, 0, 0};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto&& fn, int iters) { for (int i = 0; i < 5; ++i) fn(); hipEventRecord(e0, st); for (int i = 0; i < iters; ++i) fn(); hipEventRecord(e1, st); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / iters; };
    const float t_img = timeit([&] { afi_launch_wk6_images(&job, 1, st); }, 20);
    int rcl = rc ? launch_wk<true>(g, st) : launch_wk<false>(g, st);
    if (rcl != AFI_OK) { printf("fp32 launch rc %d\n", rcl); return 1; }
    const float t32 = timeit([&] { if (rc) launch_wk<true>(g, st); else launch_wk<false>(g, st); }, 50);
    AfiPixGemm g6 = g;
    g6.O.p = o6; g6.Bimg = img; g6.bimg_stage0 = 0; g6.bimg_nstages = afi_cdiv(Ck, 32) * 9;
    rcl = launch_wk<false>(g6, st);
    if (rcl != AFI_OK) { printf("wk6 launch rc %d\n", rcl); return 1; }
    const float t6 = timeit([&] { launch_wk<false>(g6, st); }, 50);
    CK(hipStreamSynchronize(st));
    std::vector<float> h6((size_t)P * Ncols), h32((size_t)P * Ncols);
    CK(hipMemcpy(h6.data(), o6, h6.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h32.data(), o32, h32.size() * 4, hipMemcpyDeviceToHost));
    double md = 0, mx = 0;
    for (size_t i = 0; i < h6.size(); ++i) { md = fmax(md, fabs((double)h6[i] - h32[i])); mx = fmax(mx, fabs((double)h32[i])); }
    // in-kernel stamps of the 32 x 32 form (DIAG build of the kernel: s_memtime at the phase boundaries, s_memrealtime at both ends)
    {
        AfiWkArgs wk;
        if (wk_prepare(g6, false, wk) == AFI_OK && wk6_ok(g6, wk)) {
            const int G = wk.ntile_m * wk.ntile_n;
            unsigned long long* dbg;
            CK(hipMalloc(&dbg, (size_t)G * 10 * 8)); CK(hipMemset(dbg, 0, (size_t)G * 10 * 8));
            wk.dbg = dbg;
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((afi_pix_gemm_wk6_kernel<true>), dim3((unsigned)G), dim3(512), sizeof(float) * 8 * 32 * (AFI_BK + 4), st, g6, wk);
            CK(hipStreamSynchronize(st));
            std::vector<unsigned long long> hd((size_t)G * 10);
            CK(hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost));
            double ph[6] = {0, 0, 0, 0, 0, 0}, span = 0; unsigned long long t0min = ~0ull, t1max = 0;
            for (int b = 0; b < G; ++b) {
                const unsigned long long* d = &hd[(size_t)b * 10];
                const int idx[7] = {0, 1, 2, 3, 4, 5, 8};
                for (int i = 0; i < 6; ++i) ph[i] += (double)(d[idx[i + 1]] - d[idx[i]]);
                span += (double)(d[7] - d[6]);
                if (d[6] < t0min) t0min = d[6];
                if (d[7] > t1max) t1max = d[7];
            }
            printf("stamps (32 x 32 form, %d blocks, mean per block, shader cycles): setup %.0f | first gather issued %.0f | prologue (first stage staged, second requested, first weights requested) %.0f | K loop %.0f | reduce + epilogue %.0f | tail %.0f || block span %.2f us, first start to last end %.2f us\n",
                   G, ph[0] / G, ph[1] / G, ph[2] / G, ph[3] / G, ph[4] / G, ph[5] / G, span / G * 0.01, (double)(t1max - t0min) * 0.01);
        }
    }
    const double fl = 2.0 * P * Ncols * 9.0 * Ck;
    printf("ablate %d  M %d N %d K %d rc %d | image build %.1f us | fp32 wk %.1f us (%.1f TF/s) | wk6 %.1f us (%.1f TF/s) | max diff %.3g of %.3g\n", AFI_WK6_ABLATE, P, Ncols,
           9 * Ck, rc, t_img, t32, fl / t32 * 1e-6, t6, fl / t6 * 1e-6, md, mx);
    return 0;
}
