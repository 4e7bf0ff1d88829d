// Micro-benchmark of the grouped small-map weight-gradient kernel on the bf16 matrix cores (csrc/smallmap.hip: afi_wgrad6_group_sk_kernel)
// on the config-1 problem set (seven 3x3 weight gradients of the interpolator at 25 x 34 / 50 x 68 + three packed growth-conv problems):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DAFI_WG6_DIAG] [-DAFI_WG6_SPLIT=afi_split3_pair] [-DAFI_WG6_VALU_PER_MFMA=n] tools/micro/wg6_bench.cpp -o tools/micro/wg6_bench[_x]
// AFI_WG6_DIAG: in-kernel stamps (s_memrealtime) -- per block the time in range prologues, K loop and epilogues.  (The stage ablations of
// profiles/r04/wgrad6_ablation_*.txt were taken on the kernel's first, block-phased version, removed since.)
static int g_upb = 0;
#define AFI_WG6_UPB_OVERRIDE g_upb                          // (argv[3]: run length of the stream-K cut, 0 = the launcher's own rule)
#include "../../afigan_amd/csrc/smallmap.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
static AfiWgradGemm prob(float* dy, float* x, float* dw, int H, int W, int Cout, int Cin, int ld_dy, int ld_x) {
    AfiWgradGemm g; memset(&g, 0, sizeof(g));
    g.N = 1; g.H = H; g.W = W; g.ntaps = 9; g.Mrows = Cout; g.Ncols = Cin;
    g.DY = AfiView{dy, (long long)H * W * ld_dy, (long long)W * ld_dy, ld_dy}; g.dy_up = 1; g.CoutPhase = Cout;
    g.X = AfiView{x, (long long)H * W * ld_x, (long long)W * ld_x, ld_x}; g.x_stride = 1; g.xH = H; g.xW = W;
    g.DW = dw; g.dw_sRow = 9LL * Cin; g.dw_sTap = Cin; g.alpha = 1.f;
    return g;
}
int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 25, W = argc > 2 ? atoi(argv[2]) : 34;
    const int C = 256, G = 32, L = 384, P = H * W;
    g_upb = argc > 3 ? atoi(argv[3]) : 0;
    float *act, *grad, *dw;
    CK(hipMalloc(&act, (size_t)4 * P * 1024 * 4)); CK(hipMalloc(&grad, (size_t)4 * P * 1024 * 4)); CK(hipMalloc(&dw, (size_t)1024 * 9 * 384 * 4 * 2));
    std::vector<float> h((size_t)4 * P * 1024);
    srand(2); for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(act, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(grad, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dw, 0, (size_t)1024 * 9 * 384 * 4 * 2));
    std::vector<AfiWgradGemm> pr;
    pr.push_back(prob(grad, act, dw, 2 * H, 2 * W, C, C, C, C));             // final conv on the up-sampled map
    { AfiWgradGemm g = prob(grad, act, dw, H, W, 4 * C, C, C, C); g.dy_up = 2; g.CoutPhase = C; g.DY = AfiView{grad, (long long)4 * P * C, (long long)2 * W * C, C}; pr.push_back(g); }   // conv-transpose
    pr.push_back(prob(grad, act, dw, H, W, C, C, C, C));                     // trunk
    for (int r = 0; r < 3; ++r) { pr.push_back(prob(grad, act, dw, H, W, C, L, L, L)); pr.push_back(prob(grad + C, act, dw, H, W, 4 * G, L, L, L)); }   // conv5 + packed growth convs
    pr.push_back(prob(grad, act, dw, H, W, C, C, L, C));                     // head
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double fl = 0; for (auto& g : pr) fl += 2.0 * g.N * g.H * g.W * g.Mrows * g.Ncols * 9;
    {
        int rc = afi_launch_wgrad6_group(pr.data(), (int)pr.size(), st);
        if (rc != AFI_OK) { printf("launch rc %d\n", rc); return 1; }
        for (int i = 0; i < 5; ++i) afi_launch_wgrad6_group(pr.data(), (int)pr.size(), st);
        (void)hipEventRecord(e0, st);
        const int iters = 30;
        for (int i = 0; i < iters; ++i) afi_launch_wgrad6_group(pr.data(), (int)pr.size(), st);
        (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("wg6  %dx%d  upb %d  %zu problems | %.1f us per launch (%.1f TF/s)\n", H, W, g_upb, pr.size(), ms * 1e3 / iters, fl / (ms * 1e-3 / iters) * 1e-12);
    }
#ifdef AFI_WG6_DIAG
    {
        static unsigned long long hst[4096][6];
        memset(hst, 0, sizeof(hst));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(afi_wg6_stamp), hst, sizeof(hst)));
        afi_launch_wgrad6_group(pr.data(), (int)pr.size(), st);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpyFromSymbol(hst, HIP_SYMBOL(afi_wg6_stamp), sizeof(hst)));
        double s0 = 0, s1 = 0, s2 = 0, rg = 0, hs = 0; int nb = 0; double mx = 0;
        for (int b = 0; b < 4096; ++b) if (hst[b][3]) { ++nb; s0 += hst[b][0]; s1 += hst[b][1]; s2 += hst[b][2]; rg += hst[b][3]; hs += hst[b][4]; mx = fmax(mx, (double)(hst[b][0] + hst[b][1] + hst[b][2])); }
        printf("diag: %d blocks, per block: %.2f ranges, %.1f half stages | prologue %.2f us, K loop %.2f us (%.3f us per half stage), epilogue %.2f us | slowest block %.2f us\n",
               nb, rg / nb, hs / nb, s0 / nb * 0.01, s1 / nb * 0.01, s1 / hs * 0.01, s2 / nb * 0.01, mx * 0.01);
    }
#endif
    return 0;
}
