// Micro-benchmark of the dense block's fused chain kernel (csrc/smallmap.hip: afi_rdb_chain6_kernel) at config-1 size, forward form:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DAFI_CH6_ABLATE=n] tools/micro/ch6_bench.cpp -o tools/micro/ch6_bench[_n]
// AFI_CH6_ABLATE removes parts (1 every stage, 2 weight-fragment loads, 4 MFMAs; results are wrong then).
#include "../../afigan_amd/csrc/smallmap.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 25, W = argc > 2 ? atoi(argv[2]) : 34, C = 256, G = 32, L = 384, P = H * W;
    float *b, *s, *w;
    CK(hipMalloc(&b, (size_t)P * L * 4)); CK(hipMalloc(&s, (size_t)P * C * 4));
    std::vector<float> h((size_t)P * L);
    srand(3); for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(b, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(s, h.data(), (size_t)P * C * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    AfiChain6 c; memset(&c, 0, sizeof(c));
    c.N = 1; c.H = H; c.W = W; c.a_sgn = 1; c.mode = 0;
    auto view = [&](float* p, int ld) { return AfiView{p, (long long)H * W * ld, (long long)W * ld, ld}; };
    c.src0 = view(b + C, L);
    unsigned char* imgs[3];
    for (int k = 2; k <= 4; ++k) {                          // growth conv k: [G][9][cin]
        const int cin = C + (k - 1) * G;
        CK(hipMalloc(&w, (size_t)G * 9 * cin * 4));
        std::vector<float> hw((size_t)G * 9 * cin); for (auto& v : hw) v = ((float)rand() / RAND_MAX - 0.5f) * 0.05f;
        CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&imgs[k - 2], afi_wk6_image_bytes(G, cin, 9, 1)));
        AfiWk6ImgJob job{w, 9LL * cin, cin, G, cin, 9, 1, 0, 0, imgs[k - 2], 0, 0};
        if (afi_launch_wk6_images(&job, 1, st) != AFI_OK) { printf("image build failed\n"); return 1; }
    }
    for (int ph = 0; ph < 3; ++ph) {
        for (int ci = 0; ci <= ph; ++ci) { c.ph[ph].img[ci] = imgs[ph]; c.ph[ph].stage0[ci] = (C / 32 + ci) * 9; }
        c.ph[ph].partial = view(s + ph * G, C);
        c.ph[ph].out = view(b + C + (ph + 1) * G, L);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int rc = afi_launch_rdb_chain6(c, st);
    if (rc != AFI_OK) { printf("launch rc %d\n", rc); return 1; }
    for (int i = 0; i < 5; ++i) afi_launch_rdb_chain6(c, st);
    (void)hipEventRecord(e0, st);
    const int iters = 50;
    for (int i = 0; i < iters; ++i) afi_launch_rdb_chain6(c, st);
    (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("ch6 ablate %d  %dx%d | %.1f us per launch\n", AFI_CH6_ABLATE, H, W, ms * 1e3 / iters);
    return 0;
}
