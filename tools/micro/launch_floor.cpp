// Calibration: what does one dependent kernel cost on this box?  (empty kernels, small / large kernargs, eager and hipGraph)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Big { char b[3400]; };
struct Mid { char b[460]; };
__global__ void k_empty(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }
__global__ void k_mid(Mid m, float* p) { if (p && threadIdx.x == 9999) p[0] = m.b[3]; }
__global__ void k_big(Big m, float* p) { if (p && threadIdx.x == 9999) p[0] = m.b[3]; }
__global__ void k_touch(float* p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] += 1.f; }
__global__ void k_div(long long* p, long long a, long long b, int n) {
    long long i = blockIdx.x * 256 + threadIdx.x; long long s = 0;
    for (int k = 0; k < n; ++k) s += (i * 7919 + a + k) / (b + k);
    if (s == 42) p[0] = s;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <class F> float timeit(hipStream_t st, int iters, F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) f();
    hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1e3f / iters;
}
int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    float* d; CK(hipMalloc(&d, 64 << 20));
    long long* dl; CK(hipMalloc(&dl, 1024));
    Big big{}; Mid mid{};
    for (int grid : {1, 27, 256, 512, 2048}) {
        printf("grid %5d: empty %.2f us  mid-arg %.2f us  big-arg %.2f us  touch(4MB) %.2f us  div8 %.2f  div32 %.2f\n", grid,
               timeit(st, 2000, [&] { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, st, d); }),
               timeit(st, 2000, [&] { hipLaunchKernelGGL(k_mid, dim3(grid), dim3(256), 0, st, mid, d); }),
               timeit(st, 2000, [&] { hipLaunchKernelGGL(k_big, dim3(grid), dim3(256), 0, st, big, d); }),
               timeit(st, 2000, [&] { hipLaunchKernelGGL(k_touch, dim3(4096), dim3(256), 0, st, d, 1 << 20); }),
               timeit(st, 2000, [&] { hipLaunchKernelGGL(k_div, dim3(grid), dim3(256), 0, st, dl, 12345LL, 850LL, 8); }),
               timeit(st, 2000, [&] { hipLaunchKernelGGL(k_div, dim3(grid), dim3(256), 0, st, dl, 12345LL, 850LL, 32); }));
    }
    // graph of 50 dependent empty / mid kernels
    for (int which = 0; which < 3; ++which) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 50; ++i) {
            if (which == 0) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, d);
            else if (which == 1) hipLaunchKernelGGL(k_mid, dim3(256), dim3(256), 0, st, mid, d);
            else hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, st, big, d);
        }
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float us = timeit(st, 200, [&] { hipGraphLaunch(ge, st); });
        printf("graph of 50 %s kernels: %.1f us per replay = %.2f us per kernel\n", which == 0 ? "empty" : which == 1 ? "mid-arg" : "big-arg", us, us / 50);
    }
    return 0;
}
