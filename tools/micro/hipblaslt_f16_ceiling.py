"""What the library GEMM (torch.matmul -> hipBLASLt) reaches on this chip in plain fp16 at the shapes of the step's big Winograd GEMMs, with random and
with all-zero operands: the practical f16-MFMA ceiling under the chip's power limit, to set the f16x3 kernels' executed rate against.
Usage: python tools/micro/hipblaslt_f16_ceiling.py"""
import time
import torch

dev = torch.device("cuda:0")


def rate(a, b, iters=20):
    for _ in range(3):
        torch.bmm(a, b)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        torch.bmm(a, b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / iters
    return 2.0 * a.shape[0] * a.shape[1] * a.shape[2] * b.shape[2] / dt / 1e12, dt * 1e6


# (one shape, fp16 only: the bf16 leg of this loop ended in a memory access fault inside the library's kernel on this image -- not run again)
for (planes, M, N, K) in ((36, 8448, 1024, 1024),):
    for dt_ in (torch.float16,):
        a = torch.randn(planes, M, K, device=dev).to(dt_)
        b = torch.randn(planes, N, K, device=dev).to(dt_).transpose(1, 2)      # NT: B stored [N][K]
        r, us = rate(a, b)
        z, zus = rate(torch.zeros_like(a), torch.zeros_like(b))
        print(f"planes {planes:3d} M {M:6d} N {N:5d} K {K:5d} {str(dt_):15s}: random {r:7.1f} TFLOP/s ({us:7.1f} us)   zeros {z:7.1f} TFLOP/s ({zus:7.1f} us)")
