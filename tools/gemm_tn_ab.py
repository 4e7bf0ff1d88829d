"""Timing of the Winograd weight-gradient (TN) GEMM on the step's shapes, random operands, with a correctness line against fp64 first.
The A/B runs recorded in profiles/r03/gemm_tn_variants_ab.log used this tool at commit a42e297, where the three alternative kernels
(fp32 LDS-DMA + in-register split, 16x16x32 MFMA, pre-split image operands) still exist behind a debug switch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afigan_amd import ops  # noqa: E402

SHAPES = [(36, 8448, 256, 256), (36, 8448, 512, 256), (36, 8448, 1024, 512), (36, 2176, 256, 256), (36, 2176, 512, 256), (36, 2176, 1024, 512),
          (16, 640, 1024, 512)]


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    Q = torch.randn(2, 512, 256, device=dev, generator=g)
    V = torch.randn(2, 512, 256, device=dev, generator=g)
    ref = torch.einsum("gkm,gkn->gmn", Q.double(), V.double())
    for dtype in ("fp32", "bf16x6", "bf16x3", "bf16"):
        out = ops.gemm_tn(Q, V, dtype)
        print(f"{dtype}: rel err vs fp64 {((out.double() - ref).norm() / ref.norm()).item():.3e}", flush=True)
    if os.environ.get("AB_PMC"):                       # counter pass (tools/micro/tn_ab_pmc.sh): the large shape only, three launches
        planes, T, M, N = 36, 8448, 1024, 512
        Q = torch.randn(planes, T, M, device=dev, generator=g)
        V = torch.randn(planes, T, N, device=dev, generator=g)
        out = torch.zeros(planes, M, N, device=dev)
        for _ in range(3):
            ops.gemm_tn(Q, V, "bf16x6", out=out)
        A = torch.randn(planes, T, N, device=dev, generator=g)
        B = torch.randn(planes, M, N, device=dev, generator=g)
        for _ in range(3):
            ops.gemm_nt(A, B, "bf16x6")
        torch.cuda.synchronize()
        return
    for planes, T, M, N in SHAPES:
        Q = torch.randn(planes, T, M, device=dev, generator=g)
        V = torch.randn(planes, T, N, device=dev, generator=g)
        out = torch.zeros(planes, M, N, device=dev)
        line = f"planes {planes:2d} T {T:5d} M {M:4d} N {N:4d}:"
        for dtype in ("bf16x6", "bf16x3"):
            for _ in range(3):
                ops.gemm_tn(Q, V, dtype, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm_tn(Q, V, dtype, out=out)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            line += f"  {dtype} {us:7.1f} us {2.0 * planes * T * M * N / us / 1e6:6.1f} TF"
        print(line, flush=True)


if __name__ == "__main__":
    main()
