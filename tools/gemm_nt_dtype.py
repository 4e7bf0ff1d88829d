"""Batched Winograd-plane NT / TN GEMMs in every arithmetic setting (fp32 / f16x3 / bf16x6 / bf16x3 / bf16): error against fp64 and TFLOP/s.
Usage: python tools/gemm_nt_dtype.py [planes rows N K] ..."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import afigan_amd as amd
from afigan_amd import _lib


def run(planes, rows, N, K, iters=20):
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn((planes, rows, K), device="cuda", generator=g)
    B = torch.randn((planes, N, K), device="cuda", generator=g) / K ** 0.5
    Cm = torch.empty((planes, rows, N), device="cuda")
    ref = torch.bmm(A[:2, :256].double(), B[:2].double().transpose(1, 2))
    st = amd.ops.stream_ptr()
    variants = list(_lib.DTYPES.items()) + [("f16x3/128", _lib.DTYPES["f16x3"])]       # the last: the 128 x 128 tile kernel on every shape
    for name, dt in variants:
        import ctypes
        _lib.load().afi_debug_set_nt_ablation(32 if name == "f16x3/128" else 0)
        nb = _lib.load().afi_gemm_nt_scratch_bytes(planes, N, K, dt)
        sc = torch.empty(max(int(nb), 16), device="cuda", dtype=torch.uint8)
        args = (C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(Cm.data_ptr()), planes, rows, N, K, dt, C.c_void_p(sc.data_ptr()), nb, st)
        Cm.zero_()
        _lib.check(_lib.load().afi_gemm_nt(*args), "afi_gemm_nt")
        torch.cuda.synchronize()
        err = ((Cm[:2, :256].double() - ref).abs().max() / ref.abs().max()).item()
        tail = ((Cm[-1, -128:].double() - A[-1, -128:].double() @ B[-1].double().t()).abs().max() / ref.abs().max()).item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            _lib.load().afi_gemm_nt(*args)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print(f"{planes}x{rows}x{N}x{K} {name:9s} max-norm err {err:.2e} (tail {tail:.2e})  {ms:.3f} ms  {2.0 * planes * rows * N * K / ms / 1e9:.1f} TFLOP/s", flush=True)


def run_tn(planes, rows, M, N, iters=20):
    g = torch.Generator(device="cuda").manual_seed(2)
    Q = torch.randn((planes, rows, M), device="cuda", generator=g)
    V = torch.randn((planes, rows, N), device="cuda", generator=g) / rows ** 0.5
    dU = torch.empty((planes, M, N), device="cuda")
    ref = torch.bmm(Q[:2].double().transpose(1, 2), V[:2].double())
    st = amd.ops.stream_ptr()
    sc = torch.empty(1024, device="cuda", dtype=torch.uint8)
    for name, dt in _lib.DTYPES.items():
        args = (C.c_void_p(Q.data_ptr()), C.c_void_p(V.data_ptr()), C.c_void_p(dU.data_ptr()), planes, rows, M, N, dt, C.c_void_p(sc.data_ptr()), sc.numel(), st)
        dU.zero_()
        _lib.check(_lib.load().afi_gemm_tn(*args), "afi_gemm_tn")
        torch.cuda.synchronize()
        err = ((dU[:2].double() - ref).abs().max() / ref.abs().max()).item()
        tail = ((dU[-1].double() - Q[-1].double().t() @ V[-1].double()).abs().max() / ref.abs().max()).item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            _lib.load().afi_gemm_tn(*args)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print(f"TN {planes}x{rows} -> {M}x{N} {name:7s} max-norm err {err:.2e} (tail {tail:.2e})  {ms:.3f} ms  {2.0 * planes * rows * M * N / ms / 1e9:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    cases = [tuple(a[i:i + 4]) for i in range(0, len(a), 4)] or [(16, 11264, 512, 256), (16, 11264, 1024, 512), (36, 2816, 1024, 1024), (36, 2816, 256, 1024), (16, 5632, 128, 288)]
    for cs in cases:
        run(*cs)
    if not a:
        for cs in [(36, 2816, 512, 256), (36, 2816, 1024, 512), (36, 2816, 1024, 1024), (16, 11264, 256, 256)]:
            run_tn(*cs)
