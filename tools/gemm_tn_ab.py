"""A/B of the two Winograd weight-gradient (TN) GEMM kernels in ONE process on the step's shapes: the register-staged kernel of round 2
against the LDS-DMA kernel of round 3 (afi_gemm_bf16.h).  Random operands; a correctness line against fp64 first."""
import ctypes
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afigan_amd import _lib, ops  # noqa: E402

SHAPES = [(36, 8448, 256, 256), (36, 8448, 512, 256), (36, 8448, 1024, 512), (36, 2176, 256, 256), (36, 2176, 512, 256), (36, 2176, 1024, 512),
          (16, 640, 1024, 512)]


SPLITS = {"bf16x6": 6, "bf16x3": 3, "bf16": 1}


def pre_images(lib, X, dtype):
    planes, rows, C = X.shape
    npart = {6: 3, 3: 2, 1: 1}[SPLITS[dtype]]
    img = torch.empty(planes * rows * C * 2 * npart, device=X.device, dtype=torch.uint8)
    _lib.check(lib.afi_debug_tn_split(ops._p(X), ops._p(img), planes, rows, C, SPLITS[dtype], ops.stream_ptr()), "split")
    return img


def gemm_pre(lib, Qi, Vi, out, planes, rows, M, N, dtype):
    _lib.check(lib.afi_debug_gemm_tn_pre(ops._p(Qi), ops._p(Vi), ops._p(out), planes, rows, M, N, SPLITS[dtype], ops.stream_ptr()), "gemm_tn_pre")


def main():
    lib = _lib.load()
    setv = lib.afi_debug_set_tn_variant
    setv.argtypes, setv.restype = [ctypes.c_int], None
    vp, ll, ci = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int
    lib.afi_debug_tn_split.argtypes, lib.afi_debug_tn_split.restype = [vp, vp, ci, ll, ci, ci, vp], ci
    lib.afi_debug_gemm_tn_pre.argtypes, lib.afi_debug_gemm_tn_pre.restype = [vp, vp, vp, ci, ll, ci, ci, ci, vp], ci
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    for dtype in ("bf16x6",):
        Q = torch.randn(2, 512, 256, device=dev, generator=g)
        V = torch.randn(2, 512, 256, device=dev, generator=g)
        ref = torch.einsum("gkm,gkn->gmn", Q.double(), V.double())
        for v in (0, 1, 2):
            setv(v)
            out = ops.gemm_tn(Q, V, dtype)
            print(f"{dtype} variant {v}: rel err vs fp64 {((out.double() - ref).norm() / ref.norm()).item():.3e}", flush=True)
        out = torch.zeros(2, 256, 256, device=dev)
        setv(10)
        gemm_pre(lib, pre_images(lib, Q, dtype), pre_images(lib, V, dtype), out, 2, 512, 256, 256, dtype)
        print(f"{dtype} pre-split: rel err vs fp64 {((out.double() - ref).norm() / ref.norm()).item():.3e}", flush=True)
    import os
    if os.environ.get("AB_PMC"):                       # counter pass: the large shape only, three launches of every kernel
        planes, T, M, N = 36, 8448, 1024, 512
        Q = torch.randn(planes, T, M, device=dev, generator=g)
        V = torch.randn(planes, T, N, device=dev, generator=g)
        out = torch.zeros(planes, M, N, device=dev)
        for v in (0, 1, 2):
            setv(v)
            for _ in range(3):
                ops.gemm_tn(Q, V, "bf16x6", out=out)
        Qi, Vi = pre_images(lib, Q, "bf16x6"), pre_images(lib, V, "bf16x6")
        for _ in range(3):
            gemm_pre(lib, Qi, Vi, out, planes, T, M, N, "bf16x6")
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
        dtype = "bf16x6"
        Qi, Vi = pre_images(lib, Q, dtype), pre_images(lib, V, dtype)
        ref = None
        for v in (3, 10, 11, 12):
            setv(v)
            run = (lambda: ops.gemm_tn(Q, V, dtype, out=out)) if v < 10 else (lambda: gemm_pre(lib, Qi, Vi, out, planes, T, M, N, dtype))
            out.zero_(); run()
            if ref is None:
                ref = out.clone()
            err = ((out - ref).norm() / ref.norm()).item()
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            line += f"  v{v} {us:7.1f} us {2.0 * planes * T * M * N / us / 1e6:6.1f} TF (d {err:.1e})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
