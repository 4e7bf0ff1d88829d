"""Ablations of afi_gemm_nt_f16x3_kernel (128 x 128 tile) on the step's large shapes: what the loop waits for.  Kernel-only times from the
library's own HIP-event brackets (afi_profile_*), so the stand-alone entry point's maxima passes are not in them.
Usage: python tools/micro/nt_f16_ablate.py [planes rows N K] ..."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import afigan_amd as amd
from afigan_amd import _lib

NAMES = {0: "as shipped", 1: "L2-resident operands", 2: "no DMA behind the first stage", 4: "no MFMAs", 8: "no split", 16: "split by v_fma_mix (results right)", 12: "no MFMAs, no split", 6: "no MFMAs, no DMA", 14: "no MFMAs, no DMA, no split: barriers + fragment reads", 10: "no DMA, no split", 20: "no MFMAs, mix split", 24: "?"}


def kernel_ms(lib, fn, iters=10):
    fn(); torch.cuda.synchronize()
    lib.afi_profile_enable(1)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    lib.afi_profile_enable(0)
    best = None
    for k in range(lib.afi_profile_num_kinds()):
        o = (C.c_double * 3)()
        lib.afi_profile_get(k, o)
        if o[0] > 0 and "gemm_nt" in lib.afi_profile_kind_name(k).decode():
            best = o[1] / o[0]
    return best


def main():
    lib = _lib.load()
    if not hasattr(lib, "afi_debug_set_nt_ablation"):
        raise SystemExit("this library was built without -DAFI_ABLATIONS: build an A/B copy (AFI_HIPCC_FLAGS=-DAFI_ABLATIONS, see csrc/igemm.hip) and select it "
                         "with AFI_LIB_PATH -- the product build carries neither the ablated kernels nor their process-wide switch")
    a = [int(v) for v in sys.argv[1:]]
    cases = [tuple(a[i:i + 4]) for i in range(0, len(a), 4)] or [(36, 8448, 1024, 1024), (16, 33664, 1024, 512), (36, 2176, 1024, 1024)]
    for planes, rows, N, K in cases:
        g = torch.Generator(device="cuda").manual_seed(1)
        A = torch.randn((planes, rows, K), device="cuda", generator=g)
        B = torch.randn((planes, N, K), device="cuda", generator=g) / K ** 0.5
        out = torch.empty((planes, rows, N), device="cuda")
        fl = 2.0 * planes * rows * N * K
        for tiles256, tag in ((0, "128x128"), (512, "256x256")):
            for abl in ([0, 2, 4, 8] if tiles256 == 0 else [0]):
                lib.afi_debug_set_nt_ablation(abl + (32 if tiles256 == 0 else 0))
                ms = kernel_ms(lib, lambda: amd.ops.gemm_nt(A, B, "f16x3", out=out))
                print(f"{planes}x{rows}x{N}x{K} {tag} {NAMES[abl]:55s} {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.1f} TFLOP/s", flush=True)
        lib.afi_debug_set_nt_ablation(0)
        for dt in ("bf16x6", "bf16x3"):
            ms = kernel_ms(lib, lambda: amd.ops.gemm_nt(A, B, dt, out=out))
            print(f"{planes}x{rows}x{N}x{K} {dt:63s} {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.1f} TFLOP/s", flush=True)
        # the same kernels on operands that cost the multipliers nothing: how much of the time is the clock the chip holds under random data
        for name, (A2, B2) in (("all-zero operands", (torch.zeros_like(A), torch.zeros_like(B))), ("small integers (|x| <= 2)", (torch.randint(-2, 3, A.shape, device="cuda").float(), torch.randint(-2, 3, B.shape, device="cuda").float()))):
            for tiles256, tag in ((0, "128x128"), (512, "256x256")):
                lib.afi_debug_set_nt_ablation(32 if tiles256 == 0 else 0)
                ms = kernel_ms(lib, lambda: amd.ops.gemm_nt(A2, B2, "f16x3", out=out))
                print(f"{planes}x{rows}x{N}x{K} f16x3 {tag} on {name:44s} {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.1f} TFLOP/s", flush=True)
            ms = kernel_ms(lib, lambda: amd.ops.gemm_nt(A2, B2, "bf16x6", out=out))
            print(f"{planes}x{rows}x{N}x{K} bf16x6 on {name:52s} {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.1f} TFLOP/s", flush=True)
        lib.afi_debug_set_nt_ablation(0)


if __name__ == "__main__":
    main()
