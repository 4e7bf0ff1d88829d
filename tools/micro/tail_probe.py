"""The fused discriminator tail's kernels alone (csrc/elementwise.hip, afi_launch_disc_tail_*): time per launch and bytes per second at the
step's largest size, and a check against plain tensor ops.  The launchers are C++ symbols of the library (not part of the C-ABI): looked up
by their mangled names, which is why this is a tool and not a test.

    python tools/micro/tail_probe.py [P C]            (default 134400 1024 = 2 x 200 x 336 pixels, F3 = 1024)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from afigan_amd import _lib

P, Cn = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (134400, 1024)
lib = _lib.load()
vp = C.c_void_p


class BnLoad(C.Structure):
    _fields_ = [("mean", vp), ("invstd", vp), ("gamma", vp), ("beta", vp)]


fwd = getattr(lib, "_Z24afi_launch_disc_tail_fwdPKfPK9AfiBnLoadfS0_PfxiP12ihipStream_t")
fwd.argtypes = [vp, C.POINTER(BnLoad), C.c_float, vp, vp, C.c_longlong, C.c_int, vp]
fwd.restype = C.c_int
bwd = getattr(lib, "_Z24afi_launch_disc_tail_bwdPKfS0_9AfiBnLoadfS0_PfS2_S2_S2_xiS2_S2_P12ihipStream_t")
bwd.argtypes = [vp, vp, BnLoad, C.c_float, vp, vp, vp, vp, vp, C.c_longlong, C.c_int, vp, vp, vp]
bwd.restype = C.c_int
lib.afi_disc_tail_scratch_floats.restype = C.c_longlong

torch.manual_seed(0)
dev = "cuda"
x = torch.randn(P, Cn, device=dev)
mean, invstd = torch.randn(Cn, device=dev) * 0.1, torch.rand(Cn, device=dev) + 0.5
gamma, beta = torch.rand(Cn, device=dev) + 0.5, torch.randn(Cn, device=dev) * 0.1
w3 = torch.randn(9, Cn, device=dev) / Cn ** 0.5
d9 = torch.empty(P, 16, device=dev)
dd9 = torch.zeros(P, 16, device=dev)
dd9[:, :9] = torch.randn(P, 9, device=dev)
dx = torch.empty(P, Cn, device=dev)
dgamma, dbeta, dw3 = torch.zeros(Cn, device=dev), torch.zeros(Cn, device=dev), torch.zeros(9, Cn, device=dev)
scratch = torch.empty(lib.afi_disc_tail_scratch_floats(Cn), device=dev)
amax = torch.zeros(4, device=dev)
bn = BnLoad(mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr())
st = torch.cuda.current_stream().cuda_stream


def run_fwd():
    assert fwd(x.data_ptr(), C.byref(bn), 0.2, w3.data_ptr(), d9.data_ptr(), P, Cn, st) == 0


def run_bwd():
    assert bwd(x.data_ptr(), dd9.data_ptr(), bn, 0.2, w3.data_ptr(), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), dw3.data_ptr(), P, Cn, scratch.data_ptr(),
               amax.data_ptr(), st) == 0


def timed(f, n=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


z = ((x - mean) * invstd) * gamma + beta
y = torch.where(z > 0, z, z * 0.2)
run_fwd()
ref = y.double() @ w3.double().t()
print(f"forward  rel err {((d9[:, :9].double() - ref).abs().max() / ref.abs().max()).item():.2e}; columns 9..15 zero: {bool((d9[:, 9:] == 0).all())}")
us = timed(run_fwd)
print(f"forward  {us:7.1f} us  {P * Cn * 4 / us / 1e6:6.2f} TB/s of the one read of c2 ({P * Cn * 4 / 1e6:.0f} MB)")
run_bwd()
g = dd9[:, :9].double() @ w3.double()
gm = torch.where(z > 0, g, g * 0.2)
xh = ((x - mean) * invstd).double()
s0, s1 = gm.sum(0), (gm * xh).sum(0)
ref_dx = (gamma * invstd).double() * (gm - s0 / P - xh * (s1 / P))
print(f"backward dx rel err {((dx.double() - ref_dx).abs().max() / ref_dx.abs().max()).item():.2e}")
us = timed(run_bwd)
print(f"backward {us:7.1f} us (sums + finalize + apply)  {3 * P * Cn * 4 / us / 1e6:6.2f} TB/s of two reads of c2 and one write")
