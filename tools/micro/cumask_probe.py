"""CU-masked streams: would a partition of the chip between two concurrent chains pay?  (profiles/HISTORY.md, round 3's open item (c))

    python tools/micro/cumask_probe.py > gpurun_out/cumask_probe.txt

Streams are created with hipExtStreamCreateWithCUMask (ctypes on libamdhip64: no library change for a probe) and handed to torch as
ExternalStream, so the library's launches (it launches on torch's current stream) land on them.  Measured:
  1. one bandwidth-bound pass (a 1 GiB fp32 copy) on masks of 32 ... 256 CUs, low bits and every-other-bit layouts: how many CUs HBM needs
     (measured: a mask bit pair enables a CU pair -- every other bit of 2 n reads as n low bits);
  2. the discriminator fwd+bwd at P2 (2 x 256 x 200 x 336; other sizes: H W arguments) on one stream with 96 ... 256 CUs;
  3. two such chains at once: both unmasked (what Stage1Step's overlap_d does), each on its own half of the chip, and one after the other."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import afigan_amd as amd

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int


def masked_stream(bits):
    """bits: iterable of CU indices (0..255) the stream may use"""
    words = [0] * 8
    for b in bits:
        words[b >> 5] |= 1 << (b & 31)
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, (ctypes.c_uint32 * 8)(*words))
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask: {rc}")
    return torch.cuda.ExternalStream(h.value)


def timed(streams, fns, iters):
    """run fns[i] iters times on streams[i], all at once; wall ms per iteration (of the slowest stream)"""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        for s, f in zip(streams, fns):
            with torch.cuda.stream(s):
                f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / iters


torch.cuda.init()
torch.zeros(1, device="cuda")
layouts = {"low": lambda n: range(n), "even": lambda n: range(0, 2 * n, 2) if n <= 128 else range(n)}
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 336)

print("1. 1 GiB fp32 copy (read + write = 2 GiB) on a CU-masked stream")
src = torch.empty(1 << 28, device="cuda").normal_()
dst = torch.empty_like(src)
for name, lay in layouts.items():
    for n in (32, 64, 96, 128, 160, 192, 256):
        s = masked_stream(lay(n))
        f = lambda: dst.copy_(src)
        timed([s], [f], 2)
        ms = timed([s], [f], 10)
        print(f"   {name:8s} {n:3d} CUs: {ms:7.3f} ms  {2 * src.numel() * 4 / ms / 1e6:7.1f} GB/s", flush=True)
del src, dst

print(f"2./3. discriminator fwd+bwd at 2 x 256 x {H} x {W} (default options)")
torch.manual_seed(0)
nets, xs, rs = [], [], []
for i in range(2):
    nets.append(amd.Discriminator().cuda())
    xs.append(torch.randn(2, 256, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True))
    rs.append(torch.randn(2, 1, H, W, device="cuda"))


def chain(i):
    def f():
        for p in nets[i].parameters():
            p.grad = None
        xs[i].grad = None
        (nets[i](xs[i]) * rs[i]).sum().backward()
    return f


full = [torch.cuda.Stream(), torch.cuda.Stream()]
for name in ("low",):
    lay = layouts[name]
    for n in (96, 128, 160, 192, 256):
        s = masked_stream(lay(n))
        timed([s], [chain(0)], 2)
        print(f"   one chain, {name:8s} {n:3d} CUs: {timed([s], [chain(0)], 6):7.2f} ms", flush=True)
timed(full, [chain(0), chain(1)], 2)
print(f"   one chain, unmasked stream: {timed([full[0]], [chain(0)], 6):7.2f} ms")
print(f"   two chains, one after the other (one stream): {timed([full[0], full[0]], [chain(0), chain(1)], 6):7.2f} ms")
print(f"   two chains at once, both unmasked: {timed(full, [chain(0), chain(1)], 6):7.2f} ms")
for name, a, b in (("128 + 128", range(128), range(128, 256)), ("160 + 96", range(160), range(160, 256)),
                   ("192 (shared 128..191) + 192", range(192), range(64, 256))):
    ss = [masked_stream(a), masked_stream(b)]
    timed(ss, [chain(0), chain(1)], 2)
    print(f"   two chains at once, {name}: {timed(ss, [chain(0), chain(1)], 6):7.2f} ms", flush=True)
