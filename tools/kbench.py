"""Micro-benchmark of the MFMA conv kernels on stage-1 shapes (HIP events via torch on the current stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import afigan_amd as amd
from afigan_amd import ops

def t(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

cases = [("D2@P3", 2, 100, 168, 1024, 1024), ("D1@P2", 2, 200, 336, 512, 1024), ("D0@P2", 2, 200, 336, 256, 512), ("G0@lrP2", 2, 104, 168, 256, 256),
         ("RDBc1@lrP2", 2, 104, 168, 256, 32), ("RDBc5@lrP2", 2, 104, 168, 384, 256), ("D2@P4", 2, 50, 84, 1024, 1024), ("G9@cfg1", 1, 50, 68, 256, 256), ("G0@cfg1", 1, 25, 34, 256, 256)]
which = sys.argv[1:] or ["fwd", "dgrad", "wgrad"]
for name, N, H, W, Ci, Co in (cases if set(which) & {"fwd", "dgrad", "wgrad"} else []):
    x = ops.new_pixel_major(N, Ci, H, W, "cuda"); x.normal_()
    w = ops.new_ohwi(Co, Ci, 3, 3, "cuda", zero=False); w.normal_(0, 0.02)
    dy = ops.new_pixel_major(N, Co, H, W, "cuda"); dy.normal_()
    out = ops.new_pixel_major(N, Co, H, W, "cuda"); dx = ops.new_pixel_major(N, Ci, H, W, "cuda"); dw = ops.new_ohwi(Co, Ci, 3, 3, "cuda")
    fl = 2.0 * N * H * W * Ci * Co * 9
    r = []
    if "fwd" in which: ms = t(lambda: ops.conv3x3_fwd(x, w, None, out=out)); r.append(f"fwd {ms:8.3f} ms {fl/ms/1e9:6.1f} TF")
    if "dgrad" in which: ms = t(lambda: ops.conv3x3_dgrad(dy, w, dx=dx)); r.append(f"dgrad {ms:8.3f} ms {fl/ms/1e9:6.1f} TF")
    if "wgrad" in which: ms = t(lambda: ops.conv3x3_wgrad(dy, x, dw=dw)); r.append(f"wgrad {ms:8.3f} ms {fl/ms/1e9:6.1f} TF")
    print(f"{name:12s} N{N} {H}x{W} {Ci}->{Co}: " + " | ".join(r), flush=True)

# stride-2 3x3 conv (PAFPN bottom-up): input sizes of P2 -> P3, P3 -> P4 at the config-2 / 800x1344 image sizes
if "s2" in which:
    for name, N, Hi, Wi, Ci, Co in [("PA ds3", 2, 200, 336, 256, 256), ("PA ds4", 2, 100, 168, 256, 256), ("PA ds5", 2, 50, 84, 256, 256)]:
        Ho, Wo = (Hi + 1) // 2, (Wi + 1) // 2
        x = ops.new_pixel_major(N, Ci, Hi, Wi, "cuda"); x.normal_()
        w = ops.new_ohwi(Co, Ci, 3, 3, "cuda", zero=False); w.normal_(0, 0.02)
        dy = ops.new_pixel_major(N, Co, Ho, Wo, "cuda"); dy.normal_()
        inter = ops.new_pixel_major(N, Co, Ho, Wo, "cuda"); inter.normal_()
        dx = ops.new_pixel_major(N, Ci, Hi, Wi, "cuda"); dw = ops.new_ohwi(Co, Ci, 3, 3, "cuda")
        fl = 2.0 * N * Ho * Wo * Ci * Co * 9
        r = []
        ms = t(lambda: ops.conv3x3s2_fwd(x, w, None, act=2, add=inter, keep_act=True)); r.append(f"fwd {ms:8.3f} ms {fl/ms/1e9:6.1f} TF")
        ms = t(lambda: ops.conv3x3s2_dgrad(dy, w, (Hi, Wi), dx=dx)); r.append(f"dgrad {ms:8.3f} ms {fl/ms/1e9:6.1f} TF")
        ms = t(lambda: ops.conv3x3s2_wgrad(dy, x, dw=dw)); r.append(f"wgrad {ms:8.3f} ms {fl/ms/1e9:6.1f} TF")
        print(f"{name:12s} N{N} {Hi}x{Wi}->{Ho}x{Wo} {Ci}->{Co}: " + " | ".join(r), flush=True)

# ConvTranspose2d(k6, s2, p2) of the generator (4-phase 3x3 form): forward, data gradient (4 K-phases, stride-2 gather), weight gradient
if "convT" in which:
    for name, N, H, W, C_ in [("GT@lrP2", 2, 104, 168, 256), ("GT@lrP3", 2, 52, 84, 256), ("GT@cfg1", 1, 25, 34, 256)]:
        x = ops.new_pixel_major(N, C_, H, W, "cuda"); x.normal_()
        w = torch.randn(C_, C_, 6, 6, device="cuda") * 0.02
        wp = ops.convT_pack(w)
        dy = ops.new_pixel_major(N, C_, 2 * H, 2 * W, "cuda"); dy.normal_()
        fl = 2.0 * N * H * W * C_ * C_ * 36
        r = []
        ms = t(lambda: ops.convT_fwd(x, wp, None, C_)); r.append(f"fwd {ms:8.3f} ms {fl/ms/1e9:6.1f} TF")
        ms = t(lambda: ops.convT_dgrad(dy, wp, C_)); r.append(f"dgrad {ms:8.3f} ms {fl/ms/1e9:6.1f} TF")
        ms = t(lambda: ops.convT_wgrad(dy, x)); r.append(f"wgrad {ms:8.3f} ms {fl/ms/1e9:6.1f} TF")
        print(f"{name:12s} N{N} {H}x{W} {C_}->{C_}: " + " | ".join(r), flush=True)

# Winograd F(2x2,3x3) form vs the direct kernels on the big discriminator layers
if "wino" in which:
    for name, N, H, W, Ci, Co in [("D2@P2", 2, 200, 336, 1024, 1024), ("D1@P2", 2, 200, 336, 512, 1024), ("D0@P2", 2, 200, 336, 256, 512), ("D2@P3", 2, 100, 168, 1024, 1024),
                                  ("D1@P3", 2, 100, 168, 512, 1024), ("D2@P4", 2, 50, 84, 1024, 1024), ("G0@lrP2", 2, 104, 168, 256, 256)]:
        x = ops.new_pixel_major(N, Ci, H, W, "cuda"); x.normal_()
        w = ops.new_ohwi(Co, Ci, 3, 3, "cuda", zero=False); w.normal_(0, 0.02)
        dy = ops.new_pixel_major(N, Co, H, W, "cuda"); dy.normal_()
        fl = 2.0 * N * H * W * Ci * Co * 9
        r = []
        ms = t(lambda: ops.conv3x3_fwd(x, w, None), iters=5); r.append(f"direct fwd {ms:7.3f} ms")
        ms = t(lambda: ops.conv3x3_wino_fwd(x, w, None), iters=5); r.append(f"wino fwd {ms:7.3f} ms ({fl/ms/1e9:6.1f} eff TF)")
        ms = t(lambda: ops.conv3x3_dgrad(dy, w), iters=5); r.append(f"direct dgrad {ms:7.3f} ms")
        ms = t(lambda: ops.conv3x3_wino_dgrad(dy, w), iters=5); r.append(f"wino dgrad {ms:7.3f} ms ({fl/ms/1e9:6.1f} eff TF)")
        dw = ops.new_ohwi(Co, Ci, 3, 3, "cuda")
        ms = t(lambda: ops.conv3x3_wgrad(dy, x, dw=dw), iters=5); r.append(f"direct wgrad {ms:7.3f} ms")
        ms = t(lambda: ops.conv3x3_wino_wgrad(dy, x, dw=dw), iters=5); r.append(f"wino wgrad {ms:7.3f} ms ({fl/ms/1e9:6.1f} eff TF)")
        print(f"{name:8s} N{N} {H}x{W} {Ci}->{Co}: " + " | ".join(r), flush=True)
