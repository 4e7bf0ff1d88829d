"""Per-op timing of small-map convs vs K (fixed cost vs per-stage cost of the stream-K kernel). Usage: sk_scan.py [fwd|dgrad|wgrad ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import afigan_amd as amd
from afigan_amd import ops

def t(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us

which = sys.argv[1:] or ["fwd", "dgrad"]
cases = [(1, 25, 34, 256, 256), (1, 25, 34, 1024, 256), (1, 25, 34, 4096, 256), (1, 25, 34, 32, 352), (1, 25, 34, 256, 32), (1, 25, 34, 384, 256),
         (1, 25, 34, 256, 1024), (1, 50, 68, 256, 256)]
for N, H, W, Ci, Co in cases:
    x = ops.new_pixel_major(N, Ci, H, W, "cuda"); x.normal_()
    w = ops.new_ohwi(Co, Ci, 3, 3, "cuda", zero=False); w.normal_(0, 0.02)
    dy = ops.new_pixel_major(N, Co, H, W, "cuda"); dy.normal_()
    out = ops.new_pixel_major(N, Co, H, W, "cuda"); dx = ops.new_pixel_major(N, Ci, H, W, "cuda"); dw = ops.new_ohwi(Co, Ci, 3, 3, "cuda")
    fl = 2.0 * N * H * W * Ci * Co * 9
    r = []
    if "fwd" in which: us = t(lambda: ops.conv3x3_fwd(x, w, None, out=out)); r.append(f"fwd {us:7.1f} us {fl/us/1e6:6.1f} TF")
    if "dgrad" in which: us = t(lambda: ops.conv3x3_dgrad(dy, w, dx=dx)); r.append(f"dgrad {us:7.1f} us {fl/us/1e6:6.1f} TF")
    if "wgrad" in which: us = t(lambda: ops.conv3x3_wgrad(dy, x, dw=dw)); r.append(f"wgrad {us:7.1f} us {fl/us/1e6:6.1f} TF")
    print(f"N{N} {H}x{W} {Ci:4d}->{Co:4d} ideal {fl/157.3e6:6.1f} us: " + " | ".join(r), flush=True)
