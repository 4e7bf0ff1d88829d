"""Mid-size maps (1-8 tiles per CU): effect of a forced split-K on the 128x128 kernels (AFI_FORCE_SK, AFI_DBG_SCRATCH_MB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import afigan_amd as amd
from afigan_amd import ops

def t(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

cases = [("G@FPN P3in", 1, 100, 168, 256, 256), ("G0@lrP2", 2, 104, 168, 256, 256), ("G9@P3", 2, 104, 168, 256, 256), ("G@FPN P2out", 1, 200, 336, 256, 256),
         ("G9@P2", 2, 208, 336, 256, 256), ("D0@P3", 2, 100, 168, 256, 512), ("D2@P3", 2, 100, 168, 1024, 1024), ("RDBc5@lrP2", 2, 104, 168, 384, 256)]
for name, N, H, W, Ci, Co in cases:
    x = ops.new_pixel_major(N, Ci, H, W, "cuda"); x.normal_()
    w = ops.new_ohwi(Co, Ci, 3, 3, "cuda", zero=False); w.normal_(0, 0.02)
    dy = ops.new_pixel_major(N, Co, H, W, "cuda"); dy.normal_()
    out = ops.new_pixel_major(N, Co, H, W, "cuda"); dx = ops.new_pixel_major(N, Ci, H, W, "cuda")
    fl = 2.0 * N * H * W * Ci * Co * 9
    ms = t(lambda: ops.conv3x3_fwd(x, w, None, out=out)); ms2 = t(lambda: ops.conv3x3_dgrad(dy, w, dx=dx))
    tiles = -(-N * H * W // 128) * -(-Co // 128)
    print(f"{name:12s} N{N} {H}x{W} {Ci}->{Co} ({tiles} tiles): fwd {ms:7.3f} ms {fl/ms/1e9:6.1f} TF | dgrad {ms2:7.3f} ms {fl/ms2/1e9:6.1f} TF", flush=True)
