"""One small-map conv per shape with AFI_SK_DIAG set: per-block phase stamps of the stream-K kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import afigan_amd as amd
from afigan_amd import ops
for N, H, W, Ci, Co in [(1, 25, 34, 256, 256), (1, 25, 34, 256, 32), (1, 25, 34, 256, 1024)]:
    x = ops.new_pixel_major(N, Ci, H, W, "cuda"); x.normal_()
    w = ops.new_ohwi(Co, Ci, 3, 3, "cuda", zero=False); w.normal_(0, 0.02)
    dy = ops.new_pixel_major(N, Co, H, W, "cuda"); dy.normal_()
    out = ops.new_pixel_major(N, Co, H, W, "cuda"); dx = ops.new_pixel_major(N, Ci, H, W, "cuda")
    for _ in range(3):
        ops.conv3x3_fwd(x, w, None, out=out)
        ops.conv3x3_dgrad(dy, w, dx=dx)
    torch.cuda.synchronize()
