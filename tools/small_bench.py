"""One small-map conv (config-1 sizes) in a loop, for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import afigan_amd as amd
from afigan_amd import ops
N, H, W, Ci, Co = 1, 25, 34, 256, 256
x = ops.new_pixel_major(N, Ci, H, W, "cuda"); x.normal_()
w = ops.new_ohwi(Co, Ci, 3, 3, "cuda", zero=False); w.normal_(0, 0.02)
out = ops.new_pixel_major(N, Co, H, W, "cuda")
for _ in range(20):
    ops.conv3x3_fwd(x, w, None, out=out)
torch.cuda.synchronize()
