import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import afigan_amd as amd
from afigan_amd import ops, _lib
import ctypes as C
lib = _lib.load()
for (N, H, W, Ci, Co) in [(1, 25, 34, 256, 256), (1, 50, 68, 256, 256), (1, 25, 34, 256, 32)]:
    x = ops.new_pixel_major(N, Ci, H, W, "cuda"); x.normal_()
    w = ops.new_ohwi(Co, Ci, 3, 3, "cuda", zero=False); w.normal_(0, 0.02)
    out = ops.new_pixel_major(N, Co, H, W, "cuda")
    for _ in range(5): ops.conv3x3_fwd(x, w, None, out=out)
    torch.cuda.synchronize()
    lib.afi_profile_enable(1)
    for _ in range(50): ops.conv3x3_fwd(x, w, None, out=out)
    torch.cuda.synchronize()
    lib.afi_profile_enable(0)
    tot = 0; n = 0
    for k in range(lib.afi_profile_num_kinds()):
        o = (C.c_double * 3)(); lib.afi_profile_get(k, o)
        if o[0] > 0: tot += o[1]; n += o[0]
    print(f"{N}x{Ci}x{H}x{W}->{Co}: {tot / n * 1e3:7.2f} us per conv (GEMM + split-K pass, HIP events)", flush=True)
