"""Does batching the two D calls of a level (N=2 -> N=4) pay?  D fwd+bwd through the module at the stage-1 level sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import afigan_amd as amd

def t(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

D = amd.Discriminator().cuda()
for name, H, W in [("P2", 200, 336), ("P3", 100, 168), ("P4", 50, 84), ("P5", 25, 42), ("P6", 13, 21)]:
    r = []
    for N in (2, 4):
        x = torch.randn(N, 256, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
        def one():
            for p in D.parameters(): p.grad = None
            D(x).sum().backward()
        r.append(t(one))
    print(f"{name} {H}x{W}: N=2 {r[0]:8.3f} ms   N=4 {r[1]:8.3f} ms   2x(N=2)/N=4 = {2 * r[0] / r[1]:.3f}", flush=True)
