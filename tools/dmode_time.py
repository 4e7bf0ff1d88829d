"""D(x) through the drop-in module in train mode: with grad (mode 1: F(2x2) forwards, a backward may follow) and under
torch.no_grad() (mode 2: the cheaper F(4x4) tiling)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import afigan_amd as amd
D = amd.Discriminator().cuda().train()
x = torch.randn(2, 256, 200, 336, device="cuda").contiguous(memory_format=torch.channels_last)
def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 10
def nog():
    with torch.no_grad(): D(x)
print(f"train mode, grad enabled: {t(lambda: D(x)):.3f} ms   under no_grad: {t(nog):.3f} ms", flush=True)
