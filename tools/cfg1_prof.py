"""config-1 interpolator fwd+bwd in a loop, for `rocprofv3 --kernel-trace --stats`: which kernels the 1.2 ms are made of."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import afigan_amd as amd
if os.environ.get("OLDZ"):           # A/B: one zero-fill per gradient tensor, as before ops.zeros_like_many
    from afigan_amd import ops
    ops.zeros_like_many = lambda ts, need: [torch.zeros_like(w) if n else None for w, n in zip(ts, need)]
torch.manual_seed(0)
G = amd.Generator(n_residual_dense_blocks=3).cuda()
x = torch.randn(1, 256, 25, 34, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
def one():
    for p in G.parameters(): p.grad = None
    x.grad = None
    G(x).sum().backward()
for _ in range(10): one()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(100): one()
b.record(); torch.cuda.synchronize()
print(f"cfg1 fwd+bwd {a.elapsed_time(b) / 100:.4f} ms", flush=True)
