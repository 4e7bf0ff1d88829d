"""Where the HOST spends its time while it enqueues one stage-1 step (no device sync inside): per C-ABI entry point and per torch op family,
total / count / max, over the last steps of a short run.  A call whose max is milliseconds is a blocking call (a synchronous copy, a full
queue); the totals say how far ahead of the GPU the host can run.
Usage: python tools/micro/host_enqueue_probe.py [steps]"""
import collections
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import afigan_amd as amd
from afigan_amd import _lib, stage1

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
torch.manual_seed(0)
G = amd.Generator(n_residual_dense_blocks=3).to(dev); D = amd.Discriminator().to(dev)
G.train(); D.train()
eng = amd.Stage1Step(G, D, base_lr=1e-3)
hr_shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
lr_shapes = [(104, 168), (52, 84), (26, 42), (13, 21), (7, 11)]
gen = torch.Generator(device=dev).manual_seed(1)
hr = [torch.randn((2, 256, h, w), device=dev, generator=gen).contiguous(memory_format=torch.channels_last) for h, w in hr_shapes]
lr = [torch.randn((2, 256, h, w), device=dev, generator=gen).contiguous(memory_format=torch.channels_last) for h, w in lr_shapes]

acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
real_call = _lib.call
on = [False]


def timed_call(name, *a):
    t = time.perf_counter()
    r = real_call(name, *a)
    if on[0]:
        d = (time.perf_counter() - t) * 1e3
        e = acc[name]; e[0] += d; e[1] += 1; e[2] = max(e[2], d)
    return r


_lib.call = timed_call
stage1.call = timed_call
for m in (amd.ops,):
    if hasattr(m, "call"):
        m.call = timed_call
for _ in range(3):
    eng.run_step(lr, hr)
torch.cuda.synchronize()
on[0] = True
t0 = time.perf_counter()
host = []
for _ in range(steps):
    t = time.perf_counter()
    eng.run_step(lr, hr)
    host.append((time.perf_counter() - t) * 1e3)
t_enq = (time.perf_counter() - t0) * 1e3
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) * 1e3
print(f"steps {steps}: host enqueue {t_enq / steps:.1f} ms/step, wall {wall / steps:.1f} ms/step; per step host ms: {[round(h, 1) for h in host]}")
tot = sum(v[0] for v in acc.values())
print(f"inside C-ABI calls: {tot / steps:.1f} ms/step of the host's time")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f"  {k:40s} total {v[0] / steps:7.2f} ms/step  calls/step {v[1] / steps:6.1f}  avg {v[0] / v[1] * 1e3:8.1f} us  max {v[2]:7.2f} ms")
