"""When does the training step start relative to the guide forwards issued beside it?  HIP events on the caller's stream (start of run_step) and on the
prefetch stream (start / end of the guide pair), un-profiled, over a few steps of the bench's schedule (take -> submit -> run_step).
Usage: python tools/micro/guide_overlap_probe.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import afigan_amd as amd
from afigan_amd.guide import GuideR50FPN

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
torch.manual_seed(0)
G = amd.Generator(n_residual_dense_blocks=3).to(dev); D = amd.Discriminator().to(dev)
G.train(); D.train()
eng = amd.Stage1Step(G, D, base_lr=1e-3)
guide = GuideR50FPN().to(dev)
images = torch.rand((2, 3, 800, 1333), device=dev) * 255.0
images_half = torch.nn.functional.interpolate(images, size=(400, 666), mode="bilinear", align_corners=False)
ev = []


def guide_pair():
    g0 = torch.cuda.Event(enable_timing=True); g0.record()
    hr_ = guide(images); lr_ = guide(images_half)
    g1 = torch.cuda.Event(enable_timing=True); g1.record()
    ev.append(("guide", g0, g1))
    return [hr_[f"p{d}"] for d in range(2, 7)], [lr_[f"p{d}"] for d in range(2, 7)]


pf = amd.GuidePrefetcher(dev)
pf.submit(guide_pair)
host = []
for i in range(steps + 3):
    if i == 3:
        torch.cuda.synchronize(); ev.clear(); host.clear(); t0 = time.perf_counter()
    hr, lr = pf.take()
    pf.submit(guide_pair)
    s0 = torch.cuda.Event(enable_timing=True); s0.record()
    th = time.perf_counter()
    eng.run_step(lr, hr)
    host.append((time.perf_counter() - th) * 1e3)
    s1 = torch.cuda.Event(enable_timing=True); s1.record()
    ev.append(("step", s0, s1))
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) * 1e3 / steps
base = ev[0][1]
print(f"wall {wall:.2f} ms/step; host ms inside run_step: {[round(h, 1) for h in host]}")
for kind, a, b in ev:
    print(f"{kind:6s} start {base.elapsed_time(a):9.2f}  end {base.elapsed_time(b):9.2f}  ({a.elapsed_time(b):7.2f} ms)")
