"""The stage-1 loop under two arithmetic settings from identical initial weights and inputs: the loss trajectories side by side over a few hundred
iterations (does the default `f16x3` train like the fp32 MFMA kernels?).  GAN training amplifies rounding differences through LeakyReLU decisions,
so the trajectories separate slowly; what is checked is that they stay finite and close in the sense a re-seeded fp32 run would.
A setting is `dtype` or `dtype:option=value[,option=value]` (context options of both of the engine's contexts), e.g. `fp32:winograd_f4_forward=0`.
Usage: python tools/long_run_dtypes.py [iterations] [setting_a] [setting_b]"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import afigan_amd as amd
from afigan_amd.guide import GuideR50FPN

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dts = (sys.argv[2] if len(sys.argv) > 2 else "f16x3", sys.argv[3] if len(sys.argv) > 3 else "fp32")
dev = torch.device("cuda")
torch.manual_seed(0)
G0 = amd.Generator(n_residual_dense_blocks=3).to(dev); D0 = amd.Discriminator().to(dev)
guide = GuideR50FPN().to(dev)
gen = torch.Generator(device=dev).manual_seed(100)
batches = []
for b in range(4):                                      # four different batches, cycled
    images = torch.rand((2, 3, 800, 1333), device=dev, generator=gen) * 255.0
    images_half = torch.nn.functional.interpolate(images, size=(400, 666), mode="bilinear", align_corners=False)
    hr = guide(images); lr = guide(images_half)
    batches.append(([lr[f"p{i}"] for i in range(2, 7)], [hr[f"p{i}"] for i in range(2, 7)]))
del guide
traj = {}
for dt in dts:
    G, D = copy.deepcopy(G0), copy.deepcopy(D0)
    G.train(); D.train()
    name, _, opts = dt.partition(":")
    eng = amd.Stage1Step(G, D, base_lr=1e-3, dtype=name)
    for kv in filter(None, opts.split(",")):
        k, _, v = kv.partition("=")
        eng.set_option(k, int(v))
    rows = []
    for it in range(n_iter):
        lrs, hrs = batches[it % len(batches)]
        eng.run_step(lrs, hrs)
        if it % 10 == 9 or it < 5:
            m = eng.metrics(check_finite=False)
            rows.append((it, sum(m[f"d_loss_p{l}"] for l in range(2, 7)), sum(m[f"content_loss_p{l}"] for l in range(2, 7)), sum(m[f"adv_loss_p{l}"] for l in range(2, 7))))
    traj[dt] = rows
    del eng, G, D
    torch.cuda.empty_cache()
a, b = traj[dts[0]], traj[dts[1]]
print(f"iteration | sum d_loss {dts[0]} / {dts[1]} | sum content_loss | sum adv_loss")
worst = 0.0
for (it, d0, c0, a0), (_, d1, c1, a1) in zip(a, b):
    print(f"{it:5d} | {d0:10.5f} / {d1:10.5f} | {c0:9.5f} / {c1:9.5f} | {a0:9.5f} / {a1:9.5f}")
    assert all(v == v and abs(v) < 1e30 for v in (d0, c0, a0, d1, c1, a1)), "non-finite loss"
    worst = max(worst, abs(c0 - c1) / abs(c1))
print(f"largest relative difference of the summed content loss: {worst:.3e}")
