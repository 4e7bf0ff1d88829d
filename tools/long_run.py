"""Many stage-1 iterations on the bench's synthetic batch, losses printed per iteration (divergence / NaN hunt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import afigan_amd as amd
from afigan_amd.guide import GuideR50FPN
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda")
torch.manual_seed(0)
G = amd.Generator(n_residual_dense_blocks=3).to(dev); D = amd.Discriminator().to(dev)
G.train(); D.train()
step = amd.Stage1Step(G, D, base_lr=1e-3)
guide = GuideR50FPN().to(dev)
gen = torch.Generator(device=dev).manual_seed(100)
images = torch.rand((2, 3, 800, 1333), device=dev, generator=gen) * 255.0
images_half = torch.nn.functional.interpolate(images, size=(400, 666), mode="bilinear", align_corners=False)
hr = guide(images); lr = guide(images_half)
hrs = [hr[f"p{i}"] for i in range(2, 7)]; lrs = [lr[f"p{i}"] for i in range(2, 7)]
print("feature stats", [(float(t.abs().max()), float(t.std())) for t in hrs], flush=True)
for it in range(n_iter):
    step.run_step(lrs, hrs)
    m = step.metrics(check_finite=False)
    gmax = max(float(p.abs().max()) for p in G.parameters()); dmax = max(float(p.abs().max()) for p in D.parameters())
    print(it, " ".join(f"{k}={v:.4g}" for k, v in m.items() if k.startswith(("d_loss", "content"))), f"|G|max={gmax:.3g} |D|max={dmax:.3g}", flush=True)
    if not all(v == v for v in m.values()): break
