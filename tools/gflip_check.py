"""The interpolator's forward + backward (loss = <out, r>) against an fp64 evaluation, per forward-conv algorithm: the counterpart of tools/dflip_check.py
for G, whose LeakyReLUs sit behind the 32-channel growth convs that read the Winograd convs' outputs.  Variants: torch fp32 / fp64 (the oracle's op
sequence on the host cores: test infrastructure, a tool only), and this library with the interpolator's forwards on F(2x2) (default) or F(4x4)
(winograd_f4_forward bit 16), or direct (winograd off).  Relative L2 of dx and of the worst parameter gradient.
Usage: python tools/gflip_check.py [H W]   (lr map; default 52 84 at batch 2: the P3 call of the stage-1 step)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import afigan_amd as amd
from oracle import afigan_oracle as orc

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (52, 84)
torch.manual_seed(0)
G = amd.Generator(n_residual_dense_blocks=3).cuda().train()
x0 = torch.randn(2, 256, H, W, device="cuda")
r = torch.randn(2, 256, 2 * H, 2 * W, device="cuda")
names = [n for n, _ in G.named_parameters()]
sd = {k: v.detach().clone() for k, v in G.state_dict().items()}


def torch_run(dt):
    torch.backends.cudnn.allow_tf32 = False
    p = {k: v.cpu().to(dt).contiguous().requires_grad_(True) for k, v in sd.items()}      # logical OIHW shapes; the oracle runs on the host
    xx = x0.cpu().to(dt).requires_grad_(True)
    out = orc.generator_forward(xx, p, n_rdb=3)
    (out * r.cpu().to(dt)).sum().backward()
    return {"dx": xx.grad.double().cpu(), **{n: p[n].grad.double().cpu() for n in names}}


def hip_run(opts):
    cx = amd._lib.current_ctx()
    base = {k: cx.get_option(k) for k in opts}
    for k, v in opts.items():
        cx.set_option(k, v)
    try:
        for q in G.parameters():
            q.grad = None
        x = x0.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        (G(x) * r).sum().backward()
        return {"dx": x.grad.double().cpu(), **{n: q.grad.double().cpu() for n, q in G.named_parameters()}}
    finally:
        for k, v in base.items():
            cx.set_option(k, v)


ref = torch_run(torch.float64)
f4 = amd._lib.current_ctx().get_option("winograd_f4_forward")
runs = {"torch fp32 ops": torch_run(torch.float32), "direct (winograd = 0)": hip_run({"winograd": 0}),
        "F(2x2) forwards in G": hip_run({"winograd_f4_forward": f4 & ~16}), "F(4x4) forwards in G": hip_run({"winograd_f4_forward": f4 | 16})}
live = [k for k in ref if ref[k].norm() > 1e-9 * ref[k].numel() ** 0.5]
for tag, o in runs.items():
    worst, wname = max((((o[k] - ref[k]).norm() / ref[k].norm()).item(), k) for k in live)
    dx = ((o["dx"] - ref["dx"]).norm() / ref["dx"].norm()).item()
    print(f"{tag:26s} vs fp64: dx rel-L2 {dx:.3e}   worst tensor rel-L2 {worst:.3e} ({wname})", flush=True)
