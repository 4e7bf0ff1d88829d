"""How far do the input gradients of the BiFPN training fixture move when the INPUTS move by one part in 10^6 / 10^7?  Evaluated on the CPU
oracle in fp64, where rounding plays no role: the answer is a property of the function (28 interpolator calls = ~500 LeakyReLU layers and
14 zero-padded max-pools between the inputs and the loss: every kink an input perturbation crosses moves the gradient by a finite amount),
and it bounds from below what ANY fp32 implementation can be held to, since fp32 rounding perturbs every intermediate by ~1e-7 .. 1e-6.
Output: profiles/r03/bifpn_grad_sensitivity.txt (cited by tests/test_gpu_bifpn.py for its gradient bars)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from oracle import afigan_oracle as orc
from test_oracle_golden import _bifpn_train_case

fx = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bifpn_train.npz")))


def grads(eps, seed):
    p, feats, R = _bifpn_train_case(fx)
    p = {k: (v.detach().double().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.detach().double() if v.is_floating_point() else v.clone()))
         for k, v in p.items()}
    g = torch.Generator().manual_seed(seed)
    feats = [(f.detach().double() * (1.0 + eps * torch.randn(f.shape, generator=g, dtype=torch.float64))).requires_grad_(True) for f in feats]
    out = orc.bifpn_afigan_forward(feats, p, train_buffers={})
    sum((o * R[k].double()).sum() for k, o in out.items()).backward()
    return [f.grad for f in feats]


base = grads(0.0, 0)
for eps in (1e-7, 1e-6):
    for seed in (1, 2):
        g = grads(eps, seed)
        print(f"relative input perturbation {eps:.0e} (seed {seed}): d(stage3..5) max-norm deviation "
              + "  ".join(f"{((a - b).abs().max() / b.abs().max()).item():.2e}" for a, b in zip(g, base))
              + "   relative L2 " + "  ".join(f"{((a - b).norm() / b.norm()).item():.2e}" for a, b in zip(g, base)), flush=True)
