"""Repeat the small FPN_AFIGAN forward/backward of tests/test_gpu_fpn.py and compare every gradient with the first repetition:
a sporadic deviation (race) shows up as an outlier far above the 1e-6 run-to-run noise of the fp32 atomics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import afigan_amd as amd
from oracle import afigan_oracle as orc
from test_gpu_fpn import _BottomUp
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
fuse = sys.argv[2] if len(sys.argv) > 2 else "avg"
chans, strides, C = [8, 12, 16, 20], [4, 8, 16, 32], 32
N, H5, W5 = 2, 2, 3
fpn = amd.FPN_AFIGAN(_BottomUp(chans, strides), ["res2", "res3", "res4", "res5"], C, norm="", top_block=amd.LastLevelMaxPool(), fuse_type=fuse).cuda()
gen = torch.Generator().manual_seed(5)
with torch.no_grad():
    for k, v in fpn.state_dict().items():
        if k.endswith("bias"):
            v.copy_(orc.closed_form_tensor(k, v.shape, 0.05))
    fpn.srf_module.load_state_dict(orc.closed_form_generator_params(C, 3, 32))
feats = {f"res{i + 2}": torch.randn((N, c, H5 * 2 ** (3 - i), W5 * 2 ** (3 - i)), generator=gen).cuda() for i, c in enumerate(chans)}
R = None
ref = None
bad = 0
for it in range(reps):
    fpn.zero_grad(set_to_none=True)
    fg = {k: v.clone().requires_grad_(True) for k, v in feats.items()}
    out = fpn(fg)
    if R is None:
        R = {k: torch.randn(o.shape, generator=torch.Generator().manual_seed(100 + i)).cuda() for i, (k, o) in enumerate(out.items())}
    sum((o * R[k]).sum() for k, o in out.items()).backward()
    cur = {k: p.grad.detach().clone() for k, p in fpn.named_parameters()}
    cur.update({"in/" + k: v.grad.detach().clone() for k, v in fg.items()})
    if ref is None:
        ref = cur
        continue
    worst = max((((cur[k] - ref[k]).abs().max() / (ref[k].abs().max() + 1e-30)).item(), k) for k in ref)
    if worst[0] > 1e-4:
        bad += 1
        print("rep", it, "worst", worst, flush=True)
print("reps", reps, "outliers", bad, flush=True)
