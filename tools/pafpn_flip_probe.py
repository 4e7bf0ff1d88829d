"""Anatomy of the PAFPN gradient deviation at 256 channels (profiles/r03/pafpn_flip_anatomy.txt): the operands of every ReLU backward of the
bottom-up merges, captured inside the module, against an fp64 restatement -- incoming gradient, kept activation, mask disagreements, and the
fp64 pre-activation of each disagreeing element.  Finding: ONE element of 49,152 at the 12x16 level (fp64 value 9.4e-8 on a scale of 1.3,
computed as -0.0 on the GPU) carries a gradient of 2.2 against ||dz|| = 150: 1.45e-2 relative L2 on everything behind it."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch.nn.functional as F
import afigan_amd as amd
from afigan_amd import ops
from oracle import afigan_oracle as orc
from test_gpu_fpn import _BottomUp
chans, strides, C = [8, 12, 16, 20], [4, 8, 16, 32], 256
bu = _BottomUp(chans, strides)
torch.manual_seed(11)
net = amd.PAFPN_AFIGAN(bu, ["res2", "res3", "res4", "res5"], C, norm="", top_block=amd.LastLevelMaxPool(), fuse_type="sum").cuda()
with torch.no_grad():
    net.srf_module.load_state_dict(orc.closed_form_generator_params(C, 3, 32))
gen = torch.Generator().manual_seed(8)
feats = {f"res{i + 2}": torch.randn((1, c, 12 * 2 ** (3 - i), 16 * 2 ** (3 - i)), generator=gen) for i, c in enumerate(chans)}
p = {k: v.detach().cpu().double().contiguous().clone().requires_grad_(True) for k, v in net.state_dict().items()}
gp = {k[len("srf_module."):]: v for k, v in p.items() if k.startswith("srf_module.")}
xs = [feats[f"res{i + 2}"].double().clone().requires_grad_(True) for i in range(4)][::-1]
ss = [5, 4, 3, 2]
prev = F.conv2d(xs[0], p["fpn_lateral5.weight"], p["fpn_lateral5.bias"]); topdown = [prev]
for x, s in zip(xs[1:], ss[1:]):
    prev = F.conv2d(x, p[f"fpn_lateral{s}.weight"], p[f"fpn_lateral{s}.bias"]) + orc.generator_forward(prev, gp, 3)
    topdown.insert(0, prev)
pa = topdown[0]; pas = [pa]; zs = []
for inter, s in zip(topdown[1:], [3, 4, 5]):
    z = F.conv2d(pa, p[f"pafpn_downsample{s}.weight"], p[f"pafpn_downsample{s}.bias"], 2, 1); z.retain_grad(); zs.append(z)
    pa = inter + F.relu(z); pa.retain_grad(); pas.append(pa)
p5 = F.conv2d(pa, p["pafpn_output5.weight"], p["pafpn_output5.bias"], 1, 1)
R5 = torch.randn(p5.shape, generator=torch.Generator().manual_seed(303))
(p5 * R5.double()).sum().backward()
rec = []
orig = ops.relu_bwd
def spy(g, act, scale=1.0):
    out = orig(g, act, scale)
    rec.append((g.detach().clone(), act.detach().clone(), out.detach().clone()))
    return out
ops.relu_bwd = spy
import afigan_amd.pafpn_sr as ps
fg = {k: v.cuda().requires_grad_(True) for k, v in feats.items()}
with amd.compute_dtype("fp32"):
    out = net(fg)
    (out["p5"] * R5.cuda()).sum().backward()
def l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-300)).item()
print("relu_bwd calls", len(rec))
for i, (g, act, o) in enumerate(rec):        # backward order: level 5, 4, 3
    lvl = 2 - i
    print(f"merge level {lvl + 3}: dy vs fp64 d(pa) {l2(g, pas[lvl + 1].grad):.2e}  act vs relu(z) {l2(act, F.relu(zs[lvl])):.2e}  mask mismatches {int(((act.cpu() > 0) != (zs[lvl].detach() > 0)).sum())}  dz {l2(o, zs[lvl].grad):.2e}")
g, act, o = rec[0]
mm = ((act.cpu() > 0) != (zs[2].detach() > 0))
idx = mm.nonzero()
print("fp64 pre-activation at the mismatching element(s):", zs[2].detach()[mm].tolist(), " HIP activation there:", act.cpu()[mm].tolist(), " |z| scale:", zs[2].detach().abs().mean().item(),
      " dy there:", g.cpu()[mm].tolist(), " ||dz||:", zs[2].grad.norm().item())
