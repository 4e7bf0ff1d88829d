"""GPU parity of the AFI feature-pyramid merge (afigan_amd/fpn_sr.py) against the CPU oracle's
restatement of fpn_sr.py:127-165: outputs p2..p6, gradients w.r.t. the bottom-up features, the lateral / output convs
and the interpolator.  Bar: 1e-3 relative fp32."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from oracle import afigan_oracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def amd():
    import afigan_amd
    assert torch.cuda.is_available()
    return afigan_amd


class _BottomUp(nn.Module):
    """Stand-in bottom-up network: hands back the feature maps it is given (res2..res5) and reports their shapes."""

    def __init__(self, chans, strides):
        super().__init__()
        self.chans, self.strides = chans, strides

    def output_shape(self):
        from afigan_amd.fpn_sr import ShapeSpec
        return {f"res{i + 2}": ShapeSpec(c, s) for i, (c, s) in enumerate(zip(self.chans, self.strides))}

    def forward(self, feats):
        return feats


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.mark.parametrize("fuse_type", ["sum", "avg"])
def test_fpn_afigan_matches_oracle(amd, fuse_type):
    chans, strides, C = [8, 12, 16, 20], [4, 8, 16, 32], 32
    N, H5, W5 = 2, 2, 3
    bu = _BottomUp(chans, strides)
    fpn = amd.FPN_AFIGAN(bu, ["res2", "res3", "res4", "res5"], C, norm="", top_block=amd.LastLevelMaxPool(), fuse_type=fuse_type).cuda()
    assert set(k.split(".")[0] for k in fpn.state_dict()) == {"srf_module"} | {f"fpn_lateral{s}" for s in (2, 3, 4, 5)} | {f"fpn_output{s}" for s in (2, 3, 4, 5)}
    assert fpn.size_divisibility == 32 and list(fpn.output_shape()) == ["p2", "p3", "p4", "p5", "p6"]
    gen = torch.Generator().manual_seed(5)
    with torch.no_grad():                            # non-trivial biases, generator weights large enough to matter
        for k, v in fpn.state_dict().items():
            if k.endswith("bias"):
                v.copy_(orc.closed_form_tensor(k, v.shape, 0.05))
        fpn.srf_module.load_state_dict(orc.closed_form_generator_params(C, 3, 32))
    feats = {f"res{i + 2}": torch.randn((N, c, H5 * 2 ** (3 - i), W5 * 2 ** (3 - i)), generator=gen) for i, c in enumerate(chans)}
    fg = {k: v.cuda().requires_grad_(True) for k, v in feats.items()}
    out = fpn(fg)
    # loss = <out, R> with fixed random R: O(1) gradients everywhere.  (A mean-of-squares loss leaves the interpolator's weight
    # gradients at ~1e-9 through cancellation, where the fp32-atomic summation order of the split weight-gradient GEMM shows
    # up as percent-level relative noise although every term is accurate to 1e-7.)
    R = {k: torch.randn(o.shape, generator=torch.Generator().manual_seed(100 + i)) for i, (k, o) in enumerate(out.items())}
    loss = sum((o * R[k].cuda()).sum() for k, o in out.items())
    # The backward runs TWICE on the one saved forward: the data-gradient chain has no atomics, so the input gradients
    # must agree bit for bit; weight gradients meet in fp32 atomics and may differ by their summation order only.  Each interpolator call's
    # own weight-gradient contribution is tapped before autograd sums the three calls that share the weights, so a mismatch names the call.
    from afigan_amd import generator_rdb as gen_mod
    runs = []
    for rep in range(2):
        for t in list(fg.values()) + list(fpn.parameters()):
            t.grad = None
        gen_mod.grad_tap = taps = []
        try:
            loss.backward(retain_graph=(rep == 0))
        finally:
            gen_mod.grad_tap = None
        runs.append(({k: v.grad.clone() for k, v in fg.items()}, {k: q.grad.contiguous().clone() for k, q in fpn.named_parameters()}, taps))
    gnames = [k for k, _ in fpn.srf_module.named_parameters()]
    order = fpn.srf_module._ordered_params()
    tap_names = [next(n for n, q in fpn.srf_module.named_parameters() if q is o) for o in order]

    def per_call_report():
        lines = []
        for ci, (a, b) in enumerate(zip(runs[0][2], runs[1][2])):
            for n, ga, gb in zip(tap_names, a["grads"], b["grads"]):
                d = (ga - gb).abs().max().item()
                lines.append(f"call {ci} shape {a['shape']} ctx {a['ctx']:#x} {n}: |run1-run2|max {d:.3e} of {ga.abs().max().item():.3e}")
        return "\n".join(l for l in lines if not l.endswith("0.000e+00 of 0.000e+00"))
    assert len(runs[0][2]) == len(runs[1][2]) == 3
    for k in fg:
        assert torch.equal(runs[0][0][k], runs[1][0][k]), f"input gradient {k} differs between two backward passes of one forward\n" + per_call_report()
    nondet = {k: _rel(runs[0][1][k], runs[1][1][k]) for k in runs[0][1]}
    assert max(nondet.values()) <= 1e-6, f"weight gradients differ between two backward passes: { {k: v for k, v in nondet.items() if v > 1e-6} }\n" + per_call_report()
    assert gnames and all(("srf_module." + n) in runs[0][1] for n in gnames)

    pr = {k: v.detach().cpu().contiguous().clone().requires_grad_(True) for k, v in fpn.state_dict().items()}
    fr = [feats[f"res{i + 2}"].clone().requires_grad_(True) for i in range(4)]
    ref = orc.fpn_afigan_forward(fr, [2, 3, 4, 5], pr, fuse_type=fuse_type)
    sum((o * R[k]).sum() for k, o in ref.items()).backward()
    assert list(out) == list(ref)
    for k in ref:
        assert _rel(out[k], ref[k]) < 1e-3, k
    for i in range(4):
        assert _rel(fg[f"res{i + 2}"].grad, fr[i].grad) < 1e-3, i
    bad = {}
    for k, p in fpn.named_parameters():                 # every tensor is checked (no stop at the first), so a failure shows its extent
        assert p.grad is not None, k
        e = _rel(p.grad.contiguous(), pr[k].grad)
        if not e < 1e-3:
            bad[k] = e
    assert not bad, f"parameter gradients off the oracle: {bad}\n" + per_call_report()


def test_fpn_afigan_frozen_interpolator(amd):
    class Cfg:
        class MODEL:
            AFI_FREEZE = True
    bu = _BottomUp([8, 8], [4, 8])
    fpn = amd.FPN_AFIGAN(bu, ["res2", "res3"], 32, top_block=None, cfg=Cfg).cuda()
    assert all(not p.requires_grad for p in fpn.srf_module.parameters())
    feats = {"res2": torch.randn(1, 8, 8, 12, device="cuda"), "res3": torch.randn(1, 8, 4, 6, device="cuda")}
    out = fpn(feats)
    sum(o.sum() for o in out.values()).backward()
    assert all(p.grad is None for p in fpn.srf_module.parameters())
    assert fpn.fpn_lateral3.weight.grad is not None and list(out) == ["p2", "p3"]


@pytest.mark.parametrize("fuse", ["sum", "avg"])
def test_fpn_afigan_vs_reference_fixture(amd, golden_dir, fuse):
    """256-channel FPN merge against the fixture captured from the imported reference FPN_AFIGAN."""
    import numpy as np
    from test_oracle_golden import _fpn_params_and_feats
    fx = dict(np.load(f"{golden_dir}/fpn_{fuse}.npz"))
    p, feats = _fpn_params_and_feats(fx)
    bu = _BottomUp([8, 12, 16, 20], [4, 8, 16, 32])
    fpn = amd.FPN_AFIGAN(bu, ["res2", "res3", "res4", "res5"], 256, top_block=amd.LastLevelMaxPool(), fuse_type=fuse).cuda()
    assert set(fpn.state_dict()) == set(p)
    fpn.load_state_dict(p, strict=True)
    fg = {f"res{i + 2}": f.cuda().requires_grad_(True) for i, f in enumerate(feats)}
    out = fpn(fg)
    for k, o in out.items():
        ref = fx["out/" + k]
        got = o.detach().cpu().numpy() if k != "p2" else o.detach().cpu()[:, ::4].numpy()
        assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max(), k
    sum((o * o).mean() for o in out.values()).backward()
    for i in range(4):
        ref = fx[f"dfeat/res{i + 2}"]
        assert np.abs(fg[f"res{i + 2}"].grad.cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max(), i
    for k, q in fpn.named_parameters():
        f = q.grad.detach().contiguous().reshape(-1).double().cpu()
        rd = fx["gd/" + k]
        assert abs(f.norm().item() - rd[1]) <= 1e-3 * rd[1] + 1e-12, k


def test_fpn_afigan_with_a_norm_layer(amd):
    """cfg.MODEL.FPN.NORM != "" (fpn_sr.py:74-81: convs without bias, each followed by get_norm(norm)): the convs stay on the HIP
    kernels, the norm is torch's; forward and every gradient against a plain torch restatement on the CPU."""
    import torch.nn.functional as F
    chans, strides, C = [8, 12, 16, 20], [4, 8, 16, 32], 32
    bu = _BottomUp(chans, strides)
    fpn = amd.FPN_AFIGAN(bu, ["res2", "res3", "res4", "res5"], C, norm="GN", top_block=amd.LastLevelMaxPool(), fuse_type="sum").cuda()
    assert "fpn_lateral2.bias" not in fpn.state_dict() and "fpn_lateral2.norm.weight" in fpn.state_dict()
    with torch.no_grad():
        fpn.srf_module.load_state_dict(orc.closed_form_generator_params(C, 3, 32))
        for k, v in fpn.state_dict().items():
            if ".norm." in k:
                v.copy_(1.0 + orc.closed_form_tensor(k, v.shape, 0.2) if k.endswith("weight") else orc.closed_form_tensor(k, v.shape, 0.1))
    gen = torch.Generator().manual_seed(6)
    feats = {f"res{i + 2}": torch.randn((2, c, 2 * 2 ** (3 - i), 3 * 2 ** (3 - i)), generator=gen) for i, c in enumerate(chans)}
    fg = {k: v.cuda().requires_grad_(True) for k, v in feats.items()}
    out = fpn(fg)
    R = {k: torch.randn(o.shape, generator=torch.Generator().manual_seed(200 + i)) for i, (k, o) in enumerate(out.items())}
    sum((o * R[k].cuda()).sum() for k, o in out.items()).backward()

    p = {k: v.detach().cpu().contiguous().clone().requires_grad_(True) for k, v in fpn.state_dict().items()}
    fr = {k: v.clone().requires_grad_(True) for k, v in feats.items()}
    gp = {k[len("srf_module."):]: v for k, v in p.items() if k.startswith("srf_module.")}

    def cn(name, x, pad):
        y = F.conv2d(x, p[name + ".weight"], None, 1, pad)
        return F.group_norm(y, 32, p[name + ".norm.weight"], p[name + ".norm.bias"])
    prev = cn("fpn_lateral5", fr["res5"], 0)
    res = [cn("fpn_output5", prev, 1)]
    for s in (4, 3, 2):
        prev = cn(f"fpn_lateral{s}", fr[f"res{s}"], 0) + orc.generator_forward(prev, gp, 3)
        res.insert(0, cn(f"fpn_output{s}", prev, 1))
    res.append(F.max_pool2d(res[-1], kernel_size=1, stride=2, padding=0))
    ref = dict(zip(["p2", "p3", "p4", "p5", "p6"], res))
    sum((o * R[k]).sum() for k, o in ref.items()).backward()
    for k in ref:
        assert _rel(out[k], ref[k]) < 1e-3, k
    for k in fr:
        assert _rel(fg[k].grad, fr[k].grad) < 1e-3, k
    for k, q in fpn.named_parameters():
        assert _rel(q.grad.contiguous(), p[k].grad) < 1e-3, k
