"""GPU parity of the AFI path-aggregation pyramid (afigan_amd/pafpn_sr.py) against the CPU oracle's
restatement of pafpn_sr.py:147-193 and against the fixture captured from the imported reference: outputs p2..p6, gradients
w.r.t. the bottom-up features, the lateral / downsample / output convs and the interpolator.  Bar: 1e-3 relative fp32."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import afigan_oracle as orc  # noqa: E402
from test_gpu_fpn import _BottomUp, _rel  # noqa: E402


@pytest.fixture(scope="module")
def amd():
    import afigan_amd
    assert torch.cuda.is_available()
    return afigan_amd


@pytest.mark.parametrize("fuse_type", ["sum", "avg"])
def test_pafpn_afigan_matches_oracle(amd, fuse_type):
    chans, strides, C = [8, 12, 16, 20], [4, 8, 16, 32], 32
    N, H5, W5 = 2, 2, 3
    bu = _BottomUp(chans, strides)
    net = amd.PAFPN_AFIGAN(bu, ["res2", "res3", "res4", "res5"], C, norm="", top_block=amd.LastLevelMaxPool(), fuse_type=fuse_type).cuda()
    want = {"srf_module"} | {f"fpn_lateral{s}" for s in (2, 3, 4, 5)} | {f"pafpn_output{s}" for s in (2, 3, 4, 5)} | {f"pafpn_downsample{s}" for s in (3, 4, 5)}
    assert set(k.split(".")[0] for k in net.state_dict()) == want
    assert net.size_divisibility == 32 and list(net.output_shape()) == ["p2", "p3", "p4", "p5", "p6"]
    gen = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if k.endswith("bias"):
                v.copy_(orc.closed_form_tensor(k, v.shape, 0.05))
        net.srf_module.load_state_dict(orc.closed_form_generator_params(C, 3, 32))
    feats = {f"res{i + 2}": torch.randn((N, c, H5 * 2 ** (3 - i), W5 * 2 ** (3 - i)), generator=gen) for i, c in enumerate(chans)}
    fg = {k: v.cuda().requires_grad_(True) for k, v in feats.items()}
    out = net(fg)
    # <out, R> with fixed random R: O(1) gradients (see tests/test_gpu_fpn.py)
    R = {k: torch.randn(o.shape, generator=torch.Generator().manual_seed(100 + i)) for i, (k, o) in enumerate(out.items())}
    sum((o * R[k].cuda()).sum() for k, o in out.items()).backward()

    pr = {k: v.detach().cpu().contiguous().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    fr = [feats[f"res{i + 2}"].clone().requires_grad_(True) for i in range(4)]
    ref = orc.pafpn_afigan_forward(fr, [2, 3, 4, 5], pr, fuse_type=fuse_type)
    sum((o * R[k]).sum() for k, o in ref.items()).backward()
    assert list(out) == list(ref)
    for k in ref:
        assert _rel(out[k], ref[k]) < 1e-3, k
    for i in range(4):
        assert _rel(fg[f"res{i + 2}"].grad, fr[i].grad) < 1e-3, i
    for k, p in net.named_parameters():
        assert p.grad is not None, k
        assert _rel(p.grad.contiguous(), pr[k].grad) < 1e-3, k


@pytest.mark.parametrize("fuse", ["sum", "avg"])
def test_pafpn_afigan_vs_reference_fixture(amd, golden_dir, fuse):
    """256-channel PAFPN against the fixture captured from the imported reference PAFPN_AFIGAN."""
    from test_oracle_golden import _pafpn_params_and_feats
    fx = dict(np.load(f"{golden_dir}/pafpn_{fuse}.npz"))
    p, feats = _pafpn_params_and_feats(fx)
    bu = _BottomUp([8, 12, 16, 20], [4, 8, 16, 32])
    net = amd.PAFPN_AFIGAN(bu, ["res2", "res3", "res4", "res5"], 256, top_block=amd.LastLevelMaxPool(), fuse_type=fuse).cuda()
    assert set(net.state_dict()) == set(p)
    net.load_state_dict(p, strict=True)
    fg = {f"res{i + 2}": f.cuda().requires_grad_(True) for i, f in enumerate(feats)}
    out = net(fg)
    for k, o in out.items():
        ref = fx["out/" + k]
        got = o.detach().cpu().numpy() if k != "p2" else o.detach().cpu()[:, ::4].numpy()
        assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max(), k
    sum((o * o).mean() for o in out.values()).backward()
    for i in range(4):
        ref = fx[f"dfeat/res{i + 2}"]
        assert np.abs(fg[f"res{i + 2}"].grad.cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max(), i
    for k, q in net.named_parameters():
        f = q.grad.detach().contiguous().reshape(-1).double().cpu()
        rd = fx["gd/" + k]
        assert abs(f.norm().item() - rd[1]) <= 1e-3 * rd[1] + 1e-12, k


def test_pafpn_odd_level_sizes_rejected(amd):
    """The stride-2 output of a level must be the next level's size (what size_divisibility padding guarantees)."""
    bu = _BottomUp([8, 8], [4, 8])
    net = amd.PAFPN_AFIGAN(bu, ["res2", "res3"], 32, top_block=None).cuda()
    feats = {"res2": torch.randn(1, 8, 8, 12, device="cuda"), "res3": torch.randn(1, 8, 4, 6, device="cuda")}
    out = net(feats)
    assert list(out) == ["p2", "p3"] and out["p3"].shape == (1, 32, 4, 6)
