"""GPU parity of the BiFPN inference path (afi-gan_amd/bifpn_sr.py): the per-op pieces against torch-CPU
fp32, and the whole seven-layer forward (28 interpolator calls) against the fixture captured from the imported reference
BiFPN_AFIGAN in eval mode and against the CPU oracle.  Bar: 1e-3 relative fp32."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import afigan_oracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def amd():
    import afigan_amd
    assert torch.cuda.is_available()
    return afigan_amd


def _pm(t):
    return t.cuda().contiguous(memory_format=torch.channels_last)


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.mark.parametrize("N,C,H,W", [(1, 16, 5, 7), (2, 256, 14, 22), (1, 64, 2, 4), (1, 32, 113, 97)])
def test_bifpn_pieces(amd, N, C, H, W):
    ops = amd.ops
    g = torch.Generator().manual_seed(2)
    x, y, z = (torch.randn((N, C, H, W), generator=g) for _ in range(3))
    wdw = torch.randn((C, 1, 3, 3), generator=g)
    ref = F.conv2d(F.pad(x, (1, 1, 1, 1)), wdw, None, 1, 0, 1, C)
    assert _rel(ops.dwconv3x3(_pm(x), wdw.reshape(C, 9).t().contiguous().cuda()), ref) < 1e-5
    assert _rel(ops.maxpool3s2_same(_pm(x)), orc.maxpool3s2_same(x)) == 0.0                     # exact: a max
    assert _rel(ops.maxpool3s2_same(_pm(-x.abs() - 1.0)), orc.maxpool3s2_same(-x.abs() - 1.0)) == 0.0   # the zero pad wins at the border
    w3 = torch.tensor([0.4, 1.1, 0.7])
    assert _rel(ops.fuse_swish(w3.cuda(), _pm(x), _pm(y), _pm(z)), orc.swish(w3[0] * x + w3[1] * y + w3[2] * z)) < 1e-5
    assert _rel(ops.fuse_swish(w3[:2].contiguous().cuda(), _pm(x), _pm(y)), orc.swish(w3[0] * x + w3[1] * y)) < 1e-5


class _BottomUp3(torch.nn.Module):
    _out_feature_strides = {"stage3": 8, "stage4": 16, "stage5": 32}
    _out_feature_channels = {"stage3": 8, "stage4": 12, "stage5": 16}

    def forward(self, feats):
        return feats


def test_bifpn_eval_vs_reference_fixture_and_oracle(amd, golden_dir):
    from test_oracle_golden import _bifpn_params_and_feats
    fx = dict(np.load(f"{golden_dir}/bifpn_eval.npz"))
    p, feats = _bifpn_params_and_feats(fx)
    net = amd.BiFPN_AFIGAN(_BottomUp3(), ["stage3", "stage4", "stage5"], 256, 7, norm="BN", top_block=amd.LastLevelP6P7(16, 256, "BN")).cuda()
    assert set(net.state_dict()) == set(p)                                    # the reference's state_dict contract
    net.load_state_dict(p, strict=True)
    with pytest.raises(amd.AfiError):
        net({f"stage{i + 3}": f.cuda() for i, f in enumerate(feats)})         # training mode is not built
    net.eval()
    out = net({f"stage{i + 3}": f.cuda() for i, f in enumerate(feats)})
    assert list(out) == ["p3", "p4", "p5", "p6", "p7"] and net.size_divisibility == 128
    with torch.no_grad():
        ref = orc.bifpn_afigan_forward(feats, p)
    for k, o in out.items():
        gold = fx["out/" + k]
        assert np.abs(o.cpu().numpy() - gold).max() <= 1e-3 * np.abs(gold).max(), k
        assert _rel(o, ref[k]) < 1e-3, k
    # folded constants follow parameter updates (no stale cache)
    with torch.no_grad():
        net.BiFPNLayer_6_conv3_up.norm.bias.add_(0.5)
    out2 = net({f"stage{i + 3}": f.cuda() for i, f in enumerate(feats)})
    assert _rel(out2["p3"], out["p3"] + 0.5) < 1e-5


def test_bifpn_hipgraph_capture(amd):
    """The inference forward has no host synchronisation: it captures into a hipGraph and replays bit-identically."""
    net = amd.BiFPN_AFIGAN(_BottomUp3(), ["stage3", "stage4", "stage5"], 256, 7, norm="SyncBN", top_block=amd.LastLevelP6P7(16, 256, "SyncBN")).cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(1)
    feats = {f"stage{i + 3}": torch.randn((1, c, 16 // 2 ** i, 32 // 2 ** i), device="cuda", generator=g) for i, c in enumerate([8, 12, 16])}
    eager = net(feats)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = net(feats)
    for v in feats.values():
        v.mul_(0.5)                                        # new input values in the same buffers
    graph.replay()
    torch.cuda.synchronize()
    eager2 = net(feats)
    for k in eager:
        assert torch.equal(captured[k], eager2[k]), k
        assert not torch.equal(captured[k], eager[k]), k
