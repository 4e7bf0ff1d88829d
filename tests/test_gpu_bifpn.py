"""GPU parity of the BiFPN path (afigan_amd/bifpn_sr.py): the per-op pieces and their backward against torch-CPU
fp32 / fp64, the whole seven-layer forward (28 interpolator calls) against the fixtures captured from the imported reference
BiFPN_AFIGAN in eval mode and in training mode (outputs, gradients, running statistics) and against the CPU oracle.
Bar: 1e-3 relative fp32 (named exceptions below)."""
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


@pytest.mark.parametrize("N,C,H,W", [(1, 16, 5, 7), (2, 256, 14, 22), (1, 64, 2, 4), (1, 32, 113, 97), (2, 8, 6, 6)])
def test_bifpn_pieces_backward(amd, N, C, H, W):
    """Backward of the three node pieces against fp64 autograd on the CPU."""
    ops = amd.ops
    g = torch.Generator().manual_seed(3)
    x, y, z = (torch.randn((N, C, H, W), generator=g, dtype=torch.float64).requires_grad_(True) for _ in range(3))
    # fusion + swish, three and two inputs
    for n_in in (3, 2):
        w = torch.tensor([0.4, 1.1, 0.7][:n_in], dtype=torch.float64, requires_grad=True)
        ins = (x, y, z)[:n_in]
        out = orc.swish(sum(w[i] * t for i, t in enumerate(ins)))
        R = torch.randn(out.shape, generator=g, dtype=torch.float64)
        gr = torch.autograd.grad((out * R).sum(), (w,) + ins)
        got = ops.fuse_swish_bwd(w.detach().float().cuda(), *[_pm(t.detach().float()) for t in ins], *([None] if n_in == 2 else []), _pm(R.float()))
        assert _rel(got[0], gr[0].float()) < 1e-4
        for a, b in zip(got[1:1 + n_in], gr[1:]):
            assert _rel(a, b.float()) < 1e-5
        assert n_in == 3 or got[3] is None
    # depthwise 3x3: input gradient (the forward kernel on reversed taps) and weight gradient
    wdw = torch.randn((C, 1, 3, 3), generator=g, dtype=torch.float64, requires_grad=True)
    out = F.conv2d(F.pad(x, (1, 1, 1, 1)), wdw, None, 1, 0, 1, C)
    R = torch.randn(out.shape, generator=g, dtype=torch.float64)
    gx, gw = torch.autograd.grad((out * R).sum(), (x, wdw))
    w9c = wdw.detach().float().reshape(C, 9).t().contiguous().cuda()
    assert _rel(ops.dwconv3x3(_pm(R.float()), w9c.flip(0).contiguous()), gx.float()) < 1e-5
    assert _rel(ops.dwconv3x3_wgrad(_pm(R.float()), _pm(x.detach().float())).t().reshape(C, 1, 3, 3), gw.float()) < 1e-5
    # zero-padded max-pool: same winners as torch (first maximum in scan order, pad included), so the gradient is exact
    for src in (x.detach().float(), -x.detach().float().abs() - 1.0, torch.zeros((N, C, H, W)), (x.detach().float() * 2).round() / 2):
        xs = src.clone().requires_grad_(True)
        ref = orc.maxpool3s2_same(xs)
        Rm = torch.randn(ref.shape, generator=g)
        (gxm,) = torch.autograd.grad((ref * Rm).sum(), xs)
        out, idx = ops.maxpool3s2_same_idx(_pm(src))
        assert torch.equal(out.cpu(), ref.detach())
        dx = ops.maxpool3s2_same_bwd(_pm(Rm), idx, (H, W))
        assert _rel(dx, gxm) < 1e-6


def test_bn_train_function(amd):
    """The training-mode norm used by the BiFPN nodes (eps 1e-3, momentum 0.01) against torch's BatchNorm2d on the CPU."""
    from afigan_amd.bifpn_sr import _BatchNormTrainFn
    g = torch.Generator().manual_seed(4)
    x = (torch.randn((2, 32, 9, 7), generator=g) * 3 + 1).requires_grad_(True)
    ref = torch.nn.BatchNorm2d(32, eps=1e-3, momentum=0.01)
    with torch.no_grad():
        ref.weight.copy_(torch.rand(32, generator=g) + 0.5); ref.bias.copy_(torch.randn(32, generator=g))
        ref.running_mean.copy_(torch.randn(32, generator=g)); ref.running_var.copy_(torch.rand(32, generator=g) + 0.5)
    import copy
    bn = copy.deepcopy(ref).cuda()
    R = torch.randn(x.shape, generator=g)
    yr = ref(x)
    (yr * R).sum().backward()
    xg = _pm(x.detach()).requires_grad_(True)
    out = _BatchNormTrainFn.apply(xg, bn.weight, bn.bias, bn)
    (out * R.cuda()).sum().backward()
    assert _rel(out, yr) < 1e-5 and _rel(xg.grad, x.grad) < 1e-4
    assert _rel(bn.weight.grad, ref.weight.grad) < 1e-5 and _rel(bn.bias.grad, ref.bias.grad) < 1e-5
    assert _rel(bn.running_mean, ref.running_mean) < 1e-6 and _rel(bn.running_var, ref.running_var) < 1e-6
    assert int(bn.num_batches_tracked) == 1


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
    net.eval()
    with torch.no_grad():                                   # inference: the folded fast path (detectron2 runs evaluation under no_grad)
        out = net({f"stage{i + 3}": f.cuda() for i, f in enumerate(feats)})
    assert list(out) == ["p3", "p4", "p5", "p6", "p7"] and net.size_divisibility == 128
    # eval mode with grad mode ON and trainable parameters (frozen-statistics fine-tuning): the differentiable path, attached to the
    # parameters (the fast path used to return detached outputs here), same values
    out_g = net({f"stage{i + 3}": f.cuda() for i, f in enumerate(feats)})
    assert all(o.requires_grad for o in out_g.values())
    for k in out:
        assert _rel(out_g[k], out[k]) < 1e-4, k
    g = torch.autograd.grad(out_g["p3"].sum(), net.BiFPNLayer_6_conv3_up.pointwise.weight)[0]
    assert float(g.abs().sum()) > 0
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


def test_bifpn_train_vs_reference_fixture(amd, golden_dir):
    """Training mode: batch-statistics norms and autograd through all seven layers against the fixture of the imported reference
    BiFPN_AFIGAN.train() -- outputs, gradients w.r.t. the bottom-up features and every parameter, running statistics."""
    from test_oracle_golden import _bifpn_train_case, _check_digests, _digest
    fx = dict(np.load(f"{golden_dir}/bifpn_train.npz"))
    p, feats, R = _bifpn_train_case(fx)
    net = amd.BiFPN_AFIGAN(_BottomUp3(), ["stage3", "stage4", "stage5"], 256, 7, norm="BN", top_block=amd.LastLevelP6P7(16, 256, "BN")).cuda()
    net.load_state_dict({k: v.detach() for k, v in p.items()}, strict=True)
    net.train()
    fg = {f"stage{i + 3}": f.detach().cuda().requires_grad_(True) for i, f in enumerate(feats)}
    out = net(fg)
    sum((o * R[k].cuda()).sum() for k, o in out.items()).backward()
    for k, o in out.items():
        gold = fx["out/" + k]
        got = o.detach().cpu().numpy() if k != "p3" else o.detach()[:, ::4].cpu().numpy()
        assert np.abs(got - gold).max() <= 1e-3 * np.abs(gold).max(), k
    # The input gradients pass 28 interpolator calls (~500 LeakyReLU layers), 14 zero-padded max-pools and 61 training-mode norms.  What
    # bar can ANY fp32 implementation be held to?  tools/bifpn_grad_sensitivity.py (profiles/r03/bifpn_grad_sensitivity.txt) evaluates this
    # very fixture on the fp64 CPU oracle with the inputs perturbed by one part in 10^6 -- exact arithmetic, rounding plays no role -- and
    # the input gradients move by 2e-4 .. 2.8e-3 max-norm (8e-4 L2) depending on which kinks the perturbation crosses; the reference's own
    # fp32 evaluation sits 2e-4 from fp64.  fp32 rounding perturbs every intermediate by 1e-7 .. 1e-6, so the measured 1.5e-3 .. 2.5e-3 of
    # the HIP path (whichever fp32-grade GEMM arithmetic runs) is the function's own sensitivity, and the bar is 5e-3 max-norm.
    for k, f in fg.items():
        gold = fx["dfeat/" + k]
        assert np.abs(f.grad.cpu().numpy() - gold).max() <= 5e-3 * np.abs(gold).max(), k
    grads = {k: v.grad.cpu() for k, v in net.named_parameters() if v.grad is not None}
    assert {k for k in fx if k.startswith("gd/")} == {"gd/" + k for k in grads}
    # (the conv biases in front of a training-mode norm have an analytically zero gradient: see tests/test_oracle_golden.py)
    dead = {k for k in grads if not k.startswith("srf_module.") and (k.endswith("pointwise.bias") or k.endswith(".0.bias") or k.endswith("p6.conv.bias"))}
    for k in dead:
        assert grads[k].abs().max() <= 1e-4 * grads[k[:-len("bias")] + "weight"].abs().max(), k
    _check_digests(fx, {k: g for k, g in grads.items() if k not in dead and not k.startswith("srf_module.")}, rtol=5e-3)    # (mask flips, as above: 1e-3 .. 3e-3)
    _check_digests(fx, {k: g for k, g in grads.items() if k.startswith("srf_module.")}, rtol=1e-2)     # sums over 28 calls that cancel
    sd = net.state_dict()
    for k in sd:
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == 4, k
        elif "running" in k:
            d, s = _digest(sd[k].cpu())
            assert abs(d[1] - fx["bd/" + k][1]) <= 1e-4 * fx["bd/" + k][1], k
            np.testing.assert_allclose(s, fx["bs/" + k], rtol=0, atol=1e-4 * fx["bd/" + k][2], err_msg=k)


def test_bifpn_train_small_vs_oracle(amd):
    """A 32-channel BiFPN, no norm on the top block, frozen interpolator: training forward / backward against the oracle."""
    C = 32

    class Cfg:
        class MODEL:
            AFI_FREEZE = True
    net = amd.BiFPN_AFIGAN(_BottomUp3(), ["stage3", "stage4", "stage5"], C, 7, norm="SyncBN", top_block=amd.LastLevelP6P7(16, C, ""), cfg=Cfg).cuda()
    from test_oracle_golden import _bifpn_params_and_feats  # noqa: F401  (loads make_golden's closed-form recipe)
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    p = {k: mg.bifpn_closed_form(k, v) for k, v in net.state_dict().items() if not k.startswith("srf_module.")}
    p.update({"srf_module." + k: v for k, v in orc.closed_form_generator_params(C, 3, 32).items()})
    net.load_state_dict(p, strict=True)
    net.train()
    g = torch.Generator().manual_seed(9)
    feats = [torch.randn((2, c, 16 // 2 ** i, 48 // 2 ** i), generator=g).requires_grad_(True) for i, c in enumerate([8, 12, 16])]
    fg = {f"stage{i + 3}": f.detach().cuda().requires_grad_(True) for i, f in enumerate(feats)}
    out = net(fg)
    R = {k: torch.randn(o.shape, generator=g) for k, o in out.items()}
    sum((o * R[k].cuda()).sum() for k, o in out.items()).backward()
    pr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and not k.startswith("srf_module.") else v.clone()) for k, v in p.items()}
    bufs = {}
    ref = orc.bifpn_afigan_forward(feats, pr, train_buffers=bufs)
    sum((o * R[k]).sum() for k, o in ref.items()).backward()
    for k in ref:
        assert _rel(out[k], ref[k]) < 1e-3, k
    for i, f in enumerate(feats):
        assert _rel(fg[f"stage{i + 3}"].grad, f.grad) < 2e-3, i
    n = 0
    for k, v in net.named_parameters():
        if k.startswith("srf_module."):
            assert v.grad is None
            continue
        if k.endswith("pointwise.bias") or k.endswith(".0.bias"):
            continue                                       # zero gradient behind a training-mode norm
        assert _rel(v.grad, pr[k].grad) < 2e-3, k
        n += 1
    assert n > 200
    sd = net.state_dict()
    for k, v in bufs.items():
        assert _rel(sd[k].float(), v.float()) < 1e-4, k


def test_bifpn_train_gradients_are_as_exact_as_the_reference_arithmetic(amd):
    """The exact / inexact split of the BiFPN training path (bifpn_sr.py:569-733).

    What is continuous is held tightly: with a smooth x2 map standing in for the interpolator (bilinear up-sampling: no LeakyReLU anywhere)
    the module's OUTPUTS agree with an fp64 evaluation of the oracle to 1e-5 -- fuse + swish, depthwise / pointwise convs and the
    batch-statistics norms of all 7 layers are fp32-exact.

    The GRADIENTS are not a continuous function of the inputs even then: the 14 zero-padded max-pools route each gradient to one of up to
    nine candidates, and a near-tie decided differently moves whole gradient tensors (fusion-weight gradients are sums that nearly cancel).
    Measured on this very network: torch-CPU fp32 -- the reference's own arithmetic -- sits 2e-3 .. 5e-3 from fp64 on the input gradients and
    up to 1e-1 on single fusion weights, with outputs that agree to 1e-6.  So the gradient bars of the BiFPN tests are not kernel tolerances,
    and the statement that can be held is relative: against fp64, the HIP path is no further off than the reference arithmetic is --
      * per input-gradient tensor: err(HIP) <= 3 x err(torch-CPU fp32) + 2e-4;
      * over the ~380 parameter gradients: no more tensors beyond 1e-3 than torch-CPU fp32 has (+ 5), and the error norm over all of them
        (each tensor scaled by its own max-norm) within 3x."""
    C = 32
    net = amd.BiFPN_AFIGAN(_BottomUp3(), ["stage3", "stage4", "stage5"], C, 7, norm="SyncBN", top_block=amd.LastLevelP6P7(16, C, "")).cuda()
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    p = {k: mg.bifpn_closed_form(k, v) for k, v in net.state_dict().items() if not k.startswith("srf_module.")}
    p.update({"srf_module." + k: v for k, v in orc.closed_form_generator_params(C, 3, 32).items()})
    net.load_state_dict(p, strict=True)

    class SmoothUp(torch.nn.Module):                        # the stand-in: same shapes as the interpolator, no decision anywhere
        def forward(self, x):
            return torch.nn.functional.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    net.srf_module = SmoothUp()
    net.train()
    g = torch.Generator().manual_seed(19)
    feats = [torch.randn((2, c, 16 // 2 ** i, 48 // 2 ** i), generator=g) for i, c in enumerate([8, 12, 16])]
    fg = {f"stage{i + 3}": f.detach().cuda().requires_grad_(True) for i, f in enumerate(feats)}
    out = net(fg)
    R = {k: torch.randn(o.shape, generator=g) for k, o in out.items()}
    sum((o * R[k].cuda()).sum() for k, o in out.items()).backward()

    def oracle(dt):
        fs = [f.to(dt).clone().requires_grad_(True) for f in feats]
        pr = {k: ((v.to(dt).clone().requires_grad_(True) if "running" not in k and not k.startswith("srf_module.") else v.to(dt).clone())
                  if v.is_floating_point() else v.clone()) for k, v in p.items()}
        ref = orc.bifpn_afigan_forward(fs, pr, train_buffers={}, upsampler=SmoothUp())
        sum((o * R[k].to(dt)).sum() for k, o in ref.items()).backward()
        return ref, [f.grad for f in fs], {k: v.grad for k, v in pr.items() if v.is_floating_point() and v.grad is not None}
    ref64, df64, dp64 = oracle(torch.float64)
    _, df32, dp32 = oracle(torch.float32)
    for k in ref64:
        assert _rel(out[k], ref64[k]) < 1e-5, k                                  # the continuous part: fp32-exact
    for i in range(3):
        e_hip, e_cpu = _rel(fg[f"stage{i + 3}"].grad, df64[i]), _rel(df32[i], df64[i])
        assert e_hip <= 3 * e_cpu + 2e-4, (i, e_hip, e_cpu)
    eh, ec = [], []
    for k, v in net.named_parameters():
        if k.endswith("pointwise.bias") or k.endswith(".0.bias") or k not in dp64:
            continue                                       # zero gradient behind a training-mode norm
        eh.append(_rel(v.grad, dp64[k])); ec.append(_rel(dp32[k], dp64[k]))
    assert len(eh) > 200
    eh, ec = np.array(eh), np.array(ec)
    assert (eh > 1e-3).sum() <= (ec > 1e-3).sum() + 5, ((eh > 1e-3).sum(), (ec > 1e-3).sum())
    assert np.linalg.norm(eh) <= 3 * np.linalg.norm(ec) + 1e-3, (np.linalg.norm(eh), np.linalg.norm(ec))


def test_bifpn_eval_mode_with_input_gradients(amd):
    """eval() with an input that requires grad: the autograd path with the norms on their running statistics -- the folded inference
    forward's values, and gradients that match the oracle's eval-mode graph."""
    C = 32
    net = amd.BiFPN_AFIGAN(_BottomUp3(), ["stage3", "stage4", "stage5"], C, 7, norm="BN", top_block=amd.LastLevelP6P7(16, C, "BN")).cuda()
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    p = {k: mg.bifpn_closed_form(k, v) for k, v in net.state_dict().items() if not k.startswith("srf_module.")}
    p.update({"srf_module." + k: v for k, v in orc.closed_form_generator_params(C, 3, 32).items()})
    net.load_state_dict(p, strict=True)
    net.eval()
    g = torch.Generator().manual_seed(12)
    feats = [torch.randn((1, c, 16 // 2 ** i, 32 // 2 ** i), generator=g) for i, c in enumerate([8, 12, 16])]
    with torch.no_grad():
        fast = net({f"stage{i + 3}": f.cuda() for i, f in enumerate(feats)})
    fg = {f"stage{i + 3}": f.cuda().requires_grad_(True) for i, f in enumerate(feats)}
    out = net(fg)
    R = {k: torch.randn(o.shape, generator=g) for k, o in out.items()}
    sum((o * R[k].cuda()).sum() for k, o in out.items()).backward()
    fr = [f.clone().requires_grad_(True) for f in feats]
    ref = orc.bifpn_afigan_forward(fr, p)
    sum((o * R[k]).sum() for k, o in ref.items()).backward()
    for k in ref:
        assert _rel(out[k], fast[k]) < 1e-4 and _rel(out[k], ref[k]) < 1e-3, k
    for i, f in enumerate(fr):
        assert _rel(fg[f"stage{i + 3}"].grad, f.grad) < 2e-3, i
    sd = net.state_dict()
    assert all(torch.equal(sd[k].cpu(), v) for k, v in p.items() if "running" in k or "num_batches" in k)      # eval: buffers untouched


def test_bifpn_hipgraph_capture(amd):
    """The inference forward has no host synchronisation: it captures into a hipGraph and replays bit-identically."""
    net = amd.BiFPN_AFIGAN(_BottomUp3(), ["stage3", "stage4", "stage5"], 256, 7, norm="SyncBN", top_block=amd.LastLevelP6P7(16, 256, "SyncBN")).cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(1)
    feats = {f"stage{i + 3}": torch.randn((1, c, 16 // 2 ** i, 32 // 2 ** i), device="cuda", generator=g) for i, c in enumerate([8, 12, 16])}
    with torch.no_grad():                                   # (inference: the folded path; with grad mode on the module builds an autograd graph)
        eager = net(feats)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            captured = net(feats)
        for v in feats.values():
            v.mul_(0.5)                                    # new input values in the same buffers
        graph.replay()
        torch.cuda.synchronize()
        eager2 = net(feats)
    for k in eager:
        assert torch.equal(captured[k], eager2[k]), k
        assert not torch.equal(captured[k], eager[k]), k


def _syncbn_worker(rank, world, port, tmp):
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from afigan_amd import bifpn_sr
    C_ = 24
    torch.manual_seed(7)                                       # the same layer on every rank
    bn = bifpn_sr._make_norm("SyncBN", C_, eps=1e-3, momentum=0.01)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)         # (CPU generator: the same values in the checking process)
    bn = bn.cuda().train()
    assert bn._afi_sync
    g = torch.Generator().manual_seed(100)                     # the GLOBAL batch, of which this rank takes its (unequal) share
    xg = torch.randn((5, C_, 9, 13), generator=g) * 2.0 + 0.7
    rg = torch.randn((5, C_, 9, 13), generator=g)
    sl = slice(0, 2) if rank == 0 else slice(2, 5)
    x = xg[sl].cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = bifpn_sr._norm_train(x, bn)
    (y * rg[sl].cuda()).sum().backward()
    torch.cuda.synchronize()
    out = {"y": y.detach().cpu(), "dx": x.grad.cpu(), "dgamma": bn.weight.grad.cpu(), "dbeta": bn.bias.grad.cpu(),
           "rm": bn.running_mean.cpu(), "rv": bn.running_var.cpu(), "nbt": int(bn.num_batches_tracked)}
    # (a) inside a _SyncTotals block (what BiFPN_AFIGAN._forward_train opens once per forward) the norm reads nothing back from the
    # device -- Tensor.item is counted; (b) the layer's own process group is the one both collectives use
    grp = dist.new_group([0, 1])
    bn._afi_group = grp
    x2 = xg[sl].cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    calls = []
    orig_item = torch.Tensor.item
    torch.Tensor.item = lambda self: (calls.append(1), orig_item(self))[1]
    try:
        with bifpn_sr._SyncTotals([x2.shape[0] * 9 * 13], grp) as tot:
            assert tot.totals == {x2.shape[0] * 9 * 13: 5 * 9 * 13}
            n_before = len(calls)
            y2 = bifpn_sr._norm_train(x2, bn)
            out["item_calls_inside_block"] = len(calls) - n_before
        bn._afi_group = None
        with bifpn_sr._SyncTotals([x2.shape[0] * 9 * 13], grp):          # another group's block: not this layer's -> its own read-back
            n_before = len(calls)
            bifpn_sr._norm_train(x2.detach(), bn)
            out["item_calls_other_group"] = len(calls) - n_before
    finally:
        torch.Tensor.item = orig_item
    (y2 * rg[sl].cuda()).sum().backward()
    torch.cuda.synchronize()
    out.update({"y2": y2.detach().cpu(), "dx2": x2.grad.cpu()})
    torch.save(out, os.path.join(tmp, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_syncbn_statistics_span_the_ranks(tmp_path):
    """norm = "SyncBN" (bifpn_sr.py:210,279-280) in training mode under two ranks (gloo, both on the test GPU, shares of 2 and 3 images): outputs,
    input gradients, running buffers and the SUM of the ranks' parameter gradients equal one nn.BatchNorm2d over the whole batch of five
    (what SyncBatchNorm computes); with per-rank statistics they would not."""
    import torch.multiprocessing as mp
    import os
    port = 29400 + (os.getpid() % 500)
    mp.spawn(_syncbn_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    C_ = 24
    torch.manual_seed(7)
    ref = torch.nn.BatchNorm2d(C_, eps=1e-3, momentum=0.01).double().train()
    from afigan_amd import bifpn_sr
    proto = bifpn_sr._make_norm("SyncBN", C_, eps=1e-3, momentum=0.01)
    with torch.no_grad():
        proto.weight.uniform_(0.5, 1.5); proto.bias.uniform_(-0.5, 0.5)
        ref.weight.copy_(proto.weight.double()); ref.bias.copy_(proto.bias.double())
    g = torch.Generator().manual_seed(100)
    xg = (torch.randn((5, C_, 9, 13), generator=g) * 2.0 + 0.7).double().requires_grad_(True)
    rg = torch.randn((5, C_, 9, 13), generator=g).double()
    y = ref(xg)
    (y * rg).sum().backward()
    rel = lambda a, b: ((a.double() - b).abs().max() / b.abs().max()).item()      # noqa: E731
    assert rel(torch.cat([r0["y"], r1["y"]]), y.detach()) < 1e-5
    assert rel(torch.cat([r0["dx"], r1["dx"]]), xg.grad) < 1e-4
    assert rel(r0["dgamma"] + r1["dgamma"], ref.weight.grad) < 1e-4 and rel(r0["dbeta"] + r1["dbeta"], ref.bias.grad) < 1e-4
    for r in (r0, r1):
        assert rel(r["rm"], ref.running_mean) < 1e-5 and rel(r["rv"], ref.running_var) < 1e-5 and r["nbt"] == 1
        assert r["item_calls_inside_block"] == 0 and r["item_calls_other_group"] >= 1, (r["item_calls_inside_block"], r["item_calls_other_group"])
    assert rel(torch.cat([r0["y2"], r1["y2"]]), y.detach()) < 1e-5 and rel(torch.cat([r0["dx2"], r1["dx2"]]), xg.grad) < 1e-4      # (sub-group, totals looked up)
    # and the test can tell: rank 0's own two images alone give other outputs
    solo = torch.nn.BatchNorm2d(C_, eps=1e-3, momentum=0.01).double().train()
    with torch.no_grad():
        solo.weight.copy_(ref.weight); solo.bias.copy_(ref.bias)
    assert rel(r0["y"], solo(xg.detach()[:2]).detach()) > 1e-2
