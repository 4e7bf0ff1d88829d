"""Pin the CPU oracle (oracle/afigan_oracle.py) against fixtures produced by the imported reference
(tests/golden/make_golden.py).  CPU-only; this is the oracle's parity pin."""
import os

import numpy as np
import pytest
import torch

from oracle import afigan_oracle as orc

torch.set_num_threads(8)


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def _digest(t, nsample=64):
    f = t.detach().reshape(-1).double()
    stride = max(1, f.numel() // nsample)
    return np.array([f.sum().item(), f.norm().item(), f.abs().max().item()]), f[::stride][:nsample].float().numpy()


def _check_digests(fx, grads, prefix="", rtol=2e-4):
    for k, g in grads.items():
        d, s = _digest(g)
        ref_d, ref_s = fx[prefix + "gd/" + k], fx[prefix + "gs/" + k]
        scale = max(ref_d[2], 1e-12)
        # l2 norm and abs-max are scale-stable; the plain sum can cancel, so compare it against norm
        assert abs(d[1] - ref_d[1]) <= rtol * ref_d[1] + 1e-9, (k, d, ref_d)
        assert abs(d[0] - ref_d[0]) <= rtol * ref_d[1] * np.sqrt(g.numel()) + 1e-9, (k, d, ref_d)
        np.testing.assert_allclose(s, ref_s, rtol=0, atol=rtol * scale + 1e-9, err_msg=k)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_generator_small_matches_reference(golden_dir, tag):
    fx = _load(golden_dir, f"g_small_{tag}.npz")
    p = {k[2:]: torch.from_numpy(v).clone().requires_grad_(True) for k, v in fx.items() if k.startswith("w/")}
    x = torch.from_numpy(fx["x"]).requires_grad_(True)
    out = orc.generator_forward(x, p, n_rdb=3)
    np.testing.assert_allclose(out.detach().numpy(), fx["out"], rtol=1e-5, atol=1e-6)
    (out * torch.from_numpy(fx["R"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), fx["dx"], rtol=1e-4, atol=1e-6)
    for k, v in p.items():
        np.testing.assert_allclose(v.grad.numpy(), fx["g/" + k], rtol=1e-4, atol=2e-5, err_msg=k)


def test_generator_full_cfg1_matches_reference(golden_dir):
    fx = _load(golden_dir, "g_full_cfg1.npz")
    p = {k: v.requires_grad_(True) for k, v in orc.closed_form_generator_params().items()}
    x = torch.randn(tuple(fx["x_shape"]), generator=torch.Generator().manual_seed(int(fx["x_seed"][0])))
    x.requires_grad_(True)
    out = orc.generator_forward(x, p)
    assert out.shape == (1, 256, 50, 68)
    scale = float(fx["out_absmax"][0])
    np.testing.assert_allclose(out.detach()[0, ::16, ::5, ::7].numpy(), fx["out_slice"], rtol=0, atol=1e-5 * scale)
    np.testing.assert_allclose(out.detach()[0, :, 17, :].numpy(), fx["out_row"], rtol=0, atol=1e-5 * scale)
    np.testing.assert_allclose(out.detach().double().sum(dim=(0, 2, 3)).numpy(), fx["out_chan_sum"], rtol=0,
                               atol=1e-5 * scale * 3400 ** 0.5)
    out.sum().backward()
    np.testing.assert_allclose(x.grad[0, ::16, ::5, ::7].numpy(), fx["dx_slice"], rtol=1e-4, atol=1e-5)
    _check_digests(fx, {k: v.grad for k, v in p.items()})


def test_bilinear_index_map_bit_exact(golden_dir):
    fx = _load(golden_dir, "bilinear.npz")
    for L in (1, 2, 5, 7, 25):
        i0, i1, lam = orc.bilinear2x_index_map(L)
        ramp = torch.arange(L, dtype=torch.float32)
        up = ramp[i0] * (1 - lam) + ramp[i1] * lam
        assert np.array_equal(up.numpy(), fx[f"ramp_h_{L}"]), L      # integer ramps: exact
        assert np.array_equal(up.numpy(), fx[f"ramp_w_{L}"]), L
        assert set(np.unique(lam.numpy())).issubset({0.0, 0.25, 0.75})
        assert int(i0[0]) == 0 and int(i1[-1]) == L - 1
    out = orc.bilinear2x(torch.from_numpy(fx["rand_in"]))
    np.testing.assert_allclose(out.numpy(), fx["rand_out"], rtol=0, atol=5e-7)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_discriminator_matches_reference(golden_dir, tag):
    fx = _load(golden_dir, f"d_{tag}.npz")
    p0 = orc.closed_form_discriminator_params()
    p = {k: (v.requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in p0.items()}
    x = torch.randn(tuple(fx["x_shape"]), generator=torch.Generator().manual_seed(int(fx["x_seed"][0])))
    x.requires_grad_(True)
    logits, upd = orc.discriminator_forward(x, p, training=True)
    # The last conv is a K=9216 dot product with sum|a*b| ~ 1e3 x |result| and BN layers 2/3 divide by
    # std ~ 0.035: two fp32 evaluations agree to ~5e-4 of max|logit| only.  Bar: 1e-3 of max|logit|.
    np.testing.assert_allclose(logits.detach().numpy(), fx["logits"], rtol=0,
                               atol=1e-3 * np.abs(fx["logits"]).max())
    for k, v in upd.items():
        np.testing.assert_allclose(v.detach().numpy(), fx["buf/" + k], rtol=1e-4, atol=1e-5, err_msg=k)
    (logits * torch.from_numpy(fx["R"])).sum().backward()
    np.testing.assert_allclose(x.grad[0, ::16].numpy(), fx["dx_slice"], rtol=1e-3, atol=1e-4 * fx["gd/x"][2])
    grads = {k: v.grad for k, v in p.items() if v.requires_grad}
    # conv biases feeding a train-mode BN have (mathematically) zero gradient: rounding noise only
    noise = {k for k in grads if k.endswith(".0.bias") and not k.startswith("Discriminators.0.3")}
    for k in noise:
        wn = fx["gd/" + k.replace(".bias", ".weight")][1]
        assert fx["gd/" + k][2] < 1e-4 * wn and grads[k].abs().max().item() < 1e-4 * wn
    _check_digests(fx, {k: g for k, g in grads.items() if k not in noise}, rtol=5e-4)


def test_stage1_step_matches_reference_replay(golden_dir):
    fx = _load(golden_dir, "stage1_step.npz")
    gp = orc.closed_form_generator_params()
    dp = orc.closed_form_discriminator_params()
    gen = torch.Generator().manual_seed(int(fx["seed"][0]))
    lr_f = [torch.randn((2, 256, 13, 21), generator=gen), torch.randn((2, 256, 7, 11), generator=gen)]
    hr_f = [torch.randn((2, 256, 25, 42), generator=gen), torch.randn((2, 256, 13, 21), generator=gen)]
    # Q4: crop sizes
    assert list(fx["crop_p2"]) == [2, 256, 25, 42] and list(fx["crop_p3"]) == [2, 256, 13, 21]

    d_losses, d_grads, d_bufs = orc.stage1_d_phase(gp, dp, lr_f, hr_f)
    for k, v in d_losses.items():
        assert abs(v - float(fx[k][0])) < 2e-5 * abs(float(fx[k][0])) + 1e-6, k
    noise = {k for k in d_grads if k.endswith(".0.bias") and not k.startswith("Discriminators.0.3")}
    _check_digests(fx, {k: g for k, g in d_grads.items() if k not in noise}, prefix="D", rtol=5e-4)

    # D optimizer step, then the G phase sees the UPDATED D (stage1_trainer.py:381 before :384)
    params = {k: v for k, v in dp.items() if k in d_grads}
    mom = {}
    orc.sgd_momentum_step(params, d_grads, mom, lr=float(fx["lr"][0]), momentum=float(fx["mom"][0]),
                          weight_decay=float(fx["wd"][0]))
    dp2 = dict(dp)
    dp2.update(params)
    dp2.update(d_bufs)
    for k in params:
        ref = fx["Dw_after/" + k]
        f = params[k].reshape(-1).double()
        assert abs(f.norm().item() - ref[1]) <= 1e-5 * ref[1] + 1e-9, k

    g_losses, g_grads, d_bufs2 = orc.stage1_g_phase(gp, dp2, lr_f, hr_f)
    for k, v in g_losses.items():
        assert abs(v - float(fx[k][0])) < 5e-5 * abs(float(fx[k][0])) + 1e-6, (k, v, fx[k])
    _check_digests(fx, g_grads, prefix="G", rtol=5e-4)
    # Q2: BN buffers advance 4x per level per iteration
    for k, v in d_bufs2.items():
        ref = fx["Dbuf_after/" + k]
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(ref) == 4 * 2
        else:
            np.testing.assert_allclose(v.numpy(), ref, rtol=2e-4, atol=1e-5, err_msg=k)
    gparams = dict(gp)
    gmom = {}
    orc.sgd_momentum_step(gparams, g_grads, gmom, lr=float(fx["lr"][0]), momentum=float(fx["mom"][0]),
                          weight_decay=float(fx["wd"][0]))
    for k in g_grads:
        ref = fx["Gw_after/" + k]
        assert abs(gparams[k].reshape(-1).double().norm().item() - ref[1]) <= 1e-5 * ref[1] + 1e-9, k


def _fpn_params_and_feats(fx):
    """Closed-form FPN weights (the recipe of make_golden.py) + the seeded bottom-up features."""
    p = {}
    chans = [8, 12, 16, 20]
    for i, c in enumerate(chans):
        s_ = i + 2
        for name, shape in ((f"fpn_lateral{s_}.weight", (256, c, 1, 1)), (f"fpn_lateral{s_}.bias", (256,)),
                            (f"fpn_output{s_}.weight", (256, 256, 3, 3)), (f"fpn_output{s_}.bias", (256,))):
            scale = 0.05 if name.endswith("bias") else (6.0 / (shape[1] * shape[2] * shape[3])) ** 0.5 / 3 ** 0.5
            p[name] = orc.closed_form_tensor(name, shape, scale)
    p.update({"srf_module." + k: v for k, v in orc.closed_form_generator_params().items()})
    gen = torch.Generator().manual_seed(int(fx["seed"][0]))
    feats = [torch.randn((1, c, 2 * 2 ** (3 - i), 3 * 2 ** (3 - i)), generator=gen) for i, c in enumerate(chans)]
    return p, feats


@pytest.mark.parametrize("fuse", ["sum", "avg"])
def test_fpn_merge_matches_reference(golden_dir, fuse):
    """oracle.fpn_afigan_forward vs the imported reference FPN_AFIGAN (fpn_sr.py:127-165)."""
    fx = _load(golden_dir, f"fpn_{fuse}.npz")
    p, feats = _fpn_params_and_feats(fx)
    p = {k: v.requires_grad_(True) for k, v in p.items()}
    feats = [f.requires_grad_(True) for f in feats]
    out = orc.fpn_afigan_forward(feats, [2, 3, 4, 5], p, fuse_type=fuse)
    assert list(out) == ["p2", "p3", "p4", "p5", "p6"]
    for k, o in out.items():
        ref = fx["out/" + k]
        got = o.detach().numpy() if k != "p2" else o.detach()[:, ::4].numpy()
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5 * np.abs(ref).max(), err_msg=k)
    sum((o * o).mean() for o in out.values()).backward()
    for i, f in enumerate(feats):
        ref = fx[f"dfeat/res{i + 2}"]
        np.testing.assert_allclose(f.grad.numpy(), ref, rtol=0, atol=1e-4 * np.abs(ref).max(), err_msg=f"res{i + 2}")
    _check_digests(fx, {k: v.grad for k, v in p.items()}, rtol=5e-4)


def _pafpn_params_and_feats(fx):
    """Closed-form PAFPN weights (the recipe of make_golden.py) + the seeded bottom-up features."""
    p = {}
    chans = [8, 12, 16, 20]
    for i, c in enumerate(chans):
        s_ = i + 2
        names = [(f"fpn_lateral{s_}.weight", (256, c, 1, 1)), (f"fpn_lateral{s_}.bias", (256,)),
                 (f"pafpn_output{s_}.weight", (256, 256, 3, 3)), (f"pafpn_output{s_}.bias", (256,))]
        if i > 0:
            names += [(f"pafpn_downsample{s_}.weight", (256, 256, 3, 3)), (f"pafpn_downsample{s_}.bias", (256,))]
        for name, shape in names:
            scale = 0.05 if name.endswith("bias") else (6.0 / (shape[1] * shape[2] * shape[3])) ** 0.5 / 3 ** 0.5
            p[name] = orc.closed_form_tensor(name, shape, scale)
    p.update({"srf_module." + k: v for k, v in orc.closed_form_generator_params().items()})
    gen = torch.Generator().manual_seed(int(fx["seed"][0]))
    feats = [torch.randn((1, c, 2 * 2 ** (3 - i), 3 * 2 ** (3 - i)), generator=gen) for i, c in enumerate(chans)]
    return p, feats


@pytest.mark.parametrize("fuse", ["sum", "avg"])
def test_pafpn_matches_reference(golden_dir, fuse):
    """oracle.pafpn_afigan_forward vs the imported reference PAFPN_AFIGAN (pafpn_sr.py:147-193)."""
    fx = _load(golden_dir, f"pafpn_{fuse}.npz")
    p, feats = _pafpn_params_and_feats(fx)
    p = {k: v.requires_grad_(True) for k, v in p.items()}
    feats = [f.requires_grad_(True) for f in feats]
    out = orc.pafpn_afigan_forward(feats, [2, 3, 4, 5], p, fuse_type=fuse)
    assert list(out) == ["p2", "p3", "p4", "p5", "p6"]
    for k, o in out.items():
        ref = fx["out/" + k]
        got = o.detach().numpy() if k != "p2" else o.detach()[:, ::4].numpy()
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5 * np.abs(ref).max(), err_msg=k)
    sum((o * o).mean() for o in out.values()).backward()
    for i, f in enumerate(feats):
        ref = fx[f"dfeat/res{i + 2}"]
        np.testing.assert_allclose(f.grad.numpy(), ref, rtol=0, atol=1e-4 * np.abs(ref).max(), err_msg=f"res{i + 2}")
    _check_digests(fx, {k: v.grad for k, v in p.items()}, rtol=5e-4)


def _bifpn_params_and_feats(fx):
    """BiFPN state dict from the fixture's name:shape contract + the closed-form recipe of make_golden.py, and the seeded inputs."""
    import importlib.util
    import json
    import os
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    p = {}
    for entry in fx["state_dict_contract"]:
        name, shape = str(entry).rsplit(":", 1)
        shape = tuple(json.loads(shape))
        p[name] = mg.bifpn_closed_form(name, torch.empty(shape))
    p.update({"srf_module." + k: v for k, v in orc.closed_form_generator_params().items()})
    gen = torch.Generator().manual_seed(int(fx["seed"][0]))
    feats = [torch.randn((1, c, 16 // 2 ** i, 32 // 2 ** i), generator=gen) for i, c in enumerate([8, 12, 16])]
    return p, feats


def test_bifpn_eval_matches_reference(golden_dir):
    """oracle.bifpn_afigan_forward vs the imported reference BiFPN_AFIGAN in eval mode (bifpn_sr.py:569-733): 7 layers, 28
    interpolator calls, the quirks of the hard-wired forward (raw fusion weights, first-lateral skips, zero-padded max-pool)."""
    fx = _load(golden_dir, "bifpn_eval.npz")
    p, feats = _bifpn_params_and_feats(fx)
    assert len(p) == int(fx["n_params"][0])
    with torch.no_grad():
        out = orc.bifpn_afigan_forward(feats, p)
    assert list(out) == ["p3", "p4", "p5", "p6", "p7"]
    for k, o in out.items():
        ref = fx["out/" + k]
        np.testing.assert_allclose(o.numpy(), ref, rtol=0, atol=2e-5 * np.abs(ref).max(), err_msg=k)


def _bifpn_train_case(fx):
    """State dict (leaf tensors), the seeded training inputs and the seeded loss weights of tests/golden/bifpn_train.npz."""
    p, _ = _bifpn_params_and_feats(fx)
    p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in p.items()}
    gen = torch.Generator().manual_seed(int(fx["seed"][0]))
    feats = [torch.randn((2, c, 32 // 2 ** i, 48 // 2 ** i), generator=gen).requires_grad_(True) for i, c in enumerate([8, 12, 16])]
    shapes = {"p3": (2, 256, 32, 48), "p4": (2, 256, 16, 24), "p5": (2, 256, 8, 12), "p6": (2, 256, 4, 6), "p7": (2, 256, 2, 3)}
    R = {k: torch.randn(sh, generator=torch.Generator().manual_seed(200 + i)) for i, (k, sh) in enumerate(shapes.items())}
    return p, feats, R


def test_bifpn_train_matches_reference(golden_dir):
    """oracle.bifpn_afigan_forward in TRAINING mode vs the imported reference BiFPN_AFIGAN.train(): outputs, gradients w.r.t. the bottom-up
    features and every parameter, running statistics after the step."""
    fx = _load(golden_dir, "bifpn_train.npz")
    p, feats, R = _bifpn_train_case(fx)
    bufs = {}
    out = orc.bifpn_afigan_forward(feats, p, train_buffers=bufs)
    sum((o * R[k]).sum() for k, o in out.items()).backward()
    for k, o in out.items():
        ref = fx["out/" + k]
        got = o.detach().numpy() if k != "p3" else o.detach()[:, ::4].numpy()
        np.testing.assert_allclose(got, ref, rtol=0, atol=5e-5 * np.abs(ref).max(), err_msg=k)
    for i, f in enumerate(feats):
        ref = fx[f"dfeat/stage{i + 3}"]
        np.testing.assert_allclose(f.grad.numpy(), ref, rtol=0, atol=5e-4 * np.abs(ref).max(), err_msg=f"stage{i + 3}")
    grads = {k: v.grad for k, v in p.items() if v.requires_grad and v.grad is not None}
    assert {k for k in fx if k.startswith("gd/")} == {"gd/" + k for k in grads}
    # a conv bias in front of a training-mode norm has an analytically ZERO gradient (the norm subtracts the mean): what both sides hold
    # there is summation round-off, compared against the size of the same conv's weight gradient instead of against each other
    dead = {k for k in grads if not k.startswith("srf_module.") and (k.endswith("pointwise.bias") or k.endswith(".0.bias") or k.endswith("p6.conv.bias"))}
    assert len(dead) == 5 + 1 + 7 * 8
    for k in dead:
        wk = k[:-len("bias")] + "weight"
        assert grads[k].abs().max() <= 1e-4 * grads[wk].abs().max() and fx["gd/" + k][2] <= 1e-4 * fx["gd/" + wk][2], k
    _check_digests(fx, {k: g for k, g in grads.items() if k not in dead and not k.startswith("srf_module.")}, rtol=1e-3)
    # the interpolator's weight gradients are sums over its 28 calls that cancel to ~1e-4 of their terms: fp32 ordering noise is larger there
    _check_digests(fx, {k: g for k, g in grads.items() if k.startswith("srf_module.")}, rtol=5e-3)
    n_buf = 0
    for k, v in bufs.items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == 4
            continue
        d, s = _digest(v)
        assert abs(d[1] - fx["bd/" + k][1]) <= 1e-5 * fx["bd/" + k][1], k
        np.testing.assert_allclose(s, fx["bs/" + k], rtol=0, atol=1e-5 * fx["bd/" + k][2], err_msg=k)
        n_buf += 1
    assert n_buf == 2 * (5 + 1 + 7 * 8)


def _stage2_inputs(fx):
    gen = torch.Generator().manual_seed(int(fx["seed"][0]))
    guide = [torch.randn((2, 256, 26, 42), generator=gen), torch.randn((2, 256, 13, 21), generator=gen)]
    fpn = [torch.randn((2, 256, 13, 21), generator=gen), torch.randn((2, 256, 7, 11), generator=gen)]
    return guide, fpn


def test_stage2_adversarial_matches_reference_replay(golden_dir):
    """oracle.stage2_d_phase / stage2_g_losses vs the replay of stage2_trainer.py:299-364 over the imported reference D."""
    fx = _load(golden_dir, "stage2_adv.npz")
    dp = orc.closed_form_discriminator_params()
    guide, fpn = _stage2_inputs(fx)
    assert list(fx["crop_p2"]) == [2, 256, 13, 21] and list(fx["crop_p3"]) == [2, 256, 6, 10]     # 13x21 -> nearest half 6x10, crop
    d_losses, d_grads, d_bufs = orc.stage2_d_phase(dp, guide, fpn)
    for k, v in d_losses.items():
        assert abs(v - float(fx[k][0])) < 2e-5 * abs(float(fx[k][0])) + 1e-6, k
    noise = {k for k in d_grads if k.endswith(".0.bias") and not k.startswith("Discriminators.0.3")}
    # (LeakyReLU-mask flips between two fp32 evaluations: see tests/test_gpu_modules.py; here torch-native BN vs the explicit one)
    _check_digests(fx, {k: g for k, g in d_grads.items() if k not in noise}, prefix="D", rtol=2e-3)
    params = {k: v for k, v in dp.items() if k in d_grads}
    orc.sgd_momentum_step(params, d_grads, {}, lr=float(fx["lr"][0]), momentum=float(fx["mom"][0]), weight_decay=float(fx["wd"][0]))
    dp2 = dict(dp); dp2.update(params); dp2.update(d_bufs)
    fr = [f.clone().requires_grad_(True) for f in fpn]
    out, bufs = orc.stage2_g_losses(dp2, guide, fr)
    sum(v for k, v in out.items() if k.startswith("g_loss")).backward()
    for k, v in out.items():
        assert abs(v.item() - float(fx[k][0])) < 5e-5 * abs(float(fx[k][0])) + 1e-6, k
    for i, f in enumerate(fr):
        np.testing.assert_allclose(f.grad.numpy()[:, ::8], fx[f"dfpn_{i}"], rtol=0, atol=1e-9)
    for k, v in bufs.items():
        ref = fx["Dbuf_after/" + k]
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(ref) == 8
        else:       # the weights behind these statistics already differ by the mask-flip noise of the D gradients
            np.testing.assert_allclose(v.numpy(), ref, rtol=2e-3, atol=1e-3 * np.abs(ref).max(), err_msg=k)
