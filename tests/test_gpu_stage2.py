"""GPU parity of the stage-2 adversarial terms (afi-gan_amd/stage2.py) against the CPU oracle's
restatement of stage2_trainer.py:299-364 (D step, then the generator-side losses whose L1 gradient reaches the FPN features)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import afigan_oracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def amd():
    import afigan_amd
    assert torch.cuda.is_available()
    return afigan_amd


def test_nearest_half_is_torch_nearest():
    import torch.nn.functional as F
    for shape in ((1, 4, 8, 12), (2, 4, 7, 11), (1, 4, 25, 42)):
        x = torch.randn(shape)
        assert torch.equal(orc.nearest_half(x), F.interpolate(x, scale_factor=0.5))


def test_stage2_adversarial_vs_oracle(amd):
    C = 16
    dp = orc.closed_form_discriminator_params(C)
    D = amd.Discriminator(in_filters=C).cuda()
    D.load_state_dict(dp)
    gen = torch.Generator().manual_seed(8)
    # guide features at full size (odd sizes: 13 -> 6, crop against 7), detector FPN features at ~half size
    guide = [torch.randn((2, C, 26, 42), generator=gen), torch.randn((2, C, 13, 21), generator=gen)]
    fpn = [torch.randn((2, C, 13, 21), generator=gen), torch.randn((2, C, 7, 11), generator=gen)]
    lr0 = 0.01
    s2 = amd.Stage2Adversarial(D, base_lr=lr0, warmup_iters=0, lr_steps=())
    fg = [f.cuda().requires_grad_(True) for f in fpn]
    s2.d_step([g.cuda() for g in guide], fg)
    m = s2.d_metrics()
    d_losses, d_grads, d_bufs = orc.stage2_d_phase(dp, guide, fpn)
    for k, v in d_losses.items():
        assert abs(m[k] - v) <= 1e-3 * abs(v), (k, m[k], v)
    params = {k: v for k, v in dp.items() if k in d_grads}
    orc.sgd_momentum_step(params, d_grads, {}, lr=lr0)
    for k, p in D.named_parameters():
        assert ((p.detach().cpu() - params[k]).abs().max() / params[k].abs().max()).item() < 1e-4, k
    dp2 = dict(dp); dp2.update(params); dp2.update(d_bufs)
    # generator-side terms with the UPDATED D; the content term back-propagates into the FPN features
    out = s2.g_losses([g.cuda() for g in guide], fg)
    sum(out.values()).backward()
    fr = [f.clone().requires_grad_(True) for f in fpn]
    ref, bufs = orc.stage2_g_losses(dp2, guide, fr)
    sum(v for k, v in ref.items() if k.startswith("g_loss")).backward()
    for k in out:
        assert abs(out[k].item() - ref[k].item()) <= 1e-3 * abs(ref[k].item()), (k, out[k].item(), ref[k].item())
    for a, b in zip(fg, fr):
        assert ((a.grad.cpu() - b.grad).abs().max() / b.grad.abs().max()).item() < 1e-5
    sd = D.state_dict()
    for k, v in bufs.items():
        if "num_batches" in k:
            assert int(sd[k]) == int(v) == 8
        else:
            assert ((sd[k].cpu() - v).abs().max() / v.abs().max()).item() < 1e-3, k


def test_stage2_adversarial_vs_reference_fixture(amd, golden_dir):
    """256-channel run against the fixture replayed over the imported reference Discriminator."""
    from test_oracle_golden import _stage2_inputs
    fx = dict(np.load(f"{golden_dir}/stage2_adv.npz"))
    D = amd.Discriminator().cuda()
    D.load_state_dict(orc.closed_form_discriminator_params(), strict=True)
    guide, fpn = _stage2_inputs(fx)
    s2 = amd.Stage2Adversarial(D, base_lr=float(fx["lr"][0]), momentum=float(fx["mom"][0]), weight_decay=float(fx["wd"][0]),
                               warmup_iters=0, lr_steps=())
    fg = [f.cuda().requires_grad_(True) for f in fpn]
    gg = [g.cuda() for g in guide]
    s2.d_step(gg, fg)
    for k, v in s2.d_metrics().items():
        assert abs(v - float(fx[k][0])) <= 1e-3 * abs(float(fx[k][0])), k
    for k, p in D.named_parameters():
        ref = fx["Dw_after/" + k]
        assert abs(p.detach().double().norm().item() - ref[1]) <= 1e-5 * ref[1] + 1e-9, k
    out = s2.g_losses(gg, fg)
    sum(out.values()).backward()
    for k, v in out.items():
        assert abs(v.item() - float(fx[k][0])) <= 1e-3 * abs(float(fx[k][0])), (k, v.item(), fx[k])
    for i, f in enumerate(fg):
        ref = fx[f"dfpn_{i}"]
        assert np.abs(f.grad.cpu().numpy()[:, ::8] - ref).max() <= 1e-5 * np.abs(ref).max() + 1e-12
    sd = D.state_dict()
    for k in sd:
        if "num_batches" in k:
            assert int(sd[k]) == int(fx["Dbuf_after/" + k]) == 8
