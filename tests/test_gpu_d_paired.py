"""afi_discriminator_fwd_paired / afi_discriminator_bwd_paired: two consecutive calls of the reference's discriminator (D(real) then D(fake),
stage1_trainer.py:349-359) as ONE call whose BatchNorms take their batch statistics per half.  Checked against the two single calls made one
after the other through the same C-ABI: logits, parameter gradients, running statistics and num_batches_tracked."""
import copy
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import afigan_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import afigan_amd
    return afigan_amd


def _run(amd, D, xs, targets, paired, training=1, dtype=None):
    from afigan_amd import _lib, ops
    net = D.Discriminators[0]
    order = net._ordered_params()
    for p in order:
        p.grad = torch.zeros_like(p)
    prm, keep = net._param_struct(order)
    grd, _ = net._param_struct([p.grad for p in order], already_packed=True, grads=True)
    lib = _lib.load()
    F = (C.c_int * 4)(*net.F)
    ctx = _lib.Ctx(dtype)
    out = []
    with _lib.use_ctx(ctx):
        calls = [(torch.cat(xs, 0), targets)] if paired else [(x, (t,)) for x, t in zip(xs, targets)]
        for x, tg in calls:
            xp = ops.pixel_major(x)
            N, _, H, W = xp.shape
            n = lib.afi_discriminator_fwd_ws_floats(F, N, H, W)
            ws = torch.empty(n, device="cuda")
            logits = torch.empty(N * H * W, device="cuda")
            sfx = "_paired" if paired else ""
            _lib.call("afi_discriminator_fwd" + sfx, C.byref(prm), ops.view_of(xp), N, H, W, C.c_void_p(logits.data_ptr()), training,
                      C.c_void_p(ws.data_ptr()), n, ops.stream_ptr())
            out.append(logits.clone())
            if training != 1:
                continue
            dz = torch.empty_like(logits)
            loss = torch.zeros(1, device="cuda")
            half = logits.numel() // len(tg)
            for h, t in enumerate(tg):
                _lib.call("afi_bce_logits_fwd_bwd", C.c_void_p(logits.data_ptr() + 4 * h * half), half, t, 1.0, C.c_void_p(loss.data_ptr()), 1.0,
                          C.c_void_p(dz.data_ptr() + 4 * h * half), ops.stream_ptr())
            m = lib.afi_discriminator_bwd_ws_floats(F, N, H, W)
            sc = torch.empty(m, device="cuda")
            _lib.call("afi_discriminator_bwd" + sfx, C.byref(prm), C.byref(grd), ops.view_of(xp), N, H, W, C.c_void_p(ws.data_ptr()),
                      C.c_void_p(dz.data_ptr()), C.c_void_p(None), C.c_void_p(sc.data_ptr()), m, ops.stream_ptr())
        _lib.call("afi_ctx_wino_wgrad_flush", ctx.handle, ops.stream_ptr())
    torch.cuda.synchronize()
    return torch.cat(out).cpu(), {k: p.grad.detach().cpu().clone() for k, p in D.named_parameters()}, {k: v.detach().cpu().clone() for k, v in D.state_dict().items()}


@pytest.mark.parametrize("Cf,N,H,W", [(16, 2, 13, 21), (256, 2, 25, 42), (32, 1, 8, 12), (256, 2, 13, 21)])
@pytest.mark.parametrize("training", [1, 2])
def test_paired_call_equals_two_calls(amd, Cf, N, H, W, training):
    D0 = amd.Discriminator(in_filters=Cf).cuda()
    D0.load_state_dict(orc.closed_form_discriminator_params(Cf))
    D0.train()
    gen = torch.Generator().manual_seed(5)
    xs = [torch.randn((N, Cf, H, W), generator=gen).cuda(), (0.5 * torch.randn((N, Cf, H, W), generator=gen) + 0.25).cuda()]
    res = {}
    for paired in (False, True):
        D = copy.deepcopy(D0)
        res[paired] = _run(amd, D, xs, (1.0, 0.0), paired, training)
    (l0, g0, b0), (l1, g1, b1) = res[False], res[True]
    scale = l0.abs().max().item()
    assert (l0 - l1).abs().max().item() <= 2e-5 * scale + 1e-6
    for k in b0:
        if "num_batches" in k:
            assert int(b0[k]) == int(b1[k]) == 2, k
        elif "running" in k:
            np.testing.assert_allclose(b1[k].numpy(), b0[k].numpy(), rtol=1e-5, atol=1e-6 * b0[k].abs().max().item(), err_msg=k)
    if training == 1:
        for k in g0:
            ref = g0[k]
            den = ref.abs().max().item()
            if den == 0.0:
                assert g1[k].abs().max().item() == 0.0, k
                continue
            # LeakyReLU masks are recomputed from conv outputs that differ in the last fp32 bits between the two schedules (at 4 x 13 x 21 the pair
            # runs in Winograd form, the two single calls on the direct kernels): a flipped mask moves a gradient by one pixel's contribution --
            # ONE flipped element moves a weight gradient by up to 1.3e-2 max-norm at these sizes (tests/test_gpu_d_parity.py, header).  The bar
            # admits one such flip; a wrong half, a wrong statistic or a lost term moves a gradient by O(1)
            assert (g1[k] - ref).abs().max().item() <= 1.5e-2 * den, (k, (g1[k] - ref).abs().max().item(), den)


def test_paired_call_refuses_odd_batches_and_ignores_the_folded_affine(amd):
    from afigan_amd import _lib, ops
    D = amd.Discriminator(in_filters=16).cuda()
    net = D.Discriminators[0]
    prm, keep = net._param_struct(net._ordered_params())
    lib = _lib.load()
    F = (C.c_int * 4)(*net.F)
    x = ops.new_pixel_major(3, 16, 8, 12, "cuda", zero=True)
    n = lib.afi_discriminator_fwd_ws_floats(F, 3, 8, 12)
    ws = torch.empty(n, device="cuda"); logits = torch.empty(3 * 8 * 12, device="cuda")
    with pytest.raises(_lib.AfiError):
        _lib.call("afi_discriminator_fwd_paired", C.byref(prm), ops.view_of(x), 3, 8, 12, C.c_void_p(logits.data_ptr()), 1, C.c_void_p(ws.data_ptr()), n,
                  ops.stream_ptr())
    # the folded BatchNorm affine is one per tensor: a paired call under AFI_OPT_D_FOLD_BN_APPLY runs UNFOLDED (each half has its own affine) and gives
    # the results of the same call without the option, bit for bit
    D2 = amd.Discriminator(in_filters=128).cuda()
    net2 = D2.Discriminators[0]
    F2 = (C.c_int * 4)(*net2.F)
    xb = ops.pixel_major(torch.randn((4, 128, 40, 40), generator=torch.Generator().manual_seed(3)).cuda())
    nb = lib.afi_discriminator_fwd_ws_floats(F2, 4, 40, 40)
    outs = []
    for flag in (0, 1):
        cx = _lib.Ctx()
        cx.set_option("d_fold_bn_apply", flag)
        D2.load_state_dict({k: v.clone() for k, v in D2.state_dict().items()})      # (fresh running statistics each way)
        prm2, keep2 = net2._param_struct(net2._ordered_params())
        lg4 = torch.empty(4 * 40 * 40, device="cuda")
        wsb = torch.empty(nb, device="cuda")
        with _lib.use_ctx(cx):
            _lib.call("afi_discriminator_fwd_paired", C.byref(prm2), ops.view_of(xb), 4, 40, 40, C.c_void_p(lg4.data_ptr()), 2, C.c_void_p(wsb.data_ptr()), nb,
                      ops.stream_ptr())
        torch.cuda.synchronize()
        outs.append(lg4.clone())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("k", [0, 10])
def test_paired_halves_of_unlike_scale_bound(amd, k):
    """under f16x3 the two halves of a paired call share ONE operand scale per Winograd plane in the first conv (behind it every
    BatchNorm has normalised the halves separately).  A half whose largest magnitude lies 2^k below the other's keeps 22 - k significand bits
    there.  Measured here at 2 x 256 x 25 x 42 per half (the Winograd path) with the second half scaled by 2^-k: logits of BOTH halves against
    the two separate calls.  k = 0: 2e-5 (rounding).  k = 10: the small half's first-conv operands carry 12 bits -> its logits agree to <= 2e-3
    (2^-12 relative on c0, amplified by nothing: BatchNorm rescales it), the large half's stay at 2e-5.  The stage-1 engine pairs D(real) with
    D(fake) -- guide features and the interpolator's reconstruction of them: like scale by construction; a caller with halves of unlike
    scale pairs nothing (Stage1Step(pair_d_max_pixels=0)) or runs the pair under fp32 / bf16x6, which split exactly."""
    Cf, N, H, W = 256, 2, 25, 42
    D0 = amd.Discriminator(in_filters=Cf).cuda()
    D0.load_state_dict(orc.closed_form_discriminator_params(Cf))
    D0.train()
    gen = torch.Generator().manual_seed(9)
    xs = [torch.randn((N, Cf, H, W), generator=gen).cuda(), (torch.randn((N, Cf, H, W), generator=gen) * 2.0 ** -k).cuda()]
    res = {}
    for paired in (False, True):
        res[paired] = _run(amd, copy.deepcopy(D0), xs, (1.0, 0.0), paired, training=2)[0]
    half = N * H * W
    big = lambda t: t[:half]            # noqa: E731
    small = lambda t: t[half:]          # noqa: E731
    e_big = ((big(res[True]) - big(res[False])).abs().max() / big(res[False]).abs().max()).item()
    e_small = ((small(res[True]) - small(res[False])).abs().max() / small(res[False]).abs().max()).item()
    print(f"halves 2^{k} apart: paired vs separate logits, large half {e_big:.2e}, small half {e_small:.2e}")
    assert e_big <= 2e-5 + 1e-6
    assert e_small <= (2e-5 if k == 0 else 2e-3)
