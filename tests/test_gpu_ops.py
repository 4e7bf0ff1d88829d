"""GPU parity tests of the per-op C-ABI entry points against the CPU oracle / plain torch-CPU fp32 references.
Tolerance: 1e-3 relative fp32 (north-star), measured as max|diff| <= 1e-3 * max|ref| unless stated; index maps bit-exact."""
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


def _close(got, ref, tol=1e-3, what=""):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.2e})"


def _pm(t):
    """CPU NCHW tensor -> GPU pixel-major tensor."""
    return t.cuda().contiguous(memory_format=torch.channels_last)


def _rand(*shape, seed=0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


CONV_CASES = [
    # N, Cin, Cout, H, W
    (1, 16, 16, 5, 7),
    (2, 20, 4, 7, 11),        # growth conv of the small generator (Cin not a multiple of 32)
    (1, 32, 64, 13, 21),
    (2, 256, 32, 25, 34),     # RDB conv1 at config-1 size
    (1, 288, 256, 9, 10),     # K tail: 288 = 9 chunks of 32
    (1, 64, 160, 33, 40),     # Cout not a multiple of the N tile
    (3, 36, 12, 6, 5),
    # large maps: the 8x16-patch HALO variant of the 128x128 kernel (M > 16384 pixels, Cout > 64, patch waste < 12 %)
    (1, 32, 128, 72, 256),    # exact patch grid
    (1, 16, 192, 100, 168),   # ragged patch grid (13 x 11 patches), Cout not a multiple of 128
    (2, 36, 128, 97, 100),    # > 12 % patch waste: stays on the linear-tile kernel
    (1, 256, 256, 72, 256),   # large map, 256-multiple channels
]


@pytest.mark.parametrize("N,Cin,Cout,H,W", CONV_CASES)
def test_conv3x3_fwd_dgrad_wgrad(amd, N, Cin, Cout, H, W):
    ops = amd.ops
    x, w, b = _rand(N, Cin, H, W, seed=1), _rand(Cout, Cin, 3, 3, seed=2) * 0.1, _rand(Cout, seed=3)
    dy = _rand(N, Cout, H, W, seed=4)
    xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv2d(xg, wg, b, 1, 1)
    ref.backward(dy)
    xd, wd, bd, dyd = _pm(x), amd.ops.ohwi(w.cuda()), b.cuda(), _pm(dy)
    _close(ops.conv3x3_fwd(xd, wd, bd), ref, what="fwd")
    _close(ops.conv3x3_fwd(xd, wd, bd, lrelu=True), orc.lrelu(ref), what="fwd+lrelu")
    _close(ops.conv3x3_dgrad(dyd, wd), xg.grad, what="dgrad")
    _close(ops.conv3x3_wgrad(dyd, xd), wg.grad, what="wgrad")
    # alpha / beta / mask epilogues
    old = _rand(N, Cout, H, W, seed=5)
    got = ops.conv3x3_fwd(xd, wd, None, out=_pm(old).clone(memory_format=torch.preserve_format), alpha=0.2, beta=1.0)
    _close(got, 0.2 * F.conv2d(x, w, None, 1, 1) + old, what="alpha/beta")
    z = _rand(N, Cin, H, W, seed=6)
    got = ops.conv3x3_dgrad(dyd, wd, z=_pm(z))
    _close(got, xg.grad * torch.where(z > 0, torch.ones_like(z), torch.full_like(z, 0.2)), what="dgrad mask")


def test_conv3x3_channel_slices_and_crops(amd):
    """Channel slices of a wider buffer (the RDB dense buffer) and cropped input views (stage1_trainer.py:437-443)."""
    ops = amd.ops
    N, H, W = 2, 9, 12
    buf = _rand(N, 48, H, W, seed=1)
    w = _rand(8, 24, 3, 3, seed=2) * 0.1
    bd = _pm(buf)
    ref = orc.lrelu(F.conv2d(buf[:, :24], w, None, 1, 1))
    ops.conv3x3_fwd(bd[:, :24], w.cuda(), None, lrelu=True, out=bd[:, 24:32])
    _close(bd[:, 24:32], ref, what="slice out")
    _close(bd[:, :24], buf[:, :24], tol=0, what="slice in untouched")
    _close(bd[:, 32:], buf[:, 32:], tol=0, what="slice rest untouched")
    big = _rand(N, 16, H + 3, W + 2, seed=3)
    w2 = _rand(12, 16, 3, 3, seed=4) * 0.1
    ref = F.conv2d(big[:, :, :H, :W], w2, None, 1, 1)
    _close(ops.conv3x3_fwd(_pm(big)[:, :, :H, :W], w2.cuda()), ref, what="cropped view in")
    dy = _rand(N, 12, H, W, seed=5)
    xg = big[:, :, :H, :W].clone().requires_grad_(True)
    wg = w2.clone().requires_grad_(True)
    F.conv2d(xg, wg, None, 1, 1).backward(dy)
    _close(ops.conv3x3_wgrad(_pm(dy), _pm(big)[:, :, :H, :W]), wg.grad, what="wgrad cropped x")


@pytest.mark.parametrize("N,C,H,W", [(1, 16, 5, 7), (2, 32, 7, 11), (1, 256, 13, 17)])
def test_conv_transpose_6s2p2(amd, N, C, H, W):
    ops = amd.ops
    x, w, b = _rand(N, C, H, W, seed=1), _rand(C, C, 6, 6, seed=2) * 0.05, _rand(C, seed=3)
    dy = _rand(N, C, 2 * H, 2 * W, seed=4)
    xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv_transpose2d(xg, wg, b, stride=2, padding=2)
    ref.backward(dy)
    wp = ops.convT_pack(w.cuda())
    _close(ops.convT_fwd(_pm(x), wp, b.cuda(), C), ref, what="convT fwd")
    _close(ops.convT_dgrad(_pm(dy), wp, C), xg.grad, what="convT dgrad")
    _close(ops.convT_wgrad(_pm(dy), _pm(x)), wg.grad, what="convT wgrad")
    # oracle's 4-phase restatement agrees with torch's op too
    _close(orc.conv_transpose_6s2p2_phases(x, w, b), ref, tol=1e-5, what="oracle phases")


def test_bilinear2x_index_map_bit_exact(amd, golden_dir):
    ops = amd.ops
    fx = dict(np.load(f"{golden_dir}/bilinear.npz"))
    for L in (1, 2, 5, 7, 25):
        ramp = torch.arange(L, dtype=torch.float32).view(1, 1, L, 1).expand(1, 4, L, 3).contiguous()
        up = ops.bilinear2x(_pm(ramp)).cpu()
        assert np.array_equal(up[0, 0, :, 0].numpy(), fx[f"ramp_h_{L}"]), L          # integer ramps -> exact: index map is bit-exact
        ramp = torch.arange(L, dtype=torch.float32).view(1, 1, 1, L).expand(1, 4, 3, L).contiguous()
        up = ops.bilinear2x(_pm(ramp)).cpu()
        assert np.array_equal(up[0, 0, 0, :].numpy(), fx[f"ramp_w_{L}"]), L
    x = _rand(2, 8, 5, 7, seed=3)
    ref = F.interpolate(x, scale_factor=2, mode="bilinear")
    _close(ops.bilinear2x(_pm(x)), ref, tol=1e-6, what="bilinear fwd")
    dout = _rand(2, 8, 10, 14, seed=4)
    xg = x.clone().requires_grad_(True)
    F.interpolate(xg, scale_factor=2, mode="bilinear").backward(dout)
    _close(ops.bilinear2x_bwd(_pm(dout)), xg.grad, tol=1e-6, what="bilinear bwd")


@pytest.mark.parametrize("P,C", [(77, 8), (546, 64), (3400, 512), (20000, 128)])
def test_batchnorm_train_fwd_bwd(amd, P, C):
    ops = amd.ops
    x = _rand(P, C, seed=1) * 1.7 + 0.3
    gamma, beta = 1 + 0.2 * _rand(C, seed=2), 0.1 * _rand(C, seed=3)
    rm, rv = 0.1 * _rand(C, seed=4), 1 + 0.1 * _rand(C, seed=5).abs()
    g = _rand(P, C, seed=6)
    x4 = x.t().reshape(1, C, P, 1).clone().requires_grad_(True)
    gam, bet = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y, nrm, nrv, mean, var = orc.batchnorm_train(x4, gam, bet, rm, rv)
    out = orc.lrelu(y)
    out.backward(g.t().reshape(1, C, P, 1))
    xd, rmd, rvd = x.cuda(), rm.cuda(), rv.cuda()
    m, istd, v = ops.bn_stats(xd, rmd, rvd)
    _close(m, mean, tol=1e-5, what="mean")
    _close(v, var, tol=1e-4, what="var")
    _close(rmd, nrm, tol=1e-5, what="running_mean")
    _close(rvd, nrv, tol=1e-4, what="running_var")
    yd = ops.bn_apply_lrelu(xd, m, istd, gamma.cuda(), beta.cuda())
    _close(yd, out.reshape(C, P).t(), tol=1e-4, what="bn+lrelu")
    # backward: g masked by lrelu' first (the conv dgrad epilogue does that in the network)
    gm = g.cuda() * torch.where(yd > 0, 1.0, 0.2)
    dgam, dbet = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dx = ops.bn_bwd(gm, xd, m, istd, gamma.cuda(), dgam, dbet)
    _close(dx, x4.grad.reshape(C, P).t(), tol=1e-3, what="bn dx")
    _close(dgam, gam.grad, tol=1e-3, what="dgamma")
    _close(dbet, bet.grad, tol=1e-3, what="dbeta")
    db = torch.zeros(C, device="cuda")
    ops.colsum_accum(g.cuda(), db, alpha=0.5)
    _close(db, 0.5 * g.sum(0), tol=1e-4, what="colsum")


def test_losses(amd):
    ops = amd.ops
    z = _rand(2, 1, 13, 21, seed=1) * 5
    for target in (0.0, 1.0):
        zg = z.clone().requires_grad_(True)
        ref = F.binary_cross_entropy_with_logits(zg, torch.full_like(z, target))
        ref.backward()
        loss = torch.zeros(1, device="cuda")
        dz = ops.bce_logits(z.cuda(), target, loss)
        _close(loss, ref.reshape(1), tol=1e-5, what="bce")
        _close(dz, zg.grad, tol=1e-5, what="bce grad")
        assert abs(orc.bce_with_logits_mean(z, target).item() - ref.item()) < 1e-6
    a, b = _rand(2, 16, 14, 22, seed=2), _rand(2, 16, 13, 21, seed=3)
    ag = a.clone().requires_grad_(True)
    ref = F.l1_loss(ag[:, :, :13, :21], b)
    ref.backward()
    loss = torch.zeros(1, device="cuda")
    da = ops.l1_crop(_pm(a), _pm(b), loss)
    _close(loss, ref.reshape(1), tol=1e-5, what="l1")
    _close(da, ag.grad, tol=1e-6, what="l1 grad (zeros outside the crop)")


def test_layout_roundtrip(amd):
    ops = amd.ops
    x = _rand(2, 20, 7, 9, seed=1).cuda()
    pm = ops.pixel_major(x)
    assert pm.stride(1) == 1 and torch.equal(pm, x)
    assert torch.equal(ops.to_nchw_contiguous(pm), x)


def test_unsupported_channel_counts_are_refused(amd):
    """Channel counts must be multiples of 4 (float4 granularity): refused by the binding's view check or, for raw callers,
    by AFI_ERR_UNSUPPORTED from the C-ABI -- never silently computed."""
    import ctypes as C
    from afigan_amd import _lib
    ops = amd.ops
    with pytest.raises(amd.AfiError):
        ops.conv3x3_fwd(_pm(_rand(1, 6, 5, 5)), _rand(8, 6, 3, 3).cuda())
    with pytest.raises(amd.AfiError):
        ops.conv3x3_fwd(_pm(_rand(1, 8, 5, 5)), _rand(6, 8, 3, 3).cuda())
    with pytest.raises(amd.AfiError):
        ops.conv1x1_fwd(_pm(_rand(1, 8, 5, 5)), _rand(6, 8).cuda())
    x, out = _pm(_rand(1, 8, 5, 5)), _pm(_rand(1, 8, 5, 5))
    w = _rand(6, 8, 3, 3).cuda()
    st = _lib.load().afi_conv3x3_fwd(_lib.current_ctx().handle, ops.view_of(x), 1, 5, 5, 8, C.c_void_p(w.data_ptr()), None, 6, ops.view_of(out), 1.0, 0.0, 0,
                                    ops.stream_ptr())
    assert st == 2 and b"unsupported" in _lib.load().afi_status_string(st)


def test_conv1x1_fwd_dgrad_wgrad(amd):
    ops = amd.ops
    x, w, b = _rand(2, 20, 9, 11, seed=1), _rand(12, 20, 1, 1, seed=2) * 0.2, _rand(12, seed=3)
    add, dy = _rand(2, 12, 9, 11, seed=4), _rand(2, 12, 9, 11, seed=5)
    xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv2d(xg, wg, b) + add
    ref.backward(dy)
    _close(ops.conv1x1_fwd(_pm(x), w.cuda(), b.cuda(), add=_pm(add)), ref, what="1x1 fwd + add")
    _close(ops.conv1x1_dgrad(_pm(dy), w.cuda().reshape(12, 20)), xg.grad, what="1x1 dgrad")
    _close(ops.conv1x1_wgrad(_pm(dy), _pm(x)).reshape(12, 20, 1, 1), wg.grad, what="1x1 wgrad")
    _close(ops.bias_grad(_pm(dy)), dy.sum(dim=(0, 2, 3)), tol=1e-4, what="bias grad")


S2_CASES = [
    # N, Cin, Cout, Hi, Wi   (odd and even sizes: Ho = ceil(Hi/2); the dgrad's odd phases must not write past the input)
    (1, 16, 16, 5, 7),
    (2, 20, 8, 8, 6),
    (1, 32, 64, 13, 21),
    (2, 36, 12, 1, 9),        # single input row
    (1, 64, 160, 34, 41),
    (2, 256, 256, 50, 84),    # PAFPN P4 -> P5 at config-2 size
    (1, 128, 256, 200, 336),  # large map: 128x128 tiles (M = 16800 output pixels)
    (1, 32, 128, 271, 259),   # large odd map
]


@pytest.mark.parametrize("N,Cin,Cout,Hi,Wi", S2_CASES)
def test_conv3x3_stride2(amd, N, Cin, Cout, Hi, Wi):
    """Conv2d(k3, s2, p1) forward (+ fused ReLU / post-activation residual), dgrad (four parity-phase GEMMs) and wgrad
    against torch-CPU fp32 (pafpn_sr.py:105-117,177-183)."""
    ops = amd.ops
    x, w, b = _rand(N, Cin, Hi, Wi, seed=1), _rand(Cout, Cin, 3, 3, seed=2) * 0.1, _rand(Cout, seed=3)
    Ho, Wo = (Hi + 1) // 2, (Wi + 1) // 2
    dy, inter = _rand(N, Cout, Ho, Wo, seed=4), _rand(N, Cout, Ho, Wo, seed=5)
    xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv2d(xg, wg, b, 2, 1)
    assert ref.shape == (N, Cout, Ho, Wo)
    ref.backward(dy)
    xd, wd, bd, dyd = _pm(x), w.cuda(), b.cuda(), _pm(dy)
    _close(ops.conv3x3s2_fwd(xd, wd, bd), ref, what="fwd")
    out, act = ops.conv3x3s2_fwd(xd, wd, bd, act=2, add=_pm(inter), add_scale=0.5, post_scale=0.5, keep_act=True)
    _close(act, F.relu(ref), what="relu output")
    _close(out, 0.5 * inter + 0.5 * F.relu(ref), what="fused merge (avg)")
    _close(ops.conv3x3s2_dgrad(dyd, wd, (Hi, Wi)), xg.grad, what="dgrad")
    _close(ops.conv3x3s2_wgrad(dyd, xd), wg.grad, what="wgrad")
    old = _rand(N, Cin, Hi, Wi, seed=6)
    got = ops.conv3x3s2_dgrad(dyd, wd, (Hi, Wi), dx=_pm(old).clone(memory_format=torch.preserve_format), alpha=0.5, beta=1.0)
    _close(got, old + 0.5 * xg.grad, what="dgrad accumulate")
    g = ops.relu_bwd(dyd, act, scale=0.5)
    _close(g, 0.5 * dy * (act.cpu() > 0).float(), what="relu bwd")              # mask of the kept activation itself


WINO_CASES = [
    # N, Cin, Cout, H, W  (odd sizes: the last tile row / column is half empty; Tpad rounds the tile count up to 128)
    (1, 16, 16, 5, 7),
    (2, 20, 8, 8, 6),
    (1, 32, 64, 13, 21),
    (2, 256, 128, 50, 84),
    (1, 64, 160, 33, 41),
    (1, 128, 256, 101, 169),
]


@pytest.mark.parametrize("N,Cin,Cout,H,W", WINO_CASES)
def test_conv3x3_winograd_fwd_dgrad(amd, N, Cin, Cout, H, W):
    """Winograd F(2x2,3x3) form of the 3x3 conv (forward + bias, data gradient with the LeakyReLU' mask, a cropped-view input)
    against torch-CPU fp32; same 1e-3 bar as the direct kernels (measured ~1e-6)."""
    ops = amd.ops
    x, w, b = _rand(N, Cin, H, W, seed=1), _rand(Cout, Cin, 3, 3, seed=2) * 0.1, _rand(Cout, seed=3)
    dy, z = _rand(N, Cout, H, W, seed=4), _rand(N, Cin, H, W, seed=5)
    xg = x.clone().requires_grad_(True)
    ref = F.conv2d(xg, w, b, 1, 1)
    ref.backward(dy)
    _close(ops.conv3x3_wino_fwd(_pm(x), w.cuda(), b.cuda()), ref, tol=1e-4, what="wino fwd")
    _close(ops.conv3x3_wino_dgrad(_pm(dy), w.cuda()), xg.grad, tol=1e-4, what="wino dgrad")
    mask = torch.where(z > 0, torch.ones_like(z), torch.full_like(z, 0.2))
    _close(ops.conv3x3_wino_dgrad(_pm(dy), w.cuda(), z=_pm(z)), xg.grad * mask, tol=1e-4, what="wino dgrad + mask")
    big = _rand(N, Cin, H + 3, W + 2, seed=6)
    refc = F.conv2d(big[:, :, :H, :W], w, None, 1, 1)
    _close(ops.conv3x3_wino_fwd(_pm(big)[:, :, :H, :W], w.cuda()), refc, tol=1e-4, what="wino fwd, cropped view in")
    # weight gradient, F(3x3,2x2): accumulates into an existing buffer
    wg = w.clone().requires_grad_(True)
    F.conv2d(x, wg, None, 1, 1).backward(dy)
    _close(ops.conv3x3_wino_wgrad(_pm(dy), _pm(x)), wg.grad, tol=1e-4, what="wino wgrad")
    old = _rand(Cout, Cin, 3, 3, seed=7)
    dw0 = ops.ohwi(old.cuda()).clone(memory_format=torch.preserve_format)
    _close(ops.conv3x3_wino_wgrad(_pm(dy), _pm(x), dw=dw0, alpha=0.5), old + 0.5 * wg.grad, tol=1e-4, what="wino wgrad accumulate")


def test_wino_weight_cache_semantics(amd):
    """afi_ctx_set_wino_weight_cache: a transform is computed on first use and re-used until the caller invalidates -- a weight
    change WITHOUT an invalidation is (by contract) not seen, WITH one it is; unregistering restores the per-call transform."""
    import ctypes as C
    from afigan_amd import _lib
    ops = amd.ops
    x, w = _rand(1, 128, 40, 44, seed=11), _rand(128, 128, 3, 3, seed=12) * 0.1
    wd = ops.ohwi(w.cuda()).clone(memory_format=torch.preserve_format)      # one device buffer = one cache key
    ref1 = F.conv2d(x, w, None, 1, 1)
    cache = torch.empty(8 * 1024 * 1024, device="cuda", dtype=torch.float32)
    _lib.call("afi_ctx_set_wino_weight_cache", _lib.current_ctx().handle, C.c_void_p(cache.data_ptr()), cache.numel())
    try:
        _close(ops.conv3x3_wino_fwd(_pm(x), wd), ref1, tol=1e-4, what="first use fills the cache")
        _close(ops.conv3x3_wino_fwd(_pm(x), wd), ref1, tol=1e-4, what="second use hits it")
        wd.mul_(2.0)                                                        # weights move, cache not told
        _close(ops.conv3x3_wino_fwd(_pm(x), wd), ref1, tol=1e-4, what="stale by contract until invalidated")
        _lib.call("afi_ctx_wino_weight_cache_invalidate", _lib.current_ctx().handle)
        _close(ops.conv3x3_wino_fwd(_pm(x), wd), 2.0 * ref1, tol=1e-4, what="after invalidation")
        _close(ops.conv3x3_wino_dgrad(_pm(ref1), wd), torch.autograd.grad(F.conv2d(xg := x.clone().requires_grad_(True), 2.0 * w, None, 1, 1), xg, ref1)[0],
               tol=1e-4, what="the data-gradient transform has its own entry")
    finally:
        _lib.call("afi_ctx_set_wino_weight_cache", _lib.current_ctx().handle, C.c_void_p(None), 0)
    wd.mul_(0.5)
    _close(ops.conv3x3_wino_fwd(_pm(x), wd), ref1, tol=1e-4, what="unregistered: transformed per call")


def test_wino_wgrad_accumulator_semantics(amd):
    """afi_ctx_set_wino_wgrad_accum: calls adding into one dW sum their transform-domain gradients; dW is untouched until
    afi_ctx_wino_wgrad_flush, after which it holds the same total as per-call transforms (both tilings)."""
    import ctypes as C
    from afigan_amd import _lib
    ops = amd.ops
    for (N, Ci, Co, H, W) in [(2, 128, 256, 50, 84), (2, 256, 128, 100, 168)]:           # F(3x3,2x2) and F(3x3,4x4)
        xs = [_rand(N, Ci, H, W, seed=20 + i) for i in range(3)]
        dys = [_rand(N, Co, H, W, seed=30 + i) for i in range(3)]
        w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
        ref = sum(torch.autograd.grad(F.conv2d(x, w, None, 1, 1), w, dy)[0] for x, dy in zip(xs, dys))
        acc = torch.empty(16 * 1024 * 1024, device="cuda", dtype=torch.float32)
        dw = ops.new_ohwi(Co, Ci, 3, 3, "cuda")
        _lib.call("afi_ctx_set_wino_wgrad_accum", _lib.current_ctx().handle, C.c_void_p(acc.data_ptr()), acc.numel())
        try:
            for x, dy in zip(xs, dys):
                ops.conv3x3_wino_wgrad(_pm(dy), _pm(x), dw=dw)
            assert float(dw.abs().max()) == 0.0                                           # nothing lands before the flush
            with pytest.raises(_lib.AfiError):                                            # pending sums: refuse to drop them
                _lib.call("afi_ctx_set_wino_wgrad_accum", _lib.current_ctx().handle, C.c_void_p(None), 0)
        finally:
            _lib.call("afi_ctx_wino_wgrad_flush", _lib.current_ctx().handle, ops.stream_ptr())
            _lib.call("afi_ctx_set_wino_wgrad_accum", _lib.current_ctx().handle, C.c_void_p(None), 0)
        _close(dw, ref, tol=1e-4, what="accumulated wgrad after flush")
        ops.conv3x3_wino_wgrad(_pm(dys[0]), _pm(xs[0]), dw=dw)                            # unregistered again: per call, dw +=
        _close(dw, ref + torch.autograd.grad(F.conv2d(xs[0], w, None, 1, 1), w, dys[0])[0], tol=1e-4, what="per-call path after unregistering")


@pytest.mark.parametrize("N,Cin,Cout,H,W,act", [(1, 128, 128, 40, 44, 2), (2, 128, 256, 72, 100, 2), (1, 256, 128, 100, 168, 1), (1, 128, 128, 96, 100, 0)])
def test_conv3x3_winograd_inference_form(amd, N, Cin, Cout, H, W, act):
    """afi_conv3x3_wino_infer: act(conv + bias) with the F(4x4) tiling from 8192 pixels on (its rounding is ~3e-5 of the output
    scale, the F(2x2) tiling's ~1e-6): against torch-CPU fp32."""
    ops = amd.ops
    x, w, b = _rand(N, Cin, H, W, seed=41), _rand(Cout, Cin, 3, 3, seed=42) * 0.05, _rand(Cout, seed=43)
    ref = F.conv2d(x, w, b, 1, 1)
    ref = F.relu(ref) if act == 2 else (F.leaky_relu(ref, 0.2) if act == 1 else ref)
    out = ops.conv3x3_wino_infer(_pm(x), w.cuda(), b.cuda(), act=act)
    _close(out, ref, tol=3e-4 if N * H * W >= 8192 else 1e-4, what="wino inference conv")


@pytest.mark.parametrize("Cin,Cout,mode", [(256, 256, 0), (256, 256, 1), (64, 32, 0), (32, 96, 1)])
def test_conv_transpose_weight_image_built_from_the_parameter_layout(amd, Cin, Cout, mode):
    """The small-map schedule builds the conv-transpose weight's bf16x6 image straight from W [Cin][Cout][6][6] (LDS-tiled blocks inside
    the image launch) and lets further blocks of that launch write the packed fp32 form: both must equal, byte for byte, what the
    stand-alone pack kernel followed by the generic image job produces (generator_rdb.py:101-105 as a 4-phase 3x3 conv)."""
    import ctypes as C
    from afigan_amd import _lib, ops
    lib = _lib.load()
    W = torch.randn(Cin, Cout, 6, 6, device="cuda", generator=torch.Generator(device="cuda").manual_seed(Cin + Cout + mode))
    nbytes = C.c_longlong(0)
    _lib.check(lib.afi_debug_wk6_convT_images(None, Cin, Cout, mode, None, None, None, None, C.byref(nbytes), None), "size query")
    direct = torch.full((nbytes.value,), 0xA5, dtype=torch.uint8, device="cuda")
    via = torch.full((nbytes.value,), 0x5A, dtype=torch.uint8, device="cuda")
    pack_ride = torch.full((36 * Cin * Cout,), float("nan"), device="cuda")
    pack_ref = torch.full((36 * Cin * Cout,), float("nan"), device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(lib.afi_debug_wk6_convT_images(p(W), Cin, Cout, mode, p(direct), p(pack_ride), p(via), p(pack_ref), C.byref(nbytes), ops.stream_ptr()),
               "afi_debug_wk6_convT_images")
    torch.cuda.synchronize()
    assert torch.equal(pack_ride, pack_ref)
    assert not torch.isnan(pack_ref).any()
    assert torch.equal(direct, via)
