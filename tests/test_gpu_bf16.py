"""The arithmetic settings of the Winograd-domain GEMMs (afi_ctx_set_compute_dtype; BASELINE.json configs[4] "bf16"): each against the
fp64 / fp32 oracle with ITS OWN stated tolerance.  The reference is fp32-only: the default (bf16x6, an exact three-way split) and fp32 are
held to the same fp32 bars here and in every other test file; bf16x3 and bf16 are opt-in.

    dtype      operands                         GEMM alone (max-norm)   conv3x3 fwd / dgrad / wgrad   generator fwd+bwd, D step
    fp32       exact fp32 MFMA                  2e-6                    1e-4                          1e-3
    bf16x6     x = hi + mid + lo, 6 bf16 MFMAs  2e-6                    1e-4                          1e-3 (the default: all other tests)
    f16x3      x s = hi + lo (fp16), 3 f16 MFMAs 2e-6                   1e-4                          1e-3
    bf16x3     x = hi + lo, 3 bf16 MFMAs        2e-5                    2e-4                          1e-3 outputs / input grads; 1e-2 (L2) weight grads
    bf16       x -> bf16(x), 1 bf16 MFMA        8e-3                    2e-2 (F(2x2) tiles only)      5e-2 outputs / input grads; 1e-1 (L2) weight grads
"""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import afigan_oracle as orc  # noqa: E402

TOL_GEMM = {"fp32": 2e-6, "bf16x6": 2e-6, "f16x3": 2e-6, "bf16x3": 2e-5, "bf16": 8e-3}
TOL_CONV = {"fp32": 1e-4, "bf16x6": 1e-4, "f16x3": 1e-4, "bf16x3": 2e-4, "bf16": 2e-2}
TOL_NET = {"fp32": 1e-3, "bf16x6": 1e-3, "f16x3": 1e-3, "bf16x3": 1e-3, "bf16": 5e-2}
TOL_GRAD_L2 = {"fp32": 5e-3, "bf16x6": 5e-3, "f16x3": 5e-3, "bf16x3": 1e-2, "bf16": 1e-1}
ALL_DTYPES = ["fp32", "bf16x6", "f16x3", "bf16x3", "bf16"]


@pytest.fixture(scope="module")
def amd():
    import afigan_amd
    assert torch.cuda.is_available()
    return afigan_amd


DEFAULT_DTYPE = "f16x3"                                   # AFI_DTYPE_DEFAULT of include/afigan_hip.h


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / (b.abs().max() + 1e-300)).item()


@pytest.mark.parametrize("dtype", ALL_DTYPES)
def test_batched_gemms(amd, dtype):
    """afi_gemm_nt / afi_gemm_tn (the GEMMs every big 3x3 conv runs on) against fp64, including the last tile of the last plane."""
    from afigan_amd import _lib
    lib, dt, st = _lib.load(), _lib.DTYPES[dtype], amd.ops.stream_ptr()
    g = torch.Generator(device="cuda").manual_seed(5)
    for planes, rows, N, K in ((3, 256, 128, 32), (16, 384, 256, 288), (36, 128, 384, 1024)):
        A = torch.randn((planes, rows, K), device="cuda", generator=g)
        B = torch.randn((planes, N, K), device="cuda", generator=g)
        Cm = torch.full((planes, rows, N), float("nan"), device="cuda")
        amd.ops.gemm_nt(A, B, dtype, out=Cm)
        assert _rel(Cm, torch.bmm(A.double(), B.double().transpose(1, 2))) < TOL_GEMM[dtype], (planes, rows, N, K)
    for planes, rows, M, N in ((2, 64, 128, 128), (16, 1056, 256, 384), (36, 4096, 128, 256)):
        Q = torch.randn((planes, rows, M), device="cuda", generator=g)
        V = torch.randn((planes, rows, N), device="cuda", generator=g)
        dU = torch.ones((planes, M, N), device="cuda")                        # += semantics
        amd.ops.gemm_tn(Q, V, dtype, out=dU)
        assert _rel(dU, 1.0 + torch.bmm(Q.double().transpose(1, 2), V.double())) < TOL_GEMM[dtype], (planes, rows, M, N)
    # shapes the tile-aligned kernels do not take are refused, not mangled
    sc = torch.empty(1 << 20, device="cuda", dtype=torch.uint8)
    ptrs = (C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(Cm.data_ptr()))
    assert lib.afi_gemm_nt(*ptrs, 1, 100, 128, 32, dt, C.c_void_p(sc.data_ptr()), sc.numel(), st) == 2
    assert lib.afi_gemm_nt(*ptrs, 1, 128, 128, 32, 5, C.c_void_p(sc.data_ptr()), sc.numel(), st) == 1
    if dtype != "fp32":                                     # the emulated settings need the split-operand scratch: refused without it, never allocated inside
        pieces = 128 * 32 * 2 * {"bf16x6": 3, "bf16x3": 2, "bf16": 1, "f16x3": 2}[dtype]
        assert lib.afi_gemm_nt_scratch_bytes(1, 128, 32, dt) == pieces + (768 if dtype == "f16x3" else 0)       # f16x3: + scales and maxima
        assert lib.afi_gemm_nt(*ptrs, 1, 128, 128, 32, dt, None, 0, st) == 4
        if dtype == "f16x3":
            assert lib.afi_gemm_tn_scratch_bytes(36, dt) == 512
            assert lib.afi_gemm_tn(*ptrs, 1, 128, 128, 128, dt, None, 0, st) == 4
    else:
        assert lib.afi_gemm_nt_scratch_bytes(1, 128, 32, dt) == 0


HARD_KINDS = ["normal", "positive_large_mean", "wide_dynamic_range", "tiny_values", "sparse_post_relu"]


def _hard_operands(kind, g, planes=4, rows=512, N=256, K=1024):
    A = torch.randn((planes, rows, K), device="cuda", generator=g)
    B = torch.randn((planes, N, K), device="cuda", generator=g)
    if kind == "positive_large_mean":
        A, B = A.abs() * 3 + 100.0, B.abs() + 10.0                   # no cancellation to hide a bias of the dropped terms
    elif kind == "wide_dynamic_range":
        A = A * torch.exp2(torch.randint(-20, 20, A.shape, device="cuda", generator=g).float())
        B = B * torch.exp2(torch.randint(-12, 12, B.shape, device="cuda", generator=g).float())
    elif kind == "tiny_values":
        A, B = A * 1e-18, B * 1e-12                                  # products ~1e-30: far below bf16's precision of 1.0, inside its range
    elif kind == "sparse_post_relu":
        A, B = torch.relu(A - 1.0), B * (torch.rand(B.shape, device="cuda", generator=g) < 0.1)
    return A, B


@pytest.mark.parametrize("kind", HARD_KINDS + ["huge_values", "rows_of_different_scale"])
def test_f16x3_is_fp32_grade_on_hard_operands(amd, kind):
    """The two-piece fp16 form (AFI_DTYPE_F16X3) against the fp32 MFMA kernel on the SAME operands, both measured against fp64, NT and TN:
    not worse than fp32's own rounding (bar: 1.5x the fp32 kernel's error + 1e-7), whatever the operands look like -- fp16's range is what
    the per-plane power-of-two scales are there for: values far above 65504, far below 2^-24, 40 binades inside one plane, and (last
    case) rows of one plane whose scales differ by 2^12, where the plane-wide scale leaves the small rows with fewer normal bits in `lo`."""
    g = torch.Generator(device="cuda").manual_seed(11)
    if kind == "huge_values":
        A, B = _hard_operands("normal", g)
        A, B = A * 3e9, B * 7e5                                      # far beyond fp16's largest finite value
    elif kind == "rows_of_different_scale":
        A, B = _hard_operands("normal", g)
        A = A * torch.exp2(torch.randint(-12, 1, (A.shape[0], A.shape[1], 1), device="cuda", generator=g).float())
    else:
        A, B = _hard_operands(kind, g)
    ref = torch.bmm(A.double(), B.double().transpose(1, 2))
    err = {}
    for dt in ("fp32", "f16x3"):
        Cm = amd.ops.gemm_nt(A, B, dt)
        err[dt] = ((Cm.double() - ref).abs().max() / ref.abs().max()).item()
    assert err["f16x3"] <= 1.5 * err["fp32"] + 1e-7, (kind, "NT", err)
    assert err["fp32"] < 5e-6, (kind, err)
    if kind == "rows_of_different_scale":                  # ... and row by row, against each row's own scale
        for dt in ("fp32", "f16x3"):
            Cm = amd.ops.gemm_nt(A, B, dt)
            err[dt] = ((Cm.double() - ref).abs().amax(dim=2) / ref.abs().amax(dim=2)).max().item()
        assert err["f16x3"] <= 1.5 * err["fp32"] + 1e-7, (kind, "NT per row", err)
    # the weight-gradient form on the same data: dU = A^T-like contraction over the rows (K = 512 rows)
    Q, V = A[:, :, :256].contiguous(), B.transpose(1, 2)[:, :512, :].contiguous()        # [planes, 512, 256] x [planes, 512, 256]
    ref = torch.bmm(Q.double().transpose(1, 2), V.double())
    for dt in ("fp32", "f16x3"):
        dU = amd.ops.gemm_tn(Q, V, dt)
        err[dt] = ((dU.double() - ref).abs().max() / ref.abs().max()).item()
    assert err["f16x3"] <= 1.5 * err["fp32"] + 1e-7, (kind, "TN", err)


@pytest.mark.parametrize("kind", HARD_KINDS)
def test_bf16x6_is_fp32_grade_on_hard_operands(amd, kind):
    """The default arithmetic against the fp32 MFMA kernel on the SAME operands, both measured against fp64: the six-product form must
    not be worse than fp32's own rounding (bar: 1.5x the fp32 kernel's error + 1e-7), whatever the operands look like."""
    from afigan_amd import _lib
    lib, st = _lib.load(), amd.ops.stream_ptr()
    g = torch.Generator(device="cuda").manual_seed(11)
    A, B = _hard_operands(kind, g)
    ref = torch.bmm(A.double(), B.double().transpose(1, 2))
    err = {}
    for dt in ("fp32", "bf16x6"):
        Cm = amd.ops.gemm_nt(A, B, dt)
        err[dt] = ((Cm.double() - ref).abs().max() / ref.abs().max()).item()
    assert err["bf16x6"] <= 1.5 * err["fp32"] + 1e-7, (kind, err)
    assert err["fp32"] < 5e-6, (kind, err)


@pytest.mark.parametrize("dtype", ALL_DTYPES)
@pytest.mark.parametrize("N,Ci,Co,H,W", [(1, 256, 256, 50, 68), (2, 256, 512, 100, 84), (1, 288, 128, 33, 47)])
def test_conv3x3_winograd_under_dtype(amd, dtype, N, Ci, Co, H, W):
    """Forward, data gradient and weight gradient of a 3x3 conv on the Winograd path under the context's dtype, against fp64."""
    ops = amd.ops
    g = torch.Generator().manual_seed(6)
    x = torch.randn((N, Ci, H, W), generator=g, dtype=torch.float64)
    w = torch.randn((Co, Ci, 3, 3), generator=g, dtype=torch.float64) / (9 * Ci) ** 0.5
    b = torch.randn(Co, generator=g, dtype=torch.float64)
    dy = torch.randn((N, Co, H, W), generator=g, dtype=torch.float64)
    xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv2d(xg, wg, b, padding=1)
    gx, gw = torch.autograd.grad((y * dy).sum(), (xg, wg))
    xp, dyp = ops.pixel_major(x.float().cuda()), ops.pixel_major(dy.float().cuda())
    wk = ops.ohwi(w.float().cuda())
    with amd.compute_dtype(dtype):
        out = ops.conv3x3_wino_fwd(xp, wk, b.float().cuda())
        dx = ops.conv3x3_wino_dgrad(dyp, wk)
        dw = ops.conv3x3_wino_wgrad(dyp, xp)
    assert amd._lib.current_ctx().dtype == DEFAULT_DTYPE
    assert _rel(out, y) < TOL_CONV[dtype] and _rel(dx, gx) < TOL_CONV[dtype] and _rel(dw, gw) < TOL_CONV[dtype]
    # and the setting is not a no-op: bf16 differs from fp32 by more than fp32's own error
    if dtype == "bf16":
        assert _rel(out, y) > 1e-4


@pytest.mark.parametrize("dtype", ALL_DTYPES)
def test_generator_fwd_bwd_under_dtype(amd, dtype):
    """AF interpolator on a map large enough for the Winograd path (2 x 256 x 52 x 84), forward + full backward, against the oracle."""
    Cc = 256
    G = amd.Generator(in_channels=Cc, n_residual_dense_blocks=3).cuda()
    p = orc.closed_form_generator_params(Cc, 3, 32)
    G.load_state_dict(p)
    g = torch.Generator().manual_seed(7)
    x = torch.randn((2, Cc, 52, 84), generator=g)
    R = torch.randn((2, Cc, 104, 168), generator=g)
    xr = x.clone().requires_grad_(True)
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ref = orc.generator_forward(xr, pr, 3)
    (ref * R).sum().backward()
    xg = x.cuda().requires_grad_(True)
    from afigan_amd import _lib
    seen = []
    ob = lambda name, cx: seen.append((name, _lib.load().afi_ctx_get_compute_dtype(cx.handle)))      # noqa: E731
    _lib._observers.append(ob)
    try:
        with amd.compute_dtype(dtype):
            out = G(xg)
            (out * R.cuda()).sum().backward()
    finally:
        _lib._observers.remove(ob)
    # the backward (on PyTorch's autograd thread) really ran in the requested arithmetic: the library reports it at the call
    assert seen == [("afi_generator_fwd", _lib.DTYPES[dtype]), ("afi_generator_bwd", _lib.DTYPES[dtype])], seen
    tol = TOL_NET[dtype]
    assert _rel(out, ref) < tol and _rel(xg.grad, xr.grad) < tol
    # parameter gradients, L2-relative per tensor: the forward's error (1e-5 of a pre-activation under bf16x3, 4e-3 under bf16) flips
    # LeakyReLU masks near zero, and each flip moves the gradients behind it by a finite amount (measured: fp32 1e-5 .. 3e-3 on the
    # tensors whose gradient cancels to 3e-4 of its terms, bf16x3 1.6e-3 .. 5e-3, bf16 5e-2)
    worst = max(((v.grad.contiguous().cpu().double() - pr[k].grad.double()).norm() / pr[k].grad.double().norm()).item() for k, v in G.named_parameters())
    assert worst < TOL_GRAD_L2[dtype], worst


@pytest.mark.parametrize("dtype", ["bf16x6", "f16x3", "bf16x3", "bf16"])
def test_stage1_step_under_dtype(amd, dtype):
    """One stage-1 G+D step with the engine's dtype against the same step in fp32: losses and the parameter updates."""
    import copy
    torch.manual_seed(3)
    G0 = amd.Generator(n_residual_dense_blocks=3).cuda()
    D0 = amd.Discriminator().cuda()
    g = torch.Generator().manual_seed(9)
    hr = [torch.randn((1, 256, 100, 168), generator=g).cuda(), torch.randn((1, 256, 50, 84), generator=g).cuda()]
    lr = [torch.randn((1, 256, 52, 84), generator=g).cuda(), torch.randn((1, 256, 26, 42), generator=g).cuda()]
    res = {}
    for dt in ("fp32", dtype):
        G, D = copy.deepcopy(G0), copy.deepcopy(D0)
        step = amd.Stage1Step(G, D, base_lr=0.01, warmup_iters=0, dtype=dt)
        assert step.dtype == dt and step.bctx.dtype == dt
        step.run_step(lr, hr)
        res[dt] = (step.metrics(), [q.detach().clone() for q in list(G.parameters()) + list(D.parameters())], [q.detach().clone() for q in list(G0.parameters()) + list(D0.parameters())])
    m32, p32, p0 = res["fp32"]
    m, p, _ = res[dtype]
    tol = TOL_NET[dtype]
    for k in m32:
        assert abs(m[k] - m32[k]) <= tol * abs(m32[k]) + 1e-6, (k, m[k], m32[k])
    # the update each parameter received, over the whole net, in L2 (a D gradient carries LeakyReLU-mask flips between any two
    # evaluations -- tests/test_gpu_d_parity.py -- so the bar is that file's TOL_OWN_FORWARD_L2 on top of the dtype's own)
    num = sum(float(((a - c) - (b - c)).double().square().sum()) for a, b, c in zip(p, p32, p0))
    den = sum(float((b - c).double().square().sum()) for b, c in zip(p32, p0))
    assert (num / den) ** 0.5 < 3e-3 + 2 * tol, (num / den) ** 0.5


def test_dtype_api_guards(amd):
    from afigan_amd import _lib
    cx = _lib.Ctx()
    with pytest.raises(amd.AfiError):
        cx.set_dtype("fp16")
    dflt = _lib.DTYPES[DEFAULT_DTYPE]
    assert cx.dtype == {v: k for k, v in _lib.DTYPES.items()}[dflt]
    assert _lib.load().afi_ctx_set_compute_dtype(cx.handle, 5) == 1 and _lib.load().afi_ctx_get_compute_dtype(cx.handle) == dflt
    cx.set_dtype("bf16x3")
    assert _lib.load().afi_ctx_get_compute_dtype(cx.handle) == 3 and _lib.load().afi_ctx_get_compute_dtype(None) == dflt
    cx.set_dtype("fp32")
    assert _lib.load().afi_ctx_get_compute_dtype(cx.handle) == 0
    assert _lib.load().afi_ctx_set_compute_dtype(None, 0) == 1


@pytest.mark.parametrize("plane", [(0, 0), (0, 5), (3, 3), (5, 1)])
def test_f16x3_plane_bounds_hold_on_the_worst_patch(amd, plane):
    """f16x3 scales every F(4x4) input plane from a BOUND -- the source tensor's largest magnitude times the product of the absolute row sums of
    B^T (csrc/igemm.hip: afi_f16_bound) -- not from a measured maximum.  A 6 x 6 patch of +-1 whose signs follow rows r and c of B^T drives plane
    (r, c) of its tile to exactly that bound (49 x the source maximum for (0, 0)); with a bound too small by 2x the fp16 piece would overflow and the
    conv would come back non-finite or wrong.  Checked against the direct fp32 kernels on the whole map."""
    import numpy as np
    BT = np.array([[1, -1.5, -2, 1.5, 1, 0], [0, -1, 0.5, 2.5, 1, 0], [0, 1, -2.5, 0.5, 1, 0], [0, -2, -1, 2, 1, 0], [0, 0.5, -1, -0.5, 1, 0],
                   [0, 1, -1.5, -2, 1.5, 1]])                # (csrc/winograd.hip, AFI_WINO4_POINTS = 1)
    r, c = plane
    sgn = lambda v: np.where(v < 0, -1.0, 1.0)
    patch = np.outer(sgn(BT[r]), sgn(BT[c]))                 # zeros of B^T count as +1: they do not enter the plane
    Ci, Co, H, W = 128, 128, 96, 96                          # 9216 pixels: the F(4x4) tiling
    g = torch.Generator().manual_seed(7)
    x = (torch.rand((1, Ci, H, W), generator=g) * 2 - 1) * 0.25
    for ty, tx in ((1, 1), (5, 9), (20, 3)):                 # tile (ty, tx) reads rows 4 ty - 1 .. 4 ty + 4
        x[0, :, 4 * ty - 1:4 * ty + 5, 4 * tx - 1:4 * tx + 5] = torch.from_numpy(patch).float()
    w = torch.randn((Co, Ci, 3, 3), generator=g) / (3 * Ci ** 0.5)
    b = torch.randn((Co,), generator=g)
    xc, wc = amd.ops.pixel_major(x.cuda()), amd.ops.ohwi(w.cuda())
    with amd.compute_dtype("f16x3"):
        got = amd.ops.conv3x3_wino_infer(xc, wc, b.cuda())
    cx = amd._lib.current_ctx()
    cx.set_option("winograd", 0)
    try:
        ref = amd.ops.conv3x3_fwd(xc, wc, b.cuda())
    finally:
        cx.set_option("winograd", 1)
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 2e-5, err                                  # (F(4x4) rounding; an overflowed plane gives O(1) or inf)
