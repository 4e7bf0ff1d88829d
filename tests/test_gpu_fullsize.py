"""Parity at BASELINE.json's full sizes (configs[1]: per-GPU batch 2, P2 = 200x336) where the whole-tensor oracle would
take minutes: the dominant kernels (forward, data gradient, weight gradient of D's 512 -> 1024 and 1024 -> 1024 convs at P2 -- the latter
is the heaviest layer of the step, K = 1024 -- in both the direct and the Winograd form, and the generator's 256 -> 256 convs on the
208x336 up-sampled map) are checked

  * against an fp64 CPU evaluation of the defining sums at a few hundred sampled outputs (borders and tile seams included),
  * through size-independent properties: linearity  conv(a*x + b*y) == a*conv(x) + b*conv(y)  and the adjoint identity
    <conv(x), dy> == <x, dgrad(dy)> == <w, wgrad(dy, x)>  (one scalar each, accumulated in fp64).

Bar: 1e-3 relative fp32 (north-star)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from d_parity_util import activation_from_saved  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import afigan_amd
    assert torch.cuda.is_available()
    return afigan_amd


def _sample_positions(rng, N, H, W, n):
    """Corners, edges, the seams of the 8x16 halo patches / 128-pixel linear tiles, plus uniform samples."""
    pos = [(0, 0, 0), (N - 1, H - 1, W - 1), (0, 0, W - 1), (N - 1, H - 1, 0), (0, 7, 15), (0, 8, 16), (N - 1, H - 1, W // 2)]
    for _ in range(n - len(pos)):
        pos.append((int(rng.integers(N)), int(rng.integers(H)), int(rng.integers(W))))
    return pos


@pytest.mark.parametrize("algo", ["direct", "winograd"])
@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 512, 1024, 200, 336), (2, 1024, 1024, 200, 336), (2, 256, 256, 208, 336)])
def test_conv3x3_full_size_sampled_fp64_and_properties(amd, N, Cin, Cout, H, W, algo):
    ops = amd.ops
    rng = np.random.default_rng(0)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = ops.new_pixel_major(N, Cin, H, W, "cuda"); x.normal_(generator=g)
    y2 = ops.new_pixel_major(N, Cin, H, W, "cuda"); y2.normal_(generator=g)
    dy = ops.new_pixel_major(N, Cout, H, W, "cuda"); dy.normal_(generator=g)
    w = ops.new_ohwi(Cout, Cin, 3, 3, "cuda", zero=False); w.normal_(0, 0.02, generator=g)
    b = torch.randn(Cout, device="cuda", generator=g)

    if algo == "winograd":
        out = ops.conv3x3_wino_fwd(x, w, b)
        dx = ops.conv3x3_wino_dgrad(dy, w)
        dw = ops.conv3x3_wino_wgrad(dy, x)
    else:
        out = ops.conv3x3_fwd(x, w, b)
        dx = ops.conv3x3_dgrad(dy, w)
        dw = ops.conv3x3_wgrad(dy, x)
    xc, wc, bc, dyc = x.cpu().double(), w.cpu().double(), b.cpu().double(), dy.cpu().double()   # logical NCHW / OIHW views
    oc, dxc, dwc = out.cpu().double(), dx.cpu().double(), dw.cpu().double()
    xp = torch.nn.functional.pad(xc, (1, 1, 1, 1))
    dyp = torch.nn.functional.pad(dyc, (1, 1, 1, 1))

    # ---- sampled fp64 references ----
    scale_o, scale_dx = oc.abs().max().item(), dxc.abs().max().item()
    for (n, yy, xx) in _sample_positions(rng, N, H, W, 160):
        patch = xp[n, :, yy:yy + 3, xx:xx + 3]                                   # [Cin, 3, 3]
        ref = (wc * patch[None]).sum(dim=(1, 2, 3)) + bc                         # all Cout channels of this pixel
        assert (oc[n, :, yy, xx] - ref).abs().max().item() <= 1e-3 * scale_o, ("fwd", n, yy, xx)
        # dx[n,ci,y,x] = sum_{co,ky,kx} dy[n,co,y+1-ky,x+1-kx] * w[co,ci,ky,kx]
        dpatch = dyp[n, :, yy:yy + 3, xx:xx + 3].flip(1, 2)                      # [Cout, 3, 3], index (ky,kx) -> dy[y+1-ky, x+1-kx]
        refdx = (wc * dpatch[:, None]).sum(dim=(0, 2, 3))
        assert (dxc[n, :, yy, xx] - refdx).abs().max().item() <= 1e-3 * scale_dx, ("dgrad", n, yy, xx)
    scale_dw = dwc.abs().max().item()
    for _ in range(24):                                                          # dW[co,ci,ky,kx] = sum_pix dy[.,co,y,x] * x[.,ci,y+ky-1,x+kx-1]
        co, ci, ky, kx = int(rng.integers(Cout)), int(rng.integers(Cin)), int(rng.integers(3)), int(rng.integers(3))
        ref = (dyc[:, co] * xp[:, ci, ky:ky + H, kx:kx + W]).sum().item()
        assert abs(dwc[co, ci, ky, kx].item() - ref) <= 1e-3 * scale_dw, ("wgrad", co, ci, ky, kx)

    fwd = ops.conv3x3_wino_fwd if algo == "winograd" else ops.conv3x3_fwd
    # ---- linearity of the forward kernel at full size ----
    mix = ops.new_pixel_major(N, Cin, H, W, "cuda")
    torch.add(x * 0.75, y2, alpha=-1.25, out=mix)
    lhs = fwd(mix, w, None)
    rhs = 0.75 * fwd(x, w, None) - 1.25 * fwd(y2, w, None)
    assert ((lhs - rhs).abs().max() / rhs.abs().max()).item() < 1e-3

    # ---- adjoint identities (the three kernels agree with each other on every element, not only on samples) ----
    o0 = fwd(x, w, None)
    s_fwd = (o0.double() * dy.double()).sum().item()
    s_dgrad = (x.double() * dx.double()).sum().item()
    s_wgrad = (w.double() * dw.double()).sum().item()
    cs = (o0.double().norm() * dy.double().norm()).item()                       # Cauchy-Schwarz scale of the three inner products
    assert abs(s_fwd - s_dgrad) <= 1e-5 * cs and abs(s_fwd - s_wgrad) <= 1e-5 * cs


def test_stage1_step_full_size_layer_by_layer_fp64(amd):
    """ONE stage-1 iteration at configs[1]'s full size (batch 2, P2..P6 of 800x1333 images; stage1_trainer.py:336-433) -- what bench.py
    times -- with its largest level checked against fp64 beyond "the losses are finite":

      * the discriminator's D-phase forward on the real and on the fake P2 map (2 x 256 x 200 x 336), layer by layer: for a subset of
        output channels of every conv, the conv output over ALL 134,400 pixels in fp64 (from the saved input activation of that layer),
        its batch mean / 1/sqrt(var + eps) over all pixels, the normalised + LeakyReLU activation, and the logits of the last conv;
      * d_loss_p2 recomputed in fp64 from the logits (both BCE terms);
      * the interpolator's output on the 104 x 168 P2 input at sampled positions: the fp64 oracle on a 48 x 48 input patch around each
        (wider than its receptive field), compared at the patch centre;
      * content_loss_p2 (L1 over the cropped pair) recomputed in fp64 from the two full maps.

    Everything is read back from the engine's own workspaces (afi_discriminator_ws_layout), with the weights as they were BEFORE the step."""
    import ctypes as C
    from afigan_amd import _lib
    from oracle import afigan_oracle as orc
    torch.manual_seed(0)
    G = amd.Generator(n_residual_dense_blocks=3).cuda().train()
    D = amd.Discriminator().cuda().train()
    gw = {k: v.detach().clone() for k, v in G.state_dict().items()}
    dw = {k: v.detach().clone() for k, v in D.state_dict().items()}
    gen = torch.Generator(device="cuda").manual_seed(21)
    hr_shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    lr_shapes = [(104, 168), (52, 84), (26, 42), (13, 21), (7, 11)]
    hrs = [torch.randn((2, 256, h, w), device="cuda", generator=gen).contiguous(memory_format=torch.channels_last) for h, w in hr_shapes]
    lrs = [torch.randn((2, 256, h, w), device="cuda", generator=gen).contiguous(memory_format=torch.channels_last) for h, w in lr_shapes]
    step = amd.Stage1Step(G, D, base_lr=1e-3)
    step.run_step(lrs, hrs)
    torch.cuda.synchronize()
    m = step.metrics()
    assert all(np.isfinite(v) for v in m.values())

    N, H, W = 2, 200, 336
    P = N * H * W
    Fa = (C.c_int * 4)(256, 512, 1024, 1024)
    off = (C.c_longlong * 12)()
    _lib.check(_lib.load().afi_discriminator_ws_layout(Fa, N, H, W, off), "layout")
    written = _lib.load().afi_discriminator_saved_activations(None, Fa, N, H, W)
    tr = step._buf["goutg_ws0"]                              # G(lr_p2): 2 x 256 x 208 x 336, cropped to the hr size below
    rng = np.random.default_rng(5)
    bce = 0.0
    for target, x_in in ((1, hrs[0]), (0, tr[:, :, :H, :W])):
        ws = step._buf[f"d_ws_0_{target}"]
        logits = step._buf[f"d_ws_0_{target}_logits"][:P].view(N, H, W).double()
        a = x_in.permute(0, 2, 3, 1).double()               # [N, H, W, Cin] fp64, the layer's input
        for n in range(3):
            ci, co = (256, 512, 1024)[n], (512, 1024, 1024)[n]
            pre = f"Discriminators.0.{n}.0"
            S = torch.from_numpy(np.sort(rng.choice(co, 6, replace=False))).cuda()
            wS = dw[pre + ".weight"][S].double()             # [6, ci, 3, 3]
            ap = torch.nn.functional.pad(a, (0, 0, 1, 1, 1, 1))
            c64 = dw[pre + ".bias"][S].double().view(1, 1, 1, -1).expand(N, H, W, -1).clone()
            for ky in range(3):
                for kx in range(3):
                    c64 += ap[:, ky:ky + H, kx:kx + W, :] @ wS[:, :, ky, kx].t()
            c_hip = ws[off[n]:off[n] + P * co].view(N, H, W, co)
            mean_hip, invstd_hip = ws[off[6 + n]:off[6 + n] + co], ws[off[9 + n]:off[9 + n] + co]
            if written & (1 << n):
                y_hip = ws[off[3 + n]:off[3 + n] + P * co].view(N, H, W, co)
            else:       # Winograd path: the activation is never written; its readers see the saved conv output through the affine (same fp32 arithmetic)
                y_hip = activation_from_saved(c_hip.permute(0, 3, 1, 2), mean_hip, invstd_hip, dw[pre + ".norm.weight"].float(), dw[pre + ".norm.bias"].float()).permute(0, 2, 3, 1)
            sc = c64.abs().max().item()
            assert (c_hip[..., S].double() - c64).abs().max().item() <= 1e-4 * sc, ("conv", target, n)
            mean64 = c64.mean(dim=(0, 1, 2))
            var64 = c64.var(dim=(0, 1, 2), unbiased=False)
            is64 = torch.rsqrt(var64 + orc.BN_EPS)
            assert (mean_hip[S].double() - mean64).abs().max().item() <= 1e-5 * (mean64.abs().max().item() + var64.sqrt().max().item()), ("mean", target, n)
            assert ((invstd_hip[S].double() - is64).abs() / is64).max().item() <= 1e-5, ("invstd", target, n)
            z64 = (c64 - mean64) * is64 * dw[pre + ".norm.weight"][S].double() + dw[pre + ".norm.bias"][S].double()
            y64 = torch.where(z64 > 0, z64, 0.2 * z64)
            assert (y_hip[..., S].double() - y64).abs().max().item() <= 1e-4 * y64.abs().max().item(), ("act", target, n)
            a = y_hip.double()                               # next layer's input: the saved activation (all channels)
            del ap, c64, z64, y64
        w3 = dw["Discriminators.0.3.0.weight"][0].double()   # [1024, 3, 3]
        ap = torch.nn.functional.pad(a, (0, 0, 1, 1, 1, 1))
        z = dw["Discriminators.0.3.0.bias"].double().view(1, 1, 1).expand(N, H, W).clone()
        for ky in range(3):
            for kx in range(3):
                z += ap[:, ky:ky + H, kx:kx + W, :] @ w3[:, ky, kx]
        assert (logits - z).abs().max().item() <= 1e-4 * z.abs().max().item(), ("logits", target)
        bce += torch.nn.functional.softplus(-logits if target == 1 else logits).mean().item()
        del a, ap, z
    assert abs(m["d_loss_p2"] - bce) <= 1e-5 * abs(bce), (m["d_loss_p2"], bce)

    # ---- the interpolator at sampled positions of the P2 level, and the content loss over the whole cropped pair
    gp64 = {k: v.double().cpu() for k, v in gw.items()}
    lr0 = lrs[0].double().cpu()
    Hl, Wl = 104, 168
    # receptive field of an output pixel: 17 low-res 3x3 convs + the conv-transpose (2) + the hi-res conv (1) = 20 low-res pixels each way
    for (n, cy, cx) in ((0, 0, 0), (1, Hl - 1, Wl - 1), (0, 50, 80)):
        y0, x0 = min(max(cy - 24, 0), Hl - 48), min(max(cx - 24, 0), Wl - 48)
        patch = lr0[n:n + 1, :, y0:y0 + 48, x0:x0 + 48]
        ref = orc.generator_forward(patch, gp64, 3)          # [1, 256, 96, 96]; exact wherever the zero padding of the patch is out of reach
        for dy_ in (0, 1):
            for dx_ in (0, 1):
                oy, ox = 2 * cy + dy_, 2 * cx + dx_
                got = tr[n, :, oy, ox].double().cpu()
                want = ref[0, :, oy - 2 * y0, ox - 2 * x0]
                # (a sample at a map corner sits at the patch corner too: there the patch's zero padding IS the map's)
                assert (got - want).abs().max().item() <= 1e-3 * want.abs().max().item(), ("G", n, oy, ox)
    l1 = (tr[:, :, :H, :W].double() - hrs[0].double()).abs().mean().item()
    assert abs(m["content_loss_p2"] - l1) <= 1e-5 * l1, (m["content_loss_p2"], l1)


def test_discriminator_backward_full_size_sampled_fp64(amd):
    """The P2-level discriminator backward of configs[1] (2 x 256 x 200 x 336, default arithmetic: F(4x4) Winograd data and weight
    gradients on the bf16x6 GEMMs) against fp64 evaluations of the defining sums from the tensors the call itself saved -- the one
    reduction length no whole-tensor oracle comparison reaches (feature_patch_discriminator.py:32-41):

      * dW[o, i, ky, kx] = sum over all 134,400 pixels of  g_n[pix, o] * a_{n-1}[pix + tap, i]  for 64 random (o, i, ky, kx) of each of the
        three 3x3 convs (g_n = d loss / d conv output n, left in the backward scratch; a_{n-1} = the saved input activation), i.e. the
        F(3x3,4x4) weight-gradient GEMM with K = 538,624 tile rows per transform point;
      * dx[n, i, y, x] = sum over (o, ky, kx) of  g_0[n, o, y + 1 - ky, x + 1 - kx] * w_0[o, i, ky, kx]  at 64 positions (corners, tile seams,
        random), all 256 channels each: the F(4x4) data gradient.

    Bar: 1e-4 of the tensor's max-norm."""
    import ctypes as C
    from afigan_amd import _lib, ops
    lib = _lib.load()
    torch.manual_seed(0)
    N, H, W = 2, 200, 336
    P = N * H * W
    F = (256, 512, 1024, 1024)
    D = amd.Discriminator().cuda().train()
    net = D.Discriminators[0]
    gen = torch.Generator(device="cuda").manual_seed(31)
    x = ops.new_pixel_major(N, 256, H, W, "cuda"); x.normal_(generator=gen)
    dl = torch.randn((N, 1, H, W), device="cuda", generator=gen) / P
    params = net._ordered_params()
    prm, _keep = net._param_struct(params)
    grads = [torch.zeros_like(q) for q in params]
    gst, _ = net._param_struct(grads, already_packed=True, grads=True)
    Fa = (C.c_int * 4)(*F)
    nf, nb = lib.afi_discriminator_fwd_ws_floats(Fa, N, H, W), lib.afi_discriminator_bwd_ws_floats(Fa, N, H, W)
    ws, sc = torch.empty(nf, device="cuda"), torch.empty(nb, device="cuda")
    logits = torch.empty((N, 1, H, W), device="cuda")
    dx = ops.new_pixel_major(N, 256, H, W, "cuda")
    st = ops.stream_ptr()
    _lib.call("afi_discriminator_fwd", C.byref(prm), ops.view_of(x), N, H, W, C.c_void_p(logits.data_ptr()), 1, C.c_void_p(ws.data_ptr()), nf, st)
    _lib.call("afi_discriminator_bwd", C.byref(prm), C.byref(gst), ops.view_of(x), N, H, W, C.c_void_p(ws.data_ptr()), C.c_void_p(dl.data_ptr()),
              C.c_void_p(dx.data_ptr()), C.c_void_p(sc.data_ptr()), nb, st)
    torch.cuda.synchronize()
    off = (C.c_longlong * 12)()
    _lib.check(lib.afi_discriminator_ws_layout(Fa, N, H, W, off), "layout")
    written = lib.afi_discriminator_saved_activations(None, Fa, N, H, W)
    rng = np.random.default_rng(7)
    g_off = [0, P * F[1], P * F[1] + P * F[2]]                # DiscBwdWs: one d(conv output) buffer per block, left in place by the call
    wnames = {id(p): k for k, p in D.named_parameters()}
    for n in range(3):
        ci, co = F[n], F[n + 1]
        g_n = sc[g_off[n]:g_off[n] + P * co].view(N, H, W, co)
        if n == 0:
            a_in = x.permute(0, 2, 3, 1)
        elif written & (1 << (n - 1)):
            a_in = ws[off[3 + n - 1]:off[3 + n - 1] + P * ci].view(N, H, W, ci)
        else:       # (the backward's input transform read block n - 1's conv output through its affine: so does this check)
            pre_ = f"Discriminators.0.{n - 1}.0.norm"
            byname = {wnames[id(p)]: p for p in params}
            a_in = activation_from_saved(ws[off[n - 1]:off[n - 1] + P * ci].view(N, H, W, ci).permute(0, 3, 1, 2), ws[off[6 + n - 1]:off[6 + n - 1] + ci],
                                         ws[off[9 + n - 1]:off[9 + n - 1] + ci], byname[pre_ + ".weight"].detach().float(), byname[pre_ + ".bias"].detach().float()).permute(0, 2, 3, 1)
        wparam = [p for p in params if wnames[id(p)] == f"Discriminators.0.{n}.0.weight"][0]
        dW = grads[[id(p) for p in params].index(id(wparam))]                 # logical [co, ci, 3, 3]
        scale = dW.double().abs().max().item()
        assert scale > 0
        worst = 0.0
        for _ in range(64):
            o, i, ky, kx = int(rng.integers(co)), int(rng.integers(ci)), int(rng.integers(3)), int(rng.integers(3))
            gcol = g_n[..., o].double()                                        # [N, H, W]
            acol = torch.nn.functional.pad(a_in[..., i].double(), (1, 1, 1, 1))[:, ky:ky + H, kx:kx + W]
            ref = (gcol * acol).sum().item()
            worst = max(worst, abs(dW[o, i, ky, kx].item() - ref) / scale)
        assert worst <= 1e-4, ("wgrad", n, worst)
    # the F(4x4) data gradient of block 0: every input channel at sampled pixels
    g0 = torch.nn.functional.pad(sc[0:P * F[1]].view(N, H, W, F[1]).double(), (0, 0, 1, 1, 1, 1))
    w0 = [p for p in params if wnames[id(p)] == "Discriminators.0.0.0.weight"][0].detach().double()     # [512, 256, 3, 3]
    scale = dx.double().abs().max().item()
    worst = 0.0
    pos = _sample_positions(rng, N, H, W, 60) + [(0, 3, 3), (0, 4, 4), (1, 199, 335), (1, 196, 332)]
    for (n_, yy, xx) in pos:
        dpatch = g0[n_, yy:yy + 3, xx:xx + 3, :].flip(0, 1)                    # index (ky, kx) -> g0[y + 1 - ky, x + 1 - kx], [3, 3, 512]
        ref = torch.einsum("yxo,oiyx->i", dpatch, w0)
        worst = max(worst, (dx[n_, :, yy, xx].double() - ref).abs().max().item() / scale)
    assert worst <= 1e-4, ("dgrad", worst)
