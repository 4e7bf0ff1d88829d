"""Parity at BASELINE.json's full sizes (configs[1]: per-GPU batch 2, P2 = 200x336) where the whole-tensor oracle would
take minutes: the three dominant kernels (forward, data gradient, weight gradient of D's 512 -> 1024 conv at P2 in both the
direct and the Winograd form, and the
generator's 256 -> 256 convs on the 208x336 up-sampled map) are checked

  * against an fp64 CPU evaluation of the defining sums at a few hundred sampled outputs (borders and tile seams included),
  * through size-independent properties: linearity  conv(a*x + b*y) == a*conv(x) + b*conv(y)  and the adjoint identity
    <conv(x), dy> == <x, dgrad(dy)> == <w, wgrad(dy, x)>  (one scalar each, accumulated in fp64).

Bar: 1e-3 relative fp32 (north-star)."""
import numpy as np
import pytest
import torch

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
@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 512, 1024, 200, 336), (2, 256, 256, 208, 336)])
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
