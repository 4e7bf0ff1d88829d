"""Test infrastructure: the discriminator's forward / backward through the C-ABI next to an fp64 evaluation of the oracle.

Used by tests/test_gpu_d_parity.py and tools/d_parity_probe.py.  Everything a LeakyReLU mask depends on is read back from the forward
workspace (afi_discriminator_ws_layout), and the fp64 forward's saved tensors can be injected into it, so the backward can be checked
with masks that are identical by construction (feature_patch_discriminator.py:35-38)."""
import ctypes as C

import torch
import torch.nn.functional as F

from oracle import afigan_oracle as orc

NAMES = []
for _n in range(3):
    NAMES += [f"Discriminators.0.{_n}.0.weight", f"Discriminators.0.{_n}.0.bias", f"Discriminators.0.{_n}.0.norm.weight", f"Discriminators.0.{_n}.0.norm.bias"]
NAMES += ["Discriminators.0.3.0.weight", "Discriminators.0.3.0.bias"]


def zero_grad_bias(k):
    """bias of a conv that feeds a train-mode BatchNorm: its gradient is exactly zero (rounding noise in the reference)"""
    return k.endswith(".0.bias") and not k.startswith("Discriminators.0.3")


def fwd_ref(x, p, dt):
    """oracle forward in dtype dt, returning every intermediate the HIP backward reads"""
    h = x.to(dt)
    out = {"c": [], "y": [], "mean": [], "var": []}
    for n in range(3):
        pre = f"Discriminators.0.{n}.0"
        c = F.conv2d(h, p[pre + ".weight"].to(dt), p[pre + ".bias"].to(dt), 1, 1)
        y, _, _, mean, var = orc.batchnorm_train(c, p[pre + ".norm.weight"].to(dt), p[pre + ".norm.bias"].to(dt),
                                                 p[pre + ".norm.running_mean"].to(dt), p[pre + ".norm.running_var"].to(dt))
        h = orc.lrelu(y)
        out["c"].append(c); out["y"].append(h); out["mean"].append(mean); out["var"].append(var)
    out["logits"] = F.conv2d(h, p["Discriminators.0.3.0.weight"].to(dt), p["Discriminators.0.3.0.bias"].to(dt), 1, 1)
    return out


def activation_from_saved(c, mean, invstd, gamma, beta, slope=0.2):
    """A block's activation from what the forward keeps: lrelu(((c - mean) * invstd) * gamma + beta), every operation rounded to fp32 on its own
    (csrc/afi_bn.h) -- plain fp32 tensor ops reproduce the library's value bit for bit.  c: [N, C, H, W] (any strides); the vectors [C].
    Where the 3x3 convs run in Winograd form the library never writes the activations of blocks 0 and 1
    (include/afigan_hip.h: afi_discriminator_saved_activations) and every reader goes through this."""
    bc = lambda v: v.view(1, -1, 1, 1)
    z = ((c - bc(mean)) * bc(invstd)) * bc(gamma) + bc(beta)
    return torch.where(z > 0, z, z * slope)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-300)).item()


def l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-300)).item()


class DProbe:
    """One discriminator call at [N,256,H,W] with closed-form weights: fp64 reference (with autograd), fp32 CPU forward, HIP forward."""

    def __init__(self, amd, N, H, W, seed, in_filters=256, options=None):
        """`options`: {name: value} of afi_ctx_set_option for this probe's calls (its own context); None = the library's defaults."""
        from afigan_amd import _lib, ops
        self._lib, self.ops = _lib, ops
        lib = _lib.load()
        self.cx = None
        if options is not None:
            self.cx = _lib.Ctx()
            for k, v in options.items():
                self.cx.set_option(k, v)
        self.N, self.H, self.W, self.Cin = N, H, W, in_filters
        dp = orc.closed_form_discriminator_params(in_filters)
        self.dp = dp
        D = amd.Discriminator(in_filters=in_filters).cuda()
        D.load_state_dict(dp)
        D.train()
        self.net = net = D.Discriminators[0]
        self.D = D
        self.x = x = torch.randn((N, in_filters, H, W), generator=torch.Generator().manual_seed(seed))
        self.R = R = torch.randn((N, 1, H, W), generator=torch.Generator().manual_seed(seed + 100))
        self.p64 = {k: (v.double().clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in dp.items()}
        self.x64 = x.double().clone().requires_grad_(True)
        self.r64 = fwd_ref(self.x64, self.p64, torch.float64)
        (self.r64["logits"] * R.double()).sum().backward()
        with torch.no_grad():
            self.r32 = fwd_ref(x, dp, torch.float32)
        self.xp = ops.pixel_major(x.cuda())
        self.params = net._ordered_params()
        self.prm, self._keep = net._param_struct(self.params)
        self.Fa = (C.c_int * 4)(*net.F)
        self.nf = lib.afi_discriminator_fwd_ws_floats(self.Fa, N, H, W)
        self.nb = lib.afi_discriminator_bwd_ws_floats(self.Fa, N, H, W)
        self.ws = torch.empty(self.nf, device="cuda")
        self.sc = torch.empty(self.nb, device="cuda")
        self.logits = torch.empty((N, 1, H, W), device="cuda")
        self._call("afi_discriminator_fwd", C.byref(self.prm), ops.view_of(self.xp), N, H, W, C.c_void_p(self.logits.data_ptr()), 1,
                   C.c_void_p(self.ws.data_ptr()), self.nf, ops.stream_ptr())
        self.off = (C.c_longlong * 12)()
        _lib.call("afi_discriminator_ws_layout", self.Fa, N, H, W, self.off)
        self.written = lib.afi_discriminator_saved_activations(self.cx.handle if self.cx else None, self.Fa, N, H, W)      # bit n: the forward wrote y[n]
        self.P = N * H * W

    def _call(self, name, *args):
        if self.cx is None:
            return self._lib.call(name, *args)
        with self._lib.use_ctx(self.cx):
            return self._lib.call(name, *args)

    def ws_mat(self, o, ch):
        """[P][ch] block of the workspace as a logical NCHW tensor"""
        return self.ws[o:o + self.P * ch].view(self.N, self.H, self.W, ch).permute(0, 3, 1, 2)

    def saved(self, n):
        """(c, y, mean, invstd) of block n as the forward left them; y is reconstructed from the other three where the library does not write it"""
        ch = self.net.F[n + 1]
        off = self.off
        c, mean, invstd = self.ws_mat(off[n], ch), self.ws[off[6 + n]:off[6 + n] + ch], self.ws[off[9 + n]:off[9 + n] + ch]
        if self.written & (1 << n):
            return (c, self.ws_mat(off[3 + n], ch), mean, invstd)
        pre = f"Discriminators.0.{n}.0.norm"
        return (c, activation_from_saved(c, mean, invstd, self.dp[pre + ".weight"].float().cuda(), self.dp[pre + ".bias"].float().cuda()), mean, invstd)

    def mask_flips(self):
        """(HIP vs fp64, torch-CPU fp32 vs fp64) LeakyReLU mask disagreements per layer"""
        g, c = [], []
        for n in range(3):
            y64 = self.r64["y"][n].detach()
            g.append(int(((self.saved(n)[1].cpu() > 0) != (y64 > 0)).sum()))
            c.append(int(((self.r32["y"][n] > 0) != (y64 > 0)).sum()))
        return g, c

    def inject_fp64_forward(self):
        """overwrite the saved conv outputs, activations and statistics with the fp64 forward's (rounded to fp32): identical masks.
        The backward recomputes a LeakyReLU' mask from the saved conv output as sign(((c - mean) * invstd) * gamma + beta), every operation
        rounded to fp32 on its own (csrc/elementwise.hip: afi_bn_affine); for the handful of elements where that fp32 value lands on the
        other side of zero than the fp64 pre-activation, the injected c is moved by single ulps until it does not (a < 1e-6 relative change
        of one input element), so the masks are the reference's BY CONSTRUCTION.  Returns the number of elements nudged per layer."""
        nudged = []
        for n in range(3):
            pre = f"Discriminators.0.{n}.0.norm"
            c, y, mean, invstd = self.saved(n)
            c32 = self.r64["c"][n].detach().float().cuda()
            m32 = self.r64["mean"][n].float().cuda()
            i32 = torch.rsqrt(self.r64["var"][n] + orc.BN_EPS).float().cuda()
            ga, be = self.dp[pre + ".weight"].float().cuda(), self.dp[pre + ".bias"].float().cuda()
            want = (self.r64["y"][n].detach() > 0).cuda()
            bc = lambda v: v.view(1, -1, 1, 1)
            up = (bc(ga) > 0)                                  # z grows with c where gamma > 0 (invstd > 0)
            count = 0
            for it in range(200):
                z = ((c32 - bc(m32)) * bc(i32)) * bc(ga) + bc(be)
                bad = (z > 0) != want
                nb = int(bad.sum())
                if it == 0:
                    count = nb
                if nb == 0:
                    break
                toward = torch.where(want == up, torch.full_like(c32, float("inf")), torch.full_like(c32, float("-inf")))
                c32 = torch.where(bad, torch.nextafter(c32, toward), c32)
            else:
                raise AssertionError(f"layer {n}: could not make {nb} recomputed masks agree with the fp64 forward")
            nudged.append(count)
            c.copy_(c32)
            if self.written & (1 << n):
                y.copy_(self.r64["y"][n].detach().float().cuda())      # (elsewhere the backward reads c through the affine: the nudged c carries the reference's masks)
            mean.copy_(m32)
            invstd.copy_(i32)
        return nudged

    def backward(self):
        """afi_discriminator_bwd on the workspace as it stands; returns (dx, {name: grad}) as logical tensors"""
        _lib, ops = self._lib, self.ops
        grads = [torch.zeros_like(q) for q in self.params]
        gst, _ = self.net._param_struct(grads, already_packed=True, grads=True)
        dx = ops.new_pixel_major(self.N, self.Cin, self.H, self.W, "cuda")
        dl = self.R.cuda().contiguous()
        self._call("afi_discriminator_bwd", C.byref(self.prm), C.byref(gst), ops.view_of(self.xp), self.N, self.H, self.W, C.c_void_p(self.ws.data_ptr()),
                   C.c_void_p(dl.data_ptr()), C.c_void_p(dx.data_ptr()), C.c_void_p(self.sc.data_ptr()), self.nb, ops.stream_ptr())
        torch.cuda.synchronize()
        return dx, dict(zip(NAMES, grads))

    def errors(self, dx, grads):
        """relative L2 / max-norm of dx and of the worst parameter gradient against fp64 autograd"""
        worst_l2, worst_mx = (l2(dx, self.x64.grad), "dx"), (rel(dx, self.x64.grad), "dx")
        for k, g in grads.items():
            if zero_grad_bias(k):
                continue
            e2, em = l2(g, self.p64[k].grad), rel(g, self.p64[k].grad)
            if e2 > worst_l2[0]:
                worst_l2 = (e2, k)
            if em > worst_mx[0]:
                worst_mx = (em, k)
        return {"dx_l2": l2(dx, self.x64.grad), "dx_max": rel(dx, self.x64.grad), "worst_l2": worst_l2, "worst_max": worst_mx}

    def cpu_fp32_backward_errors(self):
        p32 = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in self.dp.items()}
        x32 = self.x.clone().requires_grad_(True)
        o32 = fwd_ref(x32, p32, torch.float32)
        (o32["logits"] * self.R).sum().backward()
        w = max((l2(p32[k].grad, self.p64[k].grad), k) for k in p32 if p32[k].grad is not None and not zero_grad_bias(k))
        return {"dx_l2": l2(x32.grad, self.x64.grad), "dx_max": rel(x32.grad, self.x64.grad), "worst_l2": w}
