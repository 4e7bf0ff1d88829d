"""Feature-patch discriminator (PatchGAN on 256-channel feature maps) on MI355X.

Drop-in for ``afigan.modeling.feat_interpol.feature_patch_discriminator`` (reference lines 16-55): same class name,
``Discriminators`` ModuleList whose element 0 is itself callable (stage1_trainer.py:349-353 calls
``D_model.Discriminators[0](x)`` directly), ``current_step``, and the same ``state_dict`` keys
(``Discriminators.0.{0,1,2}.0.{weight,bias}``, ``....0.norm.{weight,bias,running_mean,running_var,num_batches_tracked}``,
``Discriminators.0.3.0.{weight,bias}``).  BatchNorm is plain per-rank train-mode BN (not SyncBN), as in the reference.
"""
import ctypes as C
import math

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import DiscParams, call


class _BNParams(nn.Module):
    """Stands where the reference has nn.BatchNorm2d (get_norm("BN")): same parameter / buffer names and defaults."""

    def __init__(self, ch):
        super().__init__()
        self.num_features = ch
        self.eps, self.momentum = 1e-5, 0.1
        self.weight = nn.Parameter(torch.ones(ch))
        self.bias = nn.Parameter(torch.zeros(ch))
        self.register_buffer("running_mean", torch.zeros(ch))
        self.register_buffer("running_var", torch.ones(ch))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _ConvBNParams(nn.Module):
    """Stands where the reference has detectron2 Conv2d(norm=BN): weight, bias and an optional `.norm` child."""

    def __init__(self, cin, cout, norm=True):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.weight = nn.Parameter(torch.empty(cout, 3, 3, cin).permute(0, 3, 1, 2))
        self.bias = nn.Parameter(torch.zeros(cout))
        self.norm = _BNParams(cout) if norm else None
        # c2_msra_fill (feature_patch_discriminator.py:43-46): kaiming_normal_(fan_out, relu), zero bias
        with torch.no_grad():
            self.weight.normal_(0.0, math.sqrt(2.0 / (cout * 9)))


class _DiscriminatorFn(torch.autograd.Function):
    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, x, net, *params):
        xp = ops.pixel_major(x.detach())
        N, _, H, W = xp.shape
        lib = _lib.load()
        prm, keep = net._param_struct(params)
        F = (C.c_int * 4)(*net.F)
        # 0 eval; 1 train mode with a backward to come; 2 train-mode statistics only (no input of this call needs a gradient)
        # net._grad_mode: torch.is_grad_enabled() as the module's forward saw it (inside Function.forward it is always off, and
        # needs_input_grad still reports requires_grad under no_grad)
        mode = 0 if not net.training else (1 if (net._grad_mode and any(ctx.needs_input_grad)) else 2)
        ws_floats = lib.afi_discriminator_fwd_ws_floats_ex(_lib.current_ctx().handle, F, N, H, W, mode)     # what this context's call of this kind writes
        ws = ops.new_workspace(ws_floats, x.device)
        logits = torch.empty((N, 1, H, W), device=x.device, dtype=torch.float32)
        call("afi_discriminator_fwd", C.byref(prm), ops.view_of(xp), N, H, W, C.c_void_p(logits.data_ptr()), mode,
             C.c_void_p(ws.data_ptr()), ws_floats, ops.stream_ptr())
        ctx.net, ctx.shape, ctx.x_needs_grad, ctx.was_training = net, (N, H, W), x.requires_grad, net.training
        ctx.save_for_backward(xp, ws, *keep)
        return logits

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, dlogits):
        net = ctx.net
        if not ctx.was_training:
            raise _lib.AfiError("backward through the discriminator is implemented for train-mode BatchNorm only "
                                "(the reference never back-propagates through an eval-mode discriminator)")
        xp, ws, *weights = ctx.saved_tensors
        N, H, W = ctx.shape
        lib = _lib.load()
        prm, _ = net._param_struct(weights, already_packed=True)
        need = ctx.needs_input_grad[2:]
        grads = ops.zeros_like_many(weights, need)
        gst, _ = net._param_struct(grads, already_packed=True, grads=True)
        dlogits = dlogits.contiguous()
        dx = ops.new_pixel_major(N, net.F[0], H, W, dlogits.device) if ctx.x_needs_grad else None
        F = (C.c_int * 4)(*net.F)
        sc_floats = lib.afi_discriminator_bwd_ws_floats(F, N, H, W)
        scratch = ops.new_workspace(sc_floats, dlogits.device)
        call("afi_discriminator_bwd", C.byref(prm), C.byref(gst), ops.view_of(xp), N, H, W, C.c_void_p(ws.data_ptr()),
             C.c_void_p(dlogits.data_ptr()), C.c_void_p(dx.data_ptr() if dx is not None else None),
             C.c_void_p(scratch.data_ptr()), sc_floats, ops.stream_ptr())
        return (dx, None, *grads)


class _PatchDiscriminatorNet(nn.Module):
    """``Discriminators[0]``: children "0".."3" like the reference nn.Sequential, callable on its own."""

    def __init__(self, in_filters=256):
        super().__init__()
        chans = [in_filters]
        f_mult = 1
        for n in range(1, 4):                       # feature_patch_discriminator.py:32-38
            f_mult = min(2 ** n, 4)
            chans.append(in_filters * f_mult)
        self.F = tuple(chans)
        for n in range(3):
            self.add_module(str(n), nn.Sequential(_ConvBNParams(chans[n], chans[n + 1], norm=True), nn.LeakyReLU(0.2, True)))
        self.add_module("3", nn.Sequential(_ConvBNParams(chans[3], 1, norm=False)))     # :40-41

    def __getitem__(self, i):
        return getattr(self, str(i))

    def _ordered_params(self):
        ps = []
        for n in range(3):
            c = self[n][0]
            ps += [c.weight, c.bias, c.norm.weight, c.norm.bias]
        ps += [self[3][0].weight, self[3][0].bias]
        return ps

    def _param_struct(self, tensors, already_packed=False, grads=False):
        keep = []
        it = iter(tensors)

        def nxt(kind):
            t = next(it)
            if t is None:               # gradient not wanted: NULL pointer, the library skips it
                return None
            if not already_packed:
                t = t.detach()
                t = ops.ohwi(t) if kind == "ohwi" else (t if t.is_contiguous() else ops.keep_alive(t.contiguous()))
            keep.append(t)
            return t.data_ptr()

        s = DiscParams()
        for i in range(4):
            s.F[i] = self.F[i]
        for n in range(3):
            s.w[n], s.b[n], s.gamma[n], s.beta[n] = nxt("ohwi"), nxt("flat"), nxt("flat"), nxt("flat")
            if not grads:
                bn = self[n][0].norm
                s.running_mean[n], s.running_var[n] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                s.num_batches_tracked[n] = bn.num_batches_tracked.data_ptr()
        s.w3, s.b3 = nxt("ohwi"), nxt("flat")
        return s, keep

    def forward(self, feature):
        ops._check_cuda(feature)
        if feature.dim() != 4 or feature.shape[1] != self.F[0]:
            raise _lib.AfiError(f"expected [N,{self.F[0]},H,W], got {tuple(feature.shape)}")
        self._grad_mode = torch.is_grad_enabled()
        return _DiscriminatorFn.apply(feature, self, *self._ordered_params())


class Discriminator(nn.Module):
    """Signature of feature_patch_discriminator.py:18 (no arguments; 256 input channels).  `in_filters` is an extension
    used by the small-shape parity tests."""

    def __init__(self, in_filters=256):
        super().__init__()
        self.current_step = 0
        self.kw, self.padw, self.stw = 3, 1, 1
        self.Discriminators = nn.ModuleList([_PatchDiscriminatorNet(in_filters)])

    def forward(self, feature):
        return self.Discriminators[self.current_step](feature)
