"""AF interpolator (learned x2 feature up-sampler) on MI355X.

Drop-in for ``afigan.modeling.feat_interpol.generator_rdb`` (reference generator_rdb.py:73-130):
same class name, constructor signature, ``Generators`` ModuleList, ``forward(features) -> [N,C,2H,2W]`` and the same
``state_dict`` keys / shapes, so reference checkpoints and the AFI FPN call sites
(fpn_sr.py:65-72,151; pafpn_sr.py:67-74,175; bifpn_sr.py:270-276,539-543) work unchanged.

What differs is only HOW it computes: one call into libafigan_hip.so runs the whole conv stack
(3x3 implicit-GEMM convs on fp32 MFMA, the conv-transpose as a 4-phase conv with a pixel-shuffle store, the RDB
``torch.cat`` replaced by channel slices of one dense buffer, the bilinear skip fused into the last conv's epilogue).
Weights are kept as [Cout,Cin,3,3] tensors whose MEMORY is [Cout][3][3][Cin] (channels_last), which is what the
kernels read; ``load_state_dict`` / ``state_dict`` are unaffected by that.
"""
import ctypes as C
import math

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import GenParams, call


class _ConvParams(nn.Module):
    """Parameter holder standing where the reference has an nn.Conv2d / detectron2 Conv2d (names: weight, bias)."""

    def __init__(self, cin, cout, k=3, bias=True, transposed=False):
        super().__init__()
        self.cin, self.cout, self.k, self.transposed = cin, cout, k, transposed
        if transposed:      # torch ConvTranspose2d layout [Cin, Cout, k, k], plain contiguous
            w = torch.empty(cin, cout, k, k)
        else:               # logical [Cout, Cin, k, k], memory [Cout][k][k][Cin]
            w = torch.empty(cout, k, k, cin).permute(0, 3, 1, 2)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # generator_rdb.py:57-62,110-118: kaiming_normal_ (fan_in, gain sqrt 2) * 0.1, zero bias
        fan_in = self.weight.shape[1] * self.k * self.k
        with torch.no_grad():
            self.weight.normal_(0.0, math.sqrt(2.0 / fan_in))
            self.weight.mul_(0.1)
            if self.bias is not None:
                self.bias.zero_()

    def forward(self, *a, **k):
        raise _lib.AfiError("layers of the AF interpolator are fused into one HIP call; call Generator(...) instead")


class ResidualDenseBlock(nn.Module):
    """Parameter tree of generator_rdb.py:33-62 (conv1..conv4: Sequential(conv, LeakyReLU); conv5: conv)."""

    def __init__(self, in_features, growth_rate, residual_scale, kw=3, stw=1, padw=1):
        super().__init__()
        self.residual_scale = residual_scale
        for k in range(1, 5):
            setattr(self, f"conv{k}", nn.Sequential(_ConvParams(in_features + (k - 1) * growth_rate, growth_rate, kw, bias=False),
                                                    nn.LeakyReLU(negative_slope=0.2, inplace=True)))
        self.conv5 = _ConvParams(in_features + 4 * growth_rate, in_features, kw, bias=False)


class ResidualInResidual(nn.Module):
    """Parameter tree of generator_rdb.py:15-25."""

    def __init__(self, n_residual_dense_blocks, in_features, growth_rate, residual_scale, kw=3, stw=1, padw=1):
        super().__init__()
        self.RDBs = nn.Sequential(*[ResidualDenseBlock(in_features, growth_rate, residual_scale, kw, stw, padw)
                                    for _ in range(n_residual_dense_blocks)])
        self.residual_scale = residual_scale


grad_tap = None     # diagnostics: set to a list and every backward call appends its own weight-gradient contributions (clones) before
                    # autograd sums them over the calls that share the weights -- names the call and the tensor when a sum is off


class _GeneratorFn(torch.autograd.Function):
    """One HIP forward / one HIP backward for the whole interpolator."""

    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, x, gen, *params):
        xp = ops.pixel_major(x.detach())
        N, Cc, H, W = xp.shape
        lib = _lib.load()
        prm, keep = gen._param_struct(params)
        ws_floats = lib.afi_generator_fwd_ws_floats(gen.in_channels, gen.growth_rate, gen.n_residual_dense_blocks, N, H, W)
        ws = ops.new_workspace(ws_floats, x.device)
        out = ops.new_pixel_major(N, Cc, 2 * H, 2 * W, x.device)
        call("afi_generator_fwd", C.byref(prm), ops.view_of(xp), N, H, W, ops.view_of(out), C.c_void_p(ws.data_ptr()), ws_floats,
             ops.stream_ptr())
        ctx.gen = gen
        ctx.shape = (N, H, W)
        ctx.x_needs_grad = x.requires_grad
        ctx.save_for_backward(xp, ws, *keep)
        return out

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, dout):
        gen = ctx.gen
        xp, ws, *weights = ctx.saved_tensors
        N, H, W = ctx.shape
        lib = _lib.load()
        dout = dout if ops.is_dense_pm(dout) else ops.pixel_major(dout.contiguous())
        prm, _ = gen._param_struct(weights, already_packed=True)
        # only the parameters that require a gradient get one (MODEL.AFI_FREEZE, fpn_sr.py:67-69, freezes them all: the
        # library then skips every weight-gradient GEMM); zeros_like keeps the [O][kh][kw][I] memory layout
        need = ctx.needs_input_grad[2:]
        grads = ops.zeros_like_many(weights, need)
        gst, _ = gen._param_struct(grads, already_packed=True)
        dx = ops.new_pixel_major(N, gen.in_channels, H, W, dout.device) if ctx.x_needs_grad else None
        sc_floats = lib.afi_generator_bwd_ws_floats(gen.in_channels, gen.growth_rate, gen.n_residual_dense_blocks, N, H, W)
        scratch = ops.new_workspace(sc_floats, dout.device)
        call("afi_generator_bwd", C.byref(prm), C.byref(gst), ops.view_of(xp), N, H, W, C.c_void_p(ws.data_ptr()),
             C.c_void_p(dout.data_ptr()), C.c_void_p(dx.data_ptr() if dx is not None else None),
             C.c_void_p(scratch.data_ptr()), sc_floats, ops.stream_ptr())
        if grad_tap is not None:
            grad_tap.append({"shape": (N, gen.in_channels, H, W), "ctx": ctx.afi_cx.handle.value, "dtype": ctx.afi_cx.dtype,
                             "grads": [g.clone() if g is not None else None for g in grads]})
        return (dx, None, *grads)


class Generator(nn.Module):
    """AF interpolator; signature of generator_rdb.py:75."""

    def __init__(self, in_channels=256, n_residual_dense_blocks=2, growth_rate=32, residual_scale=0.2, scale=2):
        super().__init__()
        if scale != 2:
            raise _lib.AfiError("the reference hard-wires a x2 ConvTranspose2d(k6,s2,p2); scale must be 2")
        if in_channels % 4 or growth_rate % 4:
            raise _lib.AfiError("in_channels and growth_rate must be multiples of 4 (float4 granularity of the HIP kernels)")
        if not 1 <= n_residual_dense_blocks <= _lib.AFI_MAX_RDB:
            raise _lib.AfiError(f"n_residual_dense_blocks must be in [1, {_lib.AFI_MAX_RDB}]")
        self.in_channels = in_channels
        self.n_residual_dense_blocks = n_residual_dense_blocks
        self.growth_rate = growth_rate
        self.residual_scale = residual_scale
        self.scale = scale
        self.kw, self.padw, self.stw = 3, 1, 1
        Cc = in_channels
        self.Generators = nn.ModuleList()
        first_generator = nn.Sequential(
            nn.Sequential(_ConvParams(Cc, Cc, 3), nn.LeakyReLU(0.2, True)),
            ResidualInResidual(n_residual_dense_blocks, Cc, growth_rate, residual_scale, 3, 1, 1),
            nn.Sequential(_ConvParams(Cc, Cc, 3), nn.LeakyReLU(0.2, True)),
            nn.Sequential(_ConvParams(Cc, Cc, 6, transposed=True), nn.LeakyReLU(0.2, True)),
            nn.Sequential(_ConvParams(Cc, Cc, 3)),
        )
        self.Generators.append(first_generator)

    # ---- parameter plumbing -------------------------------------------------------------------------------------
    def _ordered_params(self):
        g = self.Generators[0]
        ps = [g[0][0].weight, g[0][0].bias]
        for rdb in g[1].RDBs:
            ps += [rdb.conv1[0].weight, rdb.conv2[0].weight, rdb.conv3[0].weight, rdb.conv4[0].weight, rdb.conv5.weight]
        ps += [g[2][0].weight, g[2][0].bias, g[3][0].weight, g[3][0].bias, g[4][0].weight, g[4][0].bias]
        return ps

    def _param_struct(self, tensors, already_packed=False):
        """Fill an afi_gen_params_t from tensors ordered like _ordered_params(); returns (struct, tensors kept alive)."""
        R = self.n_residual_dense_blocks
        keep = []
        it = iter(tensors)

        def nxt(kind):
            t = next(it)
            if t is None:               # gradient not wanted: a NULL pointer makes the library skip that weight gradient
                return None
            if not already_packed:
                t = t.detach()
                t = ops.ohwi(t) if kind == "ohwi" else (t if t.is_contiguous() else ops.keep_alive(t.contiguous()))
            keep.append(t)
            return t.data_ptr()

        s = GenParams()
        s.C, s.G, s.n_rdb, s.residual_scale = self.in_channels, self.growth_rate, R, float(self.residual_scale)
        s.w0, s.b0 = nxt("ohwi"), nxt("flat")
        for r in range(R):
            for k in range(5):
                s.rdb_w[r][k] = nxt("ohwi")
        s.w7, s.b7 = nxt("ohwi"), nxt("flat")
        s.wT, s.bT = nxt("flat"), nxt("flat")
        s.w9, s.b9 = nxt("ohwi"), nxt("flat")
        return s, keep

    def forward(self, features):
        """bilinear_x2(features) + Generators[0](features)   (generator_rdb.py:123-130).

        `features`: [N, C, H, W] fp32 on the GPU, NCHW-contiguous or channels_last.  Returns [N, C, 2H, 2W] in
        channels_last memory format (same values / logical shape as the reference)."""
        ops._check_cuda(features)
        if features.dim() != 4 or features.shape[1] != self.in_channels:
            raise _lib.AfiError(f"expected [N,{self.in_channels},H,W], got {tuple(features.shape)}")
        if features.numel() == 0:
            # empty batch / empty map (detectron2's Conv2d wrapper and torch >= 1.5 convs return an empty output of the right
            # shape): nothing to launch; the zero-weighted parameter sum keeps every parameter in the autograd graph with a
            # zero gradient, as the reference's empty-input path does for DDP
            N, C_, H, W = features.shape
            out = features.new_zeros((N, C_, 2 * H, 2 * W))
            if torch.is_grad_enabled() and (features.requires_grad or any(p.requires_grad for p in self.parameters())):
                out = out + features.sum() * 0.0 + sum(p.reshape(-1)[0] for p in self.parameters() if p.requires_grad) * 0.0
            return out
        return _GeneratorFn.apply(features, self, *self._ordered_params())
