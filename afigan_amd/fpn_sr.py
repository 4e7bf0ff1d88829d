"""AFI feature pyramid (top-down merge with the AF interpolator) on MI355X -- SURVEY.md section 8(f) row 1.

Mirrors ``FPN_AFIGAN`` of the reference (afigan/modeling/backbone/fpn_sr.py:20-199): same constructor arguments, the
``srf_module`` attribute (what checkpoint.py:94,120 greps for), ``fpn_lateral{stage}`` / ``fpn_output{stage}`` parameter
names and shapes, ``forward(x) -> {"p2".."p6"}``, ``output_shape()``, ``size_divisibility`` and ``LastLevelMaxPool``.
The merge itself runs on this package's kernels in channels_last throughout (no layout change around G):

    top_down = srf_module(prev)                                  one HIP call (generator_rdb.py)
    prev     = lateral_1x1(feat) + bias + top_down               ONE fp32-MFMA GEMM, the add fused in its epilogue
    p_k      = output_3x3(prev) + bias                           implicit-GEMM conv

``bottom_up`` is any module that returns a dict of feature maps and has ``output_shape()`` (objects with ``.channels`` and
``.stride``); with detectron2 installed the reference's registry names are registered too (see the bottom of this file).
"""
import math
from collections import namedtuple

import torch
import torch.nn as nn

from . import _lib, ops
from .config import afi_freeze
from .generator_rdb import Generator

ShapeSpec = namedtuple("ShapeSpec", ["channels", "stride"])


def _dense_pm(t):
    t = ops.pixel_major(t)
    return t if ops.is_dense_pm(t) else t.contiguous(memory_format=torch.channels_last)


class _LateralMergeFn(torch.autograd.Function):
    """prev = conv1x1(feat, w) + b + top_down   (fpn_sr.py:152-153), add fused into the GEMM epilogue."""

    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, feat, w, b, top_down):
        featp = ops.pixel_major(feat.detach())
        td = ops.pixel_major(top_down.detach()) if top_down is not None else None
        out = ops.conv1x1_fwd(featp, w.detach(), b.detach() if b is not None else None, add=td)
        ctx.save_for_backward(featp, w.detach())
        ctx.has_bias, ctx.has_td = b is not None, top_down is not None
        return out

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, dy):
        featp, w = ctx.saved_tensors
        dy = _dense_pm(dy)
        need = ctx.needs_input_grad
        dfeat = ops.conv1x1_dgrad(dy, w.reshape(w.shape[0], -1)) if need[0] else None
        dw = ops.conv1x1_wgrad(dy, featp).reshape(w.shape) if need[1] else None
        db = ops.bias_grad(dy) if (ctx.has_bias and need[2]) else None
        dtd = dy if (ctx.has_td and need[3]) else None
        return dfeat, dw, db, dtd


def _winograd_pays(x, cout):
    """Large maps with many channels: the Winograd F(2x2,3x3) form of the 3x3 conv (csrc/winograd.hip) beats the direct kernel."""
    n, cin, h, w = x.shape
    return n * h * w >= 4096 and cin >= 128 and cout >= 128


class _Conv3x3Fn(torch.autograd.Function):
    """p = conv3x3(prev, w) + b   (fpn_sr.py:145,158)."""

    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, x, w, b):
        xp = ops.pixel_major(x.detach())
        wk = ops.ohwi(w.detach())
        ctx.wino = _winograd_pays(xp, wk.shape[0])
        fwd = ops.conv3x3_wino_fwd if ctx.wino else ops.conv3x3_fwd
        out = fwd(xp, wk, b.detach() if b is not None else None)
        ctx.save_for_backward(xp, wk)
        ctx.has_bias = b is not None
        return out

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, dy):
        xp, wk = ctx.saved_tensors
        dy = _dense_pm(dy)
        need = ctx.needs_input_grad
        dgrad, wgrad = (ops.conv3x3_wino_dgrad, ops.conv3x3_wino_wgrad) if ctx.wino else (ops.conv3x3_dgrad, ops.conv3x3_wgrad)
        dx = dgrad(dy, wk) if need[0] else None
        dw = wgrad(dy, xp) if need[1] else None
        db = ops.bias_grad(dy) if (ctx.has_bias and need[2]) else None
        return dx, dw, db


class FrozenBatchNorm2d(nn.Module):
    """detectron2.layers.FrozenBatchNorm2d: BatchNorm2d with fixed statistics and affine, all four held as BUFFERS (weight, bias,
    running_mean, running_var; eps 1e-5) -- no parameters, no num_batches_tracked, so reference / detectron2 checkpoints load strictly
    and the module pickles.  A checkpoint written by a plain BatchNorm2d (with num_batches_tracked) loads too."""
    _version = 3

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def forward(self, x):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        bias = self.bias - self.running_mean * scale
        return x * scale.reshape(1, -1, 1, 1).to(x.dtype) + bias.reshape(1, -1, 1, 1).to(x.dtype)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        state_dict.pop(prefix + "num_batches_tracked", None)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def __repr__(self):
        return f"FrozenBatchNorm2d(num_features={self.num_features}, eps={self.eps})"


def get_norm(norm, out_channels):
    """detectron2.layers.get_norm for the strings the reference's configs use (fpn_sr.py:74-81 passes cfg.MODEL.FPN.NORM through):
    "" -> None, "BN", "SyncBN" (torch's nn.SyncBatchNorm: statistics ARE exchanged between ranks on the FPN / PAFPN paths, unlike the BiFPN
    path, which refuses multi-rank SyncBN training -- bifpn_sr.py), "FrozenBN" (fixed statistics and affine as buffers), "GN" (32 groups)."""
    if norm is None or norm == "":
        return None
    if norm == "BN":
        return nn.BatchNorm2d(out_channels)
    if norm == "SyncBN":
        return nn.SyncBatchNorm(out_channels)
    if norm == "GN":
        return nn.GroupNorm(32, out_channels)
    if norm == "FrozenBN":
        return FrozenBatchNorm2d(out_channels)
    raise _lib.AfiError(f'norm "{norm}" is not available without detectron2 (supported: "", "BN", "SyncBN", "FrozenBN", "GN")')


class _FpnConv(nn.Module):
    """Stands where the reference has detectron2 Conv2d: weight [Cout, Cin, k, k], c2_xavier_fill; norm == "" -> bias, fused epilogues;
    any other norm -> no bias and a `.norm` child applied to the conv's output (Conv2d.forward of detectron2: conv, then norm)."""

    def __init__(self, cin, cout, k, norm=""):
        super().__init__()
        self.k = k
        w = torch.empty(cout, k, k, cin).permute(0, 3, 1, 2) if k == 3 else torch.empty(cout, cin, 1, 1)
        self.weight = nn.Parameter(w)
        self.norm = get_norm(norm, cout)
        self.bias = nn.Parameter(torch.zeros(cout)) if self.norm is None else None     # fpn_sr.py:76: use_bias = norm == ""
        nn.init.kaiming_uniform_(self.weight, a=1)           # c2_xavier_fill (fvcore): kaiming_uniform_(a=1), zero bias

    def forward(self, x, add=None):
        if self.norm is not None:                          # conv -> norm (torch) -> top-down add: the add cannot ride in the GEMM epilogue
            y = _LateralMergeFn.apply(x, self.weight, None, None) if self.k == 1 else _Conv3x3Fn.apply(x, self.weight, None)
            y = self.norm(y)
            return y if add is None else y + add
        if self.k == 1:
            return _LateralMergeFn.apply(x, self.weight, self.bias, add)
        assert add is None
        return _Conv3x3Fn.apply(x, self.weight, self.bias)


class LastLevelMaxPool(nn.Module):
    """P6 from P5: max_pool2d(kernel 1, stride 2) == a stride-2 subsample (fpn_sr.py:187-199)."""

    def __init__(self):
        super().__init__()
        self.num_levels = 1
        self.in_feature = "p5"

    def forward(self, x):
        return [x[:, :, ::2, ::2]]


class _Cfg:                                                   # minimal stand-in for cfg.MODEL.AFI_FREEZE when no yacs cfg is given
    class MODEL:
        AFI_FREEZE = False


class FPN_AFIGAN(nn.Module):
    def __init__(self, bottom_up, in_features, out_channels, norm="", top_block=None, fuse_type="sum", cfg=None):
        super().__init__()
        assert fuse_type in {"avg", "sum"}
        self.cfg = cfg
        input_shapes = bottom_up.output_shape()
        in_strides = [input_shapes[f].stride for f in in_features]
        in_channels = [input_shapes[f].channels for f in in_features]
        for i, s in enumerate(in_strides[1:], 1):
            assert s == 2 * in_strides[i - 1], f"Strides {s} {in_strides[i - 1]} are not log2 contiguous"
        self.srf_module = Generator(in_channels=out_channels, n_residual_dense_blocks=3)        # fpn_sr.py:65
        if afi_freeze(cfg):                          # :67-69
            for p in self.srf_module.parameters():
                p.requires_grad = False
        lateral_convs, output_convs = [], []
        for idx, cin in enumerate(in_channels):
            stage = int(math.log2(in_strides[idx]))
            lat, out = _FpnConv(cin, out_channels, 1, norm), _FpnConv(out_channels, out_channels, 3, norm)
            self.add_module(f"fpn_lateral{stage}", lat)
            self.add_module(f"fpn_output{stage}", out)
            lateral_convs.append(lat)
            output_convs.append(out)
        self.lateral_convs = lateral_convs[::-1]              # top-down order (low to high resolution)
        self.output_convs = output_convs[::-1]
        self.top_block = top_block
        self.in_features = in_features
        self.bottom_up = bottom_up
        self._out_feature_strides = {f"p{int(math.log2(s))}": s for s in in_strides}
        if top_block is not None:
            for s in range(stage, stage + top_block.num_levels):
                self._out_feature_strides[f"p{s + 1}"] = 2 ** (s + 1)
        self._out_features = list(self._out_feature_strides.keys())
        self._out_feature_channels = {k: out_channels for k in self._out_features}
        self._size_divisibility = in_strides[-1]
        self._fuse_type = fuse_type

    @property
    def size_divisibility(self):
        return self._size_divisibility

    def forward(self, x):
        # the interpolator runs several times on one set of weights: their transformed / packed forms are computed once
        first = next(iter(x.values())) if isinstance(x, dict) else x
        if not first.is_cuda:
            return self._forward_impl(x)
        with ops.weight_transform_cache(first.device):
            return self._forward_impl(x)

    def _forward_impl(self, x):
        bottom_up_features = self.bottom_up(x)
        feats = [bottom_up_features[f] for f in self.in_features[::-1]]
        results = []
        prev = self.lateral_convs[0](feats[0])
        results.append(self.output_convs[0](prev))
        for f, lateral, output in zip(feats[1:], self.lateral_convs[1:], self.output_convs[1:]):
            top_down = self.srf_module(prev)                  # fpn_sr.py:151
            if top_down.shape[-2:] != f.shape[-2:]:
                raise _lib.AfiError(f"AFI x2 output {tuple(top_down.shape[-2:])} != lateral {tuple(f.shape[-2:])}: pad inputs to "
                                    f"size_divisibility={self._size_divisibility}")
            prev = lateral(f, add=top_down)                   # :152-153 in one GEMM
            if self._fuse_type == "avg":
                prev = prev / 2
            results.insert(0, output(prev))
        if self.top_block is not None:
            tb_in = bottom_up_features.get(self.top_block.in_feature, None)
            if tb_in is None:
                tb_in = results[self._out_features.index(self.top_block.in_feature)]
            results.extend(self.top_block(tb_in))
        assert len(self._out_features) == len(results)
        return dict(zip(self._out_features, results))

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n]) for n in self._out_features}


def _fpn_from_cfg(kind):
    def build(cfg, input_shape):
        from .registry import bottom_up_builder
        bottom_up = bottom_up_builder(kind)(cfg, input_shape)
        return FPN_AFIGAN(bottom_up=bottom_up, in_features=cfg.MODEL.FPN.IN_FEATURES, out_channels=cfg.MODEL.FPN.OUT_CHANNELS,
                          norm=cfg.MODEL.FPN.NORM, top_block=LastLevelMaxPool(), fuse_type=cfg.MODEL.FPN.FUSE_TYPE, cfg=cfg)
    return build


def _register():
    """The reference's builder names (fpn_sr.py:201-244) in detectron2's BACKBONE_REGISTRY when it is importable, else in the local one."""
    from .registry import BACKBONE_REGISTRY
    for name, kind in (("build_resnet_fpn_sr_backbone", "resnet"), ("build_resnest_fpn_sr_backbone", "resnest")):
        fn = _fpn_from_cfg(kind)
        fn.__name__ = fn.__qualname__ = name
        if name not in BACKBONE_REGISTRY:
            BACKBONE_REGISTRY.register(fn)
    return True


REGISTERED = _register()
