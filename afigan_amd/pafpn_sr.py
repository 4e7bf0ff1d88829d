"""AFI path-aggregation pyramid (PANet with the AF interpolator in the top-down path) on MI355X -- SURVEY.md section 8(f) row 1.

Mirrors ``PAFPN_AFIGAN`` of the reference (afigan/modeling/backbone/pafpn_sr.py:20-207): same constructor arguments, the
``srf_module`` attribute, ``fpn_lateral{stage}`` / ``pafpn_output{stage}`` / ``pafpn_downsample{stage}`` parameter names and
shapes, ``forward(x) -> {"p2".."p6"}``, ``output_shape()``, ``size_divisibility`` and ``LastLevelMaxPool``.  Everything
runs on this package's kernels in channels_last:

    top-down   prev_k = lateral_1x1(res_k) + bias + srf_module(prev_{k+1})     one fp32-MFMA GEMM, add fused   (:168-177)
    bottom-up  pa_k   = inter_k + relu(conv3x3_stride2(pa_{k-1}) + bias)       ONE stride-2 implicit GEMM: ReLU, the
                                                                               merge (and the /2 of "avg") in its epilogue (:183-188)
               p_k    = output_3x3(pa_k) + bias                                implicit-GEMM conv              (:180,189)

The backward of the stride-2 conv is four parity-phase GEMMs (no zero-stuffed MFMA work) + a strided weight-gradient GEMM.
"""
import math

import torch
import torch.nn as nn

from . import _lib, ops
from .fpn_sr import LastLevelMaxPool, ShapeSpec, _dense_pm, _FpnConv, get_norm
from .config import afi_freeze
from .generator_rdb import Generator

__all__ = ["PAFPN_AFIGAN", "LastLevelMaxPool"]


class _DownsampleMergeFn(torch.autograd.Function):
    """pa = fs * (inter + relu(conv3x3_s2(x, w) + b))  with fs = 1 ("sum") or 0.5 ("avg")   (pafpn_sr.py:183-188)."""

    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, x, w, b, inter, fs):
        xp = ops.pixel_major(x.detach())
        wk = ops.ohwi(w.detach())
        it = ops.pixel_major(inter.detach())
        out, act = ops.conv3x3s2_fwd(xp, wk, b.detach() if b is not None else None, act=2, add=it, add_scale=fs, post_scale=fs,
                                     keep_act=True)
        ctx.save_for_backward(xp, wk, act)
        ctx.has_bias, ctx.fs = b is not None, fs
        return out

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, dy):
        xp, wk, act = ctx.saved_tensors
        dy = _dense_pm(dy)
        need = ctx.needs_input_grad
        dz = ops.relu_bwd(dy, act, scale=ctx.fs)              # gradient at the conv output, through the ReLU and the fuse scale
        dx = ops.conv3x3s2_dgrad(dz, wk, xp.shape[-2:]) if need[0] else None
        dw = ops.conv3x3s2_wgrad(dz, xp) if need[1] else None
        db = ops.bias_grad(dz) if (ctx.has_bias and need[2]) else None
        dinter = None
        if need[3]:
            dinter = dy if ctx.fs == 1.0 else dy * ctx.fs
        return dx, dw, db, dinter, None


class _DownsampleConv(nn.Module):
    """Stands where the reference has detectron2 Conv2d(k3, stride 2, pad 1): weight, c2_xavier_fill; norm == "" -> bias and ONE fused
    kernel (ReLU + merge in the epilogue); another norm -> no bias, the stride-2 conv alone on the HIP kernel, then norm, ReLU and the
    merge as torch ops (pafpn_sr.py:103-117,183-188)."""

    def __init__(self, cin, cout, norm=""):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, 3, 3, cin).permute(0, 3, 1, 2))
        self.norm = get_norm(norm, cout)
        self.bias = nn.Parameter(torch.zeros(cout)) if self.norm is None else None
        nn.init.kaiming_uniform_(self.weight, a=1)

    def forward(self, x, inter, fs):
        if self.norm is None:
            return _DownsampleMergeFn.apply(x, self.weight, self.bias, inter, fs)
        y = _Conv3x3S2Fn.apply(x, self.weight)
        return fs * (inter + torch.relu(self.norm(y)))


class _Conv3x3S2Fn(torch.autograd.Function):
    """y = conv3x3_stride2(x, w) without bias / activation (the norm != "" form of the downsample conv)."""

    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, x, w):
        xp = ops.pixel_major(x.detach())
        wk = ops.ohwi(w.detach())
        out = ops.conv3x3s2_fwd(xp, wk, None, act=0, add=None, add_scale=1.0, post_scale=1.0, keep_act=False)
        ctx.save_for_backward(xp, wk)
        return out

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, dy):
        xp, wk = ctx.saved_tensors
        dy = _dense_pm(dy)
        dx = ops.conv3x3s2_dgrad(dy, wk, xp.shape[-2:]) if ctx.needs_input_grad[0] else None
        dw = ops.conv3x3s2_wgrad(dy, xp) if ctx.needs_input_grad[1] else None
        return dx, dw


class PAFPN_AFIGAN(nn.Module):
    def __init__(self, bottom_up, in_features, out_channels, norm="", top_block=None, fuse_type="sum", cfg=None):
        super().__init__()
        assert fuse_type in {"avg", "sum"}
        self.cfg = cfg
        input_shapes = bottom_up.output_shape()
        in_strides = [input_shapes[f].stride for f in in_features]
        in_channels = [input_shapes[f].channels for f in in_features]
        for i, s in enumerate(in_strides[1:], 1):
            assert s == 2 * in_strides[i - 1], f"Strides {s} {in_strides[i - 1]} are not log2 contiguous"
        self.srf_module = Generator(in_channels=out_channels, n_residual_dense_blocks=3)        # pafpn_sr.py:67
        if afi_freeze(cfg):                          # :69-71
            for p in self.srf_module.parameters():
                p.requires_grad = False
        lateral_convs, output_convs, downsample_convs = [], [], []
        for idx, cin in enumerate(in_channels):
            stage = int(math.log2(in_strides[idx]))
            lat, out = _FpnConv(cin, out_channels, 1, norm), _FpnConv(out_channels, out_channels, 3, norm)
            self.add_module(f"fpn_lateral{stage}", lat)
            self.add_module(f"pafpn_output{stage}", out)
            lateral_convs.append(lat)
            output_convs.append(out)
            if idx > 0:                                                                        # :103-117
                ds = _DownsampleConv(out_channels, out_channels, norm)
                self.add_module(f"pafpn_downsample{stage}", ds)
                downsample_convs.append(ds)
        self.lateral_convs = lateral_convs[::-1]              # top-down order (low to high resolution): 5 4 3 2
        self.output_convs = output_convs                      # 2 3 4 5
        self.downsample_convs = downsample_convs              # 3 4 5
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
        fs = 0.5 if self._fuse_type == "avg" else 1.0
        prev = self.lateral_convs[0](feats[0])
        topdown = [prev]
        for f, lateral in zip(feats[1:], self.lateral_convs[1:]):                              # top-down pathway
            top_down = self.srf_module(prev)
            if top_down.shape[-2:] != f.shape[-2:]:
                raise _lib.AfiError(f"AFI x2 output {tuple(top_down.shape[-2:])} != lateral {tuple(f.shape[-2:])}: pad inputs to "
                                    f"size_divisibility={self._size_divisibility}")
            prev = lateral(f, add=top_down)
            if fs != 1.0:
                prev = prev * fs
            topdown.insert(0, prev)
        pa = topdown[0]
        results = [self.output_convs[0](pa)]
        for inter, ds, output in zip(topdown[1:], self.downsample_convs, self.output_convs[1:]):   # bottom-up augmentation
            if tuple(inter.shape[-2:]) != ((pa.shape[-2] + 1) // 2, (pa.shape[-1] + 1) // 2):
                raise _lib.AfiError(f"stride-2 output of {tuple(pa.shape[-2:])} does not match the next level {tuple(inter.shape[-2:])}")
            pa = ds(pa, inter, fs)
            results.append(output(pa))
        if self.top_block is not None:
            tb_in = bottom_up_features.get(self.top_block.in_feature, None)
            if tb_in is None:
                tb_in = results[self._out_features.index(self.top_block.in_feature)]
            results.extend(self.top_block(tb_in))
        assert len(self._out_features) == len(results)
        return dict(zip(self._out_features, results))

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n]) for n in self._out_features}


def _pafpn_from_cfg(kind):
    def build(cfg, input_shape):
        from .registry import bottom_up_builder
        bottom_up = bottom_up_builder(kind)(cfg, input_shape)
        return PAFPN_AFIGAN(bottom_up=bottom_up, in_features=cfg.MODEL.FPN.IN_FEATURES, out_channels=cfg.MODEL.FPN.OUT_CHANNELS,
                            norm=cfg.MODEL.FPN.NORM, top_block=LastLevelMaxPool(), fuse_type=cfg.MODEL.FPN.FUSE_TYPE, cfg=cfg)
    return build


def _register():
    """The reference's builder names (pafpn_sr.py:237-280) in detectron2's BACKBONE_REGISTRY when it is importable, else in the local one."""
    from .registry import BACKBONE_REGISTRY
    for name, kind in (("build_resnet_pafpn_sr_backbone", "resnet"), ("build_resnest_pafpn_sr_backbone", "resnest")):
        fn = _pafpn_from_cfg(kind)
        fn.__name__ = fn.__qualname__ = name
        if name not in BACKBONE_REGISTRY:
            BACKBONE_REGISTRY.register(fn)
    return True


REGISTERED = _register()
