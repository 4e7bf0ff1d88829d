"""BiFPN with the AF interpolator as its up-sampler on MI355X -- SURVEY.md section 8(f) row 4.

Mirrors ``BiFPN_AFIGAN`` of the reference (afigan/modeling/backbone/bifpn_sr.py:203-733).  The reference ships it in an INFERENCE
config (configs/inference/AFI-GAN_cascade_rcnn_swint_BiFPN_ST.yaml); the eval-mode forward below is the fast path, and the training-mode
forward (batch-statistics norms, autograd through every node) is built from the same kernels plus their backward.  Same constructor arguments, the
``srf_module`` attribute, the same state_dict (546 tensors + the interpolator's 23: ``before_bifpn.*``,
``BiFPNLayer_{0..6}_conv{3..6}_up / conv{4..7}_down.{depthwise,pointwise,norm}.*``, ``BiFPNLayer_{l}_p{k}_w{1,2}``), and
``forward(x) -> {"p3".."p7"}`` with every quirk of the hard-wired seven-layer forward (raw fusion weights, first-lateral
skips, zero-padded "static_same" max-pool; see oracle/afigan_oracle.py:bifpn_afigan_forward).

Per BiFPN node, in channels_last on this package's kernels:

    fused = swish(w0 * a + w1 * b (+ w2 * c))          one HBM pass, weights read on the device (afi_fuse_swish_fwd)
    dw    = depthwise3x3(fused)                        one HBM pass                             (afi_dwconv3x3_fwd)
    out   = pointwise1x1(dw) with the eval-mode norm   ONE fp32-MFMA GEMM: the BatchNorm affine is folded into its weights
            folded in                                  and bias                                 (afi_conv1x1_fwd)

plus 28 interpolator forwards (one HIP call each) and the zero-padded 3x3/2 max-pools.  The forward has no host
synchronisation, so it can be captured into a hipGraph (``bench.py`` measures both).

In training mode nothing is folded: each piece is a torch.autograd.Function over its HIP forward and backward
(afi_fuse_swish_bwd, afi_dwconv3x3_fwd with reversed taps + afi_dwconv3x3_wgrad, afi_conv1x1_{dgrad,wgrad}, afi_maxpool3s2_same_bwd,
the interpolator's own backward), and the norms use batch statistics with the reference's eps / momentum (afi_bn_stats_ex,
afi_bn_apply_fwd, afi_bn_bwd).  norm = "SyncBN" (the reference default) under a process group of several ranks takes the batch statistics of
ALL ranks' pixels, as detectron2's / torch's SyncBatchNorm do: one all_gather of (mean, variance, count) per norm in the forward, one all_reduce
of (sum g, sum g xhat) in the backward (_SyncBatchNormTrainFn; afi_bn_bwd_sums / afi_bn_bwd_apply); with one rank it is plain BatchNorm.
"""
import math

import torch
import torch.nn as nn

from . import _lib, ops
from .fpn_sr import ShapeSpec, _dense_pm, _LateralMergeFn
from .config import afi_freeze
from .generator_rdb import Generator

__all__ = ["BiFPN_AFIGAN", "LastLevelP6P7"]

_MOM, _EPS = 0.01, 1e-3                                       # bifpn_sr.py:279-280


class _SeparableConv(nn.Module):
    """Parameter tree of SeparableConv2d (bifpn_layers/wrappers.py:172-199): depthwise 3x3 (no bias), pointwise 1x1 + bias, norm."""

    def __init__(self, cin, cout, norm):
        super().__init__()
        self.depthwise = nn.Conv2d(cin, cin, 3, groups=cin, bias=False)
        self.pointwise = nn.Conv2d(cin, cout, 1)
        self.norm = _make_norm(norm, cout, eps=_EPS, momentum=_MOM)

    def forward(self, *a, **k):
        raise _lib.AfiError("BiFPN nodes run fused inside BiFPN_AFIGAN.forward")


def _make_norm(norm, ch, eps=1e-5, momentum=0.1):
    if norm == "":
        return None
    if norm not in ("BN", "SyncBN"):
        raise _lib.AfiError(f'norm "{norm}" is not supported on the BiFPN path (BN / SyncBN / "")')
    bn = nn.BatchNorm2d(ch, eps=eps, momentum=momentum)      # SyncBN == BN within one process; same state_dict entries
    bn._afi_sync = norm == "SyncBN"                           # several ranks: batch statistics over all of them (_SyncBatchNormTrainFn)
    return bn


def _lateral(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 1), nn.BatchNorm2d(cout, momentum=_MOM, eps=_EPS))


class _ResampleFeature(nn.Module):                            # bifpn_sr.py:745-756
    def __init__(self, cin, cout, norm):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 1)
        nn.init.kaiming_uniform_(self.conv.weight, a=1)
        nn.init.zeros_(self.conv.bias)
        n = _make_norm(norm, cout)
        if n is not None:
            self.norm = n


class LastLevelP6P7(nn.Module):
    """P6 = maxpool(norm(conv1x1(C5))), P7 = maxpool(P6)   (bifpn_sr.py:773-789)."""

    def __init__(self, in_channels, out_channels, norm=""):
        super().__init__()
        self.num_levels = 2
        self.p6 = _ResampleFeature(in_channels, out_channels, norm)


class _BeforeBiFPN(nn.Module):                                # bifpn_sr.py:159-201
    def __init__(self, out_channels, in_channels, top_block):
        super().__init__()
        self.lateral3 = _lateral(in_channels[0], out_channels)
        self.lateral4 = _lateral(in_channels[1], out_channels)
        self.lateral5 = _lateral(in_channels[2], out_channels)
        self.top_block = top_block
        self.p4_skip = _lateral(in_channels[1], out_channels)
        self.p5_skip = _lateral(in_channels[2], out_channels)


class _FuseSwishFn(torch.autograd.Function):
    """swish(w[0]*a + w[1]*b (+ w[2]*c)) with the raw fusion weights (bifpn_sr.py:535-563) and its backward."""

    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, w, a, b, c):
        wd = w.detach().contiguous()
        a, b = _dense_pm(a.detach()), _dense_pm(b.detach())
        c = _dense_pm(c.detach()) if c is not None else None
        ctx.save_for_backward(wd, a, b, c)
        return ops.fuse_swish(wd, a, b, c)

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, dy):
        wd, a, b, c = ctx.saved_tensors
        need = ctx.needs_input_grad
        dw, da, db, dc = ops.fuse_swish_bwd(wd, a, b, c, _dense_pm(dy), need=need)
        return dw, da, db, dc


class _DepthwiseFn(torch.autograd.Function):
    """SeparableConv2d.depthwise (3x3, zero pad 1, no bias); weight [C,1,3,3] as torch keeps it."""

    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, x, w):
        C_ = w.shape[0]
        w9c = w.detach().reshape(C_, 9).t().contiguous()
        x = _dense_pm(x.detach())
        ctx.save_for_backward(x, w9c)
        return ops.dwconv3x3(x, w9c)

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, dy):
        x, w9c = ctx.saved_tensors
        dy = _dense_pm(dy)
        dx = ops.dwconv3x3(dy, w9c.flip(0).contiguous()) if ctx.needs_input_grad[0] else None      # correlation with the reversed taps
        dw = ops.dwconv3x3_wgrad(dy, x).t().reshape(-1, 1, 3, 3) if ctx.needs_input_grad[1] else None
        return dx, dw


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, x):
        out, idx = ops.maxpool3s2_same_idx(ops.pixel_major(x.detach()))
        ctx.save_for_backward(idx)
        ctx.in_hw = tuple(x.shape[-2:])
        return out

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return ops.maxpool3s2_same_bwd(_dense_pm(dy), idx, ctx.in_hw)


class _BatchNormTrainFn(torch.autograd.Function):
    """BatchNorm2d in training mode on a pixel-major tensor: batch statistics (fp64 accumulation), running buffers updated in place."""

    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, x, gamma, beta, bn):
        x = _dense_pm(x.detach())
        N, C_, H, W = x.shape
        x2d = x.permute(0, 2, 3, 1).reshape(N * H * W, C_)
        mom = bn.momentum if bn.momentum is not None else 1.0 / float(int(bn.num_batches_tracked) + 1)
        track = bn.track_running_stats and bn.running_mean is not None
        mean, invstd = ops.bn_stats_ex(x2d, bn.eps, mom, bn.running_mean if track else None, bn.running_var if track else None,
                                       bn.num_batches_tracked if track else None)
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        y = ops.bn_apply(x2d, mean, invstd, g, b)
        ctx.save_for_backward(x, mean, invstd, g)
        return y.view(N, H, W, C_).permute(0, 3, 1, 2)

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, dy):
        x, mean, invstd, g = ctx.saved_tensors
        N, C_, H, W = x.shape
        dy2d = _dense_pm(dy).permute(0, 2, 3, 1).reshape(N * H * W, C_)
        dgamma, dbeta = torch.zeros_like(g), torch.zeros_like(g)
        dx = ops.bn_bwd(dy2d, x.permute(0, 2, 3, 1).reshape(N * H * W, C_), mean, invstd, g, dgamma, dbeta)
        return dx.view(N, H, W, C_).permute(0, 3, 1, 2), dgamma, dbeta, None


class _SyncTotals:
    """Pixel counts of ALL ranks for the maps of one forward, exchanged ONCE: a SyncBN layer needs the global count N of its map on
    the HOST (afi_bn_bwd_apply takes it by value), and reading it back from the device per norm was one host synchronisation per
    BatchNorm of every BiFPN node -- a serialised, uncapturable forward.  ``BiFPN_AFIGAN._forward_train`` opens one of these around its
    seven layers: ONE all_gather of the local counts of its five pyramid levels and ONE read-back; each norm then looks its map's global
    count up by its local count.  A norm run outside such a block (or on a map the block did not announce) falls back to its own read-back."""
    _tls = __import__("threading").local()

    def __init__(self, local_counts, group=None):
        import torch.distributed as dist
        self.group = group
        counts = sorted(set(int(c) for c in local_counts))
        dev = torch.device("cuda", torch.cuda.current_device())
        mine = torch.tensor(counts, device=dev, dtype=torch.int64)
        allc = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
        dist.all_gather(allc, mine, group=group)
        tot = torch.stack(allc).sum(0).tolist()               # the one host read-back of this forward
        self.totals = dict(zip(counts, tot))

    def __enter__(self):
        self.prev = getattr(_SyncTotals._tls, "cur", None)
        _SyncTotals._tls.cur = self
        return self

    def __exit__(self, *exc):
        _SyncTotals._tls.cur = self.prev
        return False

    @staticmethod
    def lookup(local_count, group):
        cur = getattr(_SyncTotals._tls, "cur", None)
        if cur is not None and cur.group is group:
            return cur.totals.get(int(local_count))
        return None


_count_cache = {}


def _count_tensor(device, count):
    """[float(count)] on `device`, made once per (device, count): no host-to-device copy per norm."""
    key = (device, int(count))
    t = _count_cache.get(key)
    if t is None:
        t = _count_cache[key] = torch.tensor([float(count)], device=device)
    return t


class _SyncBatchNormTrainFn(torch.autograd.Function):
    """norm = "SyncBN" (the reference default, bifpn_sr.py:210,279-280: detectron2's NaiveSyncBatchNorm / nn.SyncBatchNorm) in training mode
    under a process group of several ranks: the batch statistics are those of ALL ranks' pixels.  Forward: this rank's mean / biased
    variance / pixel count (afi_bn_stats_ex), ONE all_gather of the [2 C + 1] vector, combined as torch.batch_norm_gather_stats_with_counts
    does (mean = sum n_r mean_r / N, var = sum n_r (var_r + mean_r^2) / N - mean^2), running buffers updated with the global statistics
    (unbiased variance over N).  Backward: this rank's (sum g, sum g xhat) (afi_bn_bwd_sums; dbeta / dgamma stay per rank, as torch keeps them
    for the data-parallel wrapper to average), ONE all_reduce, then dx with the global sums over N pixels (afi_bn_bwd_apply).
    The ranks are those of the layer's process group (``bn._afi_group``, None = the default group; nn.SyncBatchNorm's ``process_group``)."""

    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, x, gamma, beta, bn):
        import torch.distributed as dist
        group = getattr(bn, "_afi_group", None)
        x = _dense_pm(x.detach())
        N, C_, H, W = x.shape
        x2d = x.permute(0, 2, 3, 1).reshape(N * H * W, C_)
        mean_l, _inv_l, var_l = ops.bn_stats_ex(x2d, bn.eps, 0.0, None, None, None, want_var=True)
        mine = torch.cat([mean_l, var_l, _count_tensor(x.device, N * H * W)])
        allv = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
        dist.all_gather(allv, mine, group=group)
        st = torch.stack(allv).double()                      # [world, 2 C + 1]
        cnt = st[:, -1:]
        total = cnt.sum()
        mean = (st[:, :C_] * cnt).sum(0) / total
        var = ((st[:, C_:2 * C_] + st[:, :C_] ** 2) * cnt).sum(0) / total - mean ** 2
        var = var.clamp_min(0.0)
        invstd = torch.rsqrt(var + bn.eps).float()
        mean = mean.float()
        if bn.track_running_stats and bn.running_mean is not None:
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(int(bn.num_batches_tracked) + 1)
            unb = (var * (total / (total - 1).clamp_min(1.0))).float()
            bn.running_mean.mul_(1.0 - mom).add_(mean, alpha=mom)
            bn.running_var.mul_(1.0 - mom).add_(unb, alpha=mom)
            bn.num_batches_tracked.add_(1)
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        y = ops.bn_apply(x2d, mean, invstd, g, b)
        ctx.save_for_backward(x, mean, invstd, g)
        known = _SyncTotals.lookup(N * H * W, group)         # exchanged once per forward by the caller (BiFPN_AFIGAN._forward_train) ...
        ctx.total = int(known) if known is not None else int(total.item())      # ... else this norm's own read-back (one host sync)
        ctx.group = group
        return y.view(N, H, W, C_).permute(0, 3, 1, 2)

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, dy):
        import torch.distributed as dist
        x, mean, invstd, g = ctx.saved_tensors
        N, C_, H, W = x.shape
        dy2d = _dense_pm(dy).permute(0, 2, 3, 1).reshape(N * H * W, C_)
        x2d = x.permute(0, 2, 3, 1).reshape(N * H * W, C_)
        dgamma, dbeta = torch.zeros_like(g), torch.zeros_like(g)
        sums = ops.bn_bwd_sums(dy2d, x2d, mean, invstd, dgamma, dbeta)
        dist.all_reduce(sums, group=ctx.group)
        dx = ops.bn_bwd_apply(dy2d, x2d, mean, invstd, g, sums, ctx.total)
        return dx.view(N, H, W, C_).permute(0, 3, 1, 2), dgamma, dbeta, None


def _sync_active(bn):
    if not (getattr(bn, "_afi_sync", False) and torch.distributed.is_available() and torch.distributed.is_initialized()):
        return False
    return torch.distributed.get_world_size(getattr(bn, "_afi_group", None)) > 1


def _norm_train(x, bn):
    """A norm layer inside the autograd graph: batch statistics when the layer is in training mode (of all ranks' pixels for a "SyncBN" layer
    under a process group of several ranks), its running ones otherwise."""
    if bn is None:
        return x
    if bn.training and _sync_active(bn):
        return _SyncBatchNormTrainFn.apply(x, bn.weight, bn.bias, bn)
    if bn.training:
        return _BatchNormTrainFn.apply(x, bn.weight, bn.bias, bn)
    return torch.nn.functional.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)


def _fold(conv, bn):
    """1x1 conv followed by an eval-mode BatchNorm as one weight / bias pair:  s = gamma / sqrt(var + eps)."""
    w = conv.weight.detach().reshape(conv.weight.shape[0], -1)
    b = conv.bias.detach() if conv.bias is not None else torch.zeros(w.shape[0], device=w.device)
    if bn is None:
        return w.contiguous(), b.contiguous()
    s = bn.weight.detach() * torch.rsqrt(bn.running_var.detach() + bn.eps)
    return (w * s[:, None]).contiguous(), ((b - bn.running_mean.detach()) * s + bn.bias.detach()).contiguous()


class BiFPN_AFIGAN(nn.Module):
    N_LAYERS = 7                                              # hard-wired in the reference (its fpn_repeat argument is unused)

    def __init__(self, bottom_up, in_features, out_channels, fpn_repeat=7, norm="SyncBN", top_block=None, fuse_type="sum", cfg=None):
        super().__init__()
        self._norm = norm
        assert fuse_type in {"avg", "sum"}
        if top_block is None or getattr(top_block, "num_levels", 0) != 2:
            raise _lib.AfiError("BiFPN_AFIGAN needs the two-level top block (LastLevelP6P7), as build_swint_bifpn_sr_backbone passes")
        in_strides = [bottom_up._out_feature_strides[f] for f in in_features]
        in_channels = [bottom_up._out_feature_channels[f] for f in in_features]
        if len(in_features) != 3:
            raise _lib.AfiError("BiFPN_AFIGAN takes three bottom-up features (stage3, stage4, stage5)")
        self.in_features, self.bottom_up, self.cfg = in_features, bottom_up, cfg
        self._out_feature_strides = {f"p{int(math.log2(s))}": s for s in in_strides}
        last_stage = int(math.log2(in_strides[-1]))
        for s in range(last_stage, last_stage + top_block.num_levels):
            in_strides.append(2 ** (s + 1))
            self._out_feature_strides[f"p{s + 1}"] = 2 ** (s + 1)
        for i, s in enumerate(in_strides[1:], 1):
            assert s == 2 * in_strides[i - 1], f"Strides {s} {in_strides[i - 1]} are not log2 contiguous"
        self.before_bifpn = _BeforeBiFPN(out_channels, in_channels, top_block)
        self.srf_module = Generator(in_channels=out_channels, n_residual_dense_blocks=3)       # bifpn_sr.py:270
        if afi_freeze(cfg):
            for p in self.srf_module.parameters():
                p.requires_grad = False
        for l in range(self.N_LAYERS):
            for lv in (6, 5, 4, 3):
                setattr(self, f"BiFPNLayer_{l}_conv{lv}_up", _SeparableConv(out_channels, out_channels, norm))
            for lv in (4, 5, 6, 7):
                setattr(self, f"BiFPNLayer_{l}_conv{lv}_down", _SeparableConv(out_channels, out_channels, norm))
            for lv in (6, 5, 4, 3):
                setattr(self, f"BiFPNLayer_{l}_p{lv}_w1", nn.Parameter(torch.ones(2)))
            for lv, n in ((4, 3), (5, 3), (6, 3), (7, 2)):
                setattr(self, f"BiFPNLayer_{l}_p{lv}_w2", nn.Parameter(torch.ones(n)))
        self._out_features = list(self._out_feature_strides.keys())
        self._out_feature_channels = {k: out_channels for k in self._out_features}
        self._size_divisibility = self._out_feature_strides[self._out_features[-1]]
        self._fuse_type = fuse_type
        self._folded, self._folded_key = None, None

    @property
    def size_divisibility(self):
        return self._size_divisibility

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n]) for n in self._out_features}

    # ------------------------------------------------------------------------------------------------ folded inference weights
    def _fingerprint(self):
        return tuple(t._version for t in self.state_dict(keep_vars=True).values()) + (str(next(self.parameters()).device),)

    def _prepare(self):
        """Eval-mode constants, rebuilt only when a parameter / buffer changed: BatchNorm folded into the 1x1 convs, depthwise
        weights repacked tap-major."""
        key = self._fingerprint()
        if self._folded is not None and self._folded_key == key:
            return self._folded
        f = {}
        bb = self.before_bifpn
        for name in ("lateral3", "lateral4", "lateral5", "p4_skip", "p5_skip"):
            seq = getattr(bb, name)
            f[name] = _fold(seq[0], seq[1])
        p6 = bb.top_block.p6
        f["p6"] = _fold(p6.conv, getattr(p6, "norm", None))
        for l in range(self.N_LAYERS):
            for tag in [f"conv{lv}_up" for lv in (6, 5, 4, 3)] + [f"conv{lv}_down" for lv in (4, 5, 6, 7)]:
                m = getattr(self, f"BiFPNLayer_{l}_{tag}")
                cdw = m.depthwise.weight.shape[0]
                dw = m.depthwise.weight.detach().reshape(cdw, 9).t().contiguous()
                f[(l, tag)] = (dw,) + _fold(m.pointwise, m.norm)
        self._folded, self._folded_key = f, key
        return f

    def set_process_group(self, group):
        """The ranks whose pixels a "SyncBN" layer of this module normalises over (nn.SyncBatchNorm's ``process_group``; None = the default
        group): a model trained under a sub-group must not exchange statistics with the ranks outside it."""
        self.process_group = group
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d) and getattr(m, "_afi_sync", False):
                m._afi_group = group
        return self

    def _check_syncbn(self):
        """norm="SyncBN" (the reference default, bifpn_sr.py:210) exchanges batch statistics between ranks in training mode: the node norms
        (built by _make_norm) do, through _SyncBatchNormTrainFn -- one all_gather in the forward, one all_reduce in the backward per norm.
        Nothing to refuse any more; kept as the one place that says so."""
        return None

    # ------------------------------------------------------------------------------------------------ forward (inference)
    def forward(self, x):
        # the interpolator runs several times on one set of weights: their transformed / packed forms are computed once
        first = next(iter(x.values())) if isinstance(x, dict) else x
        if not first.is_cuda:
            return self._forward_impl(x)
        with ops.weight_transform_cache(first.device):
            return self._forward_impl(x)

    def _forward_train(self, x, bottom_up_features=None):
        """The same seven layers with nothing folded, every piece differentiable (see the module docstring)."""
        bb = self.before_bifpn
        if bottom_up_features is None:
            bottom_up_features = self.bottom_up(x)
        self._check_syncbn()
        c3, c4, c5 = [bottom_up_features[k] for k in self.in_features]       # (the Functions below make their own pixel-major copies)
        group = getattr(self, "process_group", None)
        sync = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d) and m.training and _sync_active(m)]
        if sync and getattr(_SyncTotals._tls, "cur", None) is None:
            # SyncBN over several ranks: every rank's pixel count per pyramid level, exchanged ONCE for the whole forward (p3..p7; the "same"-padded
            # 3x3 / 2 max-pools halve with ceil) -- the 60-odd norms below then run without a host synchronisation each
            n, (h, w) = c5.shape[0], c5.shape[-2:]
            counts = [c3.shape[0] * c3.shape[-2] * c3.shape[-1], c4.shape[0] * c4.shape[-2] * c4.shape[-1], n * h * w]
            for _ in range(2):
                h, w = (h + 1) // 2, (w + 1) // 2
                counts.append(n * h * w)
            with _SyncTotals(counts, group):
                return self._forward_train(x, bottom_up_features)

        def lat(t, seq):
            return _norm_train(_LateralMergeFn.apply(t, seq[0].weight, seq[0].bias, None), seq[1])

        def node(m, w, a, b, c=None):
            y = _DepthwiseFn.apply(_FuseSwishFn.apply(w, a, b, c), m.depthwise.weight)
            return _norm_train(_LateralMergeFn.apply(y, m.pointwise.weight, m.pointwise.bias, None), m.norm)

        c4_skip, c5_skip = lat(c4, bb.p4_skip), lat(c5, bb.p5_skip)
        p6 = bb.top_block.p6
        c6 = _MaxPoolFn.apply(_norm_train(_LateralMergeFn.apply(c5, p6.conv.weight, p6.conv.bias, None), getattr(p6, "norm", None)))
        c7 = _MaxPoolFn.apply(c6)
        lateral = (lat(c3, bb.lateral3), lat(c4, bb.lateral4), lat(c5, bb.lateral5), c6, c7)
        G = self.srf_module
        feats = lateral
        for l in range(self.N_LAYERS):
            p3_in, p4_in, p5_in, p6_in, p7_in = feats
            W = lambda name: getattr(self, f"BiFPNLayer_{l}_{name}")          # noqa: E731
            p6_up = node(W("conv6_up"), W("p6_w1"), p6_in, G(p7_in))
            p5_up = node(W("conv5_up"), W("p5_w1"), p5_in, G(p6_up))
            p4_up = node(W("conv4_up"), W("p4_w1"), p4_in, G(p5_up))
            p3_up = node(W("conv3_up"), W("p3_w1"), p3_in, G(p4_up))
            s4, s5 = (c4_skip, c5_skip) if l == 0 else (lateral[1], lateral[2])
            p4_out = node(W("conv4_down"), W("p4_w2"), s4, p4_up, _MaxPoolFn.apply(p3_up))
            p5_out = node(W("conv5_down"), W("p5_w2"), s5, p5_up, _MaxPoolFn.apply(p4_out))
            p6_out = node(W("conv6_down"), W("p6_w2"), lateral[3], p6_up, _MaxPoolFn.apply(p5_out))
            p7_out = node(W("conv7_down"), W("p7_w2"), lateral[4], _MaxPoolFn.apply(p6_out))
            feats = (p3_up, p4_out, p5_out, p6_out, p7_out)
        return dict(zip(self._out_features, feats))

    def _wants_graph(self, feats):
        """The differentiable path is needed whenever autograd could ask for a gradient: training mode, or grad mode on with a bottom-up
        feature or ANY parameter of this module requiring grad (eval-mode fine-tuning on frozen statistics, a trainable bottom-up whose
        input image does not require grad).  Decided on the bottom-up OUTPUTS, not on the raw input."""
        if self.training:
            return True
        if not torch.is_grad_enabled():
            return False
        return any(t.requires_grad for t in feats) or any(p.requires_grad for p in self.parameters())

    def _forward_impl(self, x):
        bottom_up_features = self.bottom_up(x)
        if self._wants_graph([bottom_up_features[k] for k in self.in_features]):
            return self._forward_train(x, bottom_up_features)
        with torch.no_grad():
            f = self._prepare()
            c3, c4, c5 = [ops.pixel_major(bottom_up_features[k]) for k in self.in_features]

            def lat(t, name):
                w, b = f[name]
                return ops.conv1x1_fwd(t, w, b)

            def node(tag, layer, w, a, b, c=None):
                dw, pw, pb = f[(layer, tag)]
                return ops.conv1x1_fwd(ops.dwconv3x3(ops.fuse_swish(w.detach(), a, b, c), dw), pw, pb)

            c4_skip, c5_skip = lat(c4, "p4_skip"), lat(c5, "p5_skip")
            c6 = ops.maxpool3s2_same(lat(c5, "p6"))
            c7 = ops.maxpool3s2_same(c6)
            lateral = (lat(c3, "lateral3"), lat(c4, "lateral4"), lat(c5, "lateral5"), c6, c7)
            G = self.srf_module
            feats = lateral
            for l in range(self.N_LAYERS):
                p3_in, p4_in, p5_in, p6_in, p7_in = feats
                W = lambda name: getattr(self, f"BiFPNLayer_{l}_{name}")          # noqa: E731
                p6_up = node("conv6_up", l, W("p6_w1"), p6_in, G(p7_in))
                p5_up = node("conv5_up", l, W("p5_w1"), p5_in, G(p6_up))
                p4_up = node("conv4_up", l, W("p4_w1"), p4_in, G(p5_up))
                p3_up = node("conv3_up", l, W("p3_w1"), p3_in, G(p4_up))
                s4, s5 = (c4_skip, c5_skip) if l == 0 else (lateral[1], lateral[2])
                p4_out = node("conv4_down", l, W("p4_w2"), s4, p4_up, ops.maxpool3s2_same(p3_up))
                p5_out = node("conv5_down", l, W("p5_w2"), s5, p5_up, ops.maxpool3s2_same(p4_out))
                p6_out = node("conv6_down", l, W("p6_w2"), lateral[3], p6_up, ops.maxpool3s2_same(p5_out))
                p7_out = node("conv7_down", l, W("p7_w2"), lateral[4], ops.maxpool3s2_same(p6_out))
                feats = (p3_up, p4_out, p5_out, p6_out, p7_out)
            assert len(self._out_features) == len(feats)
            return dict(zip(self._out_features, feats))


def build_swint_bifpn_sr_backbone(cfg, input_shape):
    """bifpn_sr.py:791-817; the Swin bottom-up builder is looked up at call time (registry.bottom_up_builder)."""
    from .registry import bottom_up_builder
    bottom_up = bottom_up_builder("swint")(cfg, input_shape)
    in_features = cfg.MODEL.BIFPN.IN_FEATURES
    cin = bottom_up.output_shape()[in_features[-1]].channels
    return BiFPN_AFIGAN(bottom_up=bottom_up, in_features=in_features, out_channels=cfg.MODEL.BIFPN.OUT_CHANNELS,
                        fpn_repeat=cfg.MODEL.BIFPN.FPN_REPEAT, norm=cfg.MODEL.BIFPN.NORM,
                        top_block=LastLevelP6P7(cin, cfg.MODEL.BIFPN.OUT_CHANNELS, cfg.MODEL.BIFPN.NORM),
                        fuse_type=cfg.MODEL.BIFPN.FUSE_TYPE, cfg=cfg)


def _register():
    from .registry import BACKBONE_REGISTRY
    if "build_swint_bifpn_sr_backbone" not in BACKBONE_REGISTRY:
        BACKBONE_REGISTRY.register(build_swint_bifpn_sr_backbone)
    return True


REGISTERED = _register()
