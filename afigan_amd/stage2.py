"""Adversarial terms of stage-2 training on MI355X -- SURVEY.md section 8(f) row 2.

The reference's stage-2 step (afigan/engine/stage2_trainer.py:279-384) is a detectron2 detector step plus, per pyramid level,
  * a D step:   real = F.interpolate(guide_p, scale_factor=0.5) (nearest), fake = the AFI detector's FPN feature, detached
                (:299-342), with its own optimizer, and
  * G-side losses added to the detector's loss dict:  g_loss_p = 1e-3 * BCE(D(fake).detach(), 1) + L1(fake, real)  (:344-364),
    whose L1 gradient flows back into the detector through autograd.
This module provides exactly those two pieces on the HIP kernels; the detector itself (detectron2 glue) stays where it is.
The nearest x0.5 down-sampling is a strided VIEW of the guide feature (even rows / columns): the kernels take strides, no copy.
"""
import contextlib
import ctypes as C
import os
from typing import Dict, Sequence

import numpy as np
import torch

from . import _lib, ops
from ._lib import call
from .feature_patch_discriminator import Discriminator
from .stage1 import Stage1Step, _FlatOptim, allreduce_sum_, warmup_multistep_lr


def nearest_half(x: torch.Tensor) -> torch.Tensor:
    """F.interpolate(x, scale_factor=0.5) (mode "nearest", stage2_trainer.py:302) as a zero-copy view."""
    H, W = x.shape[2] // 2, x.shape[3] // 2
    return x[:, :, 0:2 * H:2, 0:2 * W:2]


class _L1CropFn(torch.autograd.Function):
    """F.l1_loss(a[..., :h, :w], b[..., :h, :w]) over the common extent (stage2_trainer.py:358 with _reshape_feature), one HIP
    pass producing the loss and d loss / d a."""

    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, a, b):
        ap, bp = ops.pixel_major(a.detach()), ops.pixel_major(b.detach())
        loss = torch.zeros(1, device=a.device, dtype=torch.float32)
        da = ops.l1_crop(ap, bp, loss, want_grad=a.requires_grad)
        ctx.save_for_backward(da) if da is not None else None
        ctx.has = da is not None
        return loss.reshape(())

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, g):
        if not ctx.has:
            return None, None
        (da,) = ctx.saved_tensors
        return da * g, None


def l1_loss_common(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return _L1CropFn.apply(a, b)


@contextlib.contextmanager
def _wino_weight_cache(helper, device):
    """Transformed conv weights shared by the calls of one phase (include/afigan_hip.h: afi_ctx_set_wino_weight_cache), on this engine's
    own contexts (forward / backward, stage1.py); registered for the duration of the phase only, during which D's weights do not change."""
    with _lib.use_ctx(helper.ctx):
        if not getattr(helper, "weight_cache", True):
            yield
            return
        for c_, key in ((helper.ctx, "wino_wcache"), (helper.bctx, "wino_wcache_b")):
            buf = helper._scratch(key, Stage1Step.WINO_WCACHE_FLOATS, device)
            call("afi_ctx_set_wino_weight_cache", c_.handle, C.c_void_p(buf.data_ptr()), Stage1Step.WINO_WCACHE_FLOATS)
        try:
            yield
        finally:
            call("afi_ctx_set_wino_weight_cache", helper.ctx.handle, C.c_void_p(None), 0)
            call("afi_ctx_set_wino_weight_cache", helper.bctx.handle, C.c_void_p(None), 0)


class Stage2Adversarial:
    """D step + generator-side loss terms of one stage-2 iteration for a Discriminator living on this GPU."""

    def __init__(self, D: Discriminator, base_lr: float = 1e-2, momentum: float = 0.9, weight_decay: float = 1e-4,
                 weight_decay_norm: float = 0.0, lr_steps: Sequence[int] = (120000, 160000), lr_gamma: float = 0.1,
                 warmup_factor: float = 1e-3, warmup_iters: int = 1000, first_level: int = 2, process_group=None, dtype=None,
                 overlap_d: bool = True, weight_cache: bool = True, pair_d_max_pixels: int = 40000):
        self.D, self.dnet = D, D.Discriminators[0]
        self.base_lr, self.momentum = base_lr, momentum
        self.sched = (tuple(lr_steps), lr_gamma, warmup_factor, warmup_iters)
        self.first_level = first_level
        self.pg = process_group
        self.distributed = torch.distributed.is_available() and torch.distributed.is_initialized() and \
            torch.distributed.get_world_size(process_group) > 1
        self.world = torch.distributed.get_world_size(process_group) if self.distributed else 1
        names = {id(p): n for n, p in D.named_parameters()}
        self.order = self.dnet._ordered_params()
        self.opt = _FlatOptim([(names[id(p)], p) for p in self.order], weight_decay, weight_decay_norm)
        self._prm, _ = self.dnet._param_struct(self.order)
        self._grad, _ = self.dnet._param_struct([p.grad for p in self.order], already_packed=True, grads=True)
        self._helper = Stage1Step.__new__(Stage1Step)          # reuse the raw D forward/backward plumbing
        self._helper.dnet, self._helper._dprm, self._helper._dgrad = self.dnet, self._prm, self._grad
        self._helper._lib, self._helper._buf = _lib.load(), {}
        self._helper.ctx, self._helper.bctx = _lib.Ctx(dtype), _lib.Ctx(dtype)     # forward / backward stream (stage1.py)
        self._helper.weight_cache = weight_cache
        # levels up to this many pixels (N*h*w): D(real) and D(fake) of a step as ONE call with per-batch BatchNorm statistics (stage1.py)
        self._helper.pair_d_max_pixels = pair_d_max_pixels
        self.overlap_d = overlap_d                               # forwards on the caller's stream, backwards on a second one (stage1.py)
        self.iter = 0
        self.losses = None
        self._names = []
        self._bstream = None

    def set_option(self, name: str, value: int) -> None:
        """A library option (afi_ctx_set_option) on both of the engine's contexts, forward and backward (Stage1Step.set_option)."""
        self._helper.ctx.set_option(name, value)
        self._helper.bctx.set_option(name, value)

    def d_step(self, guide_feats: Sequence[torch.Tensor], fpn_feats: Sequence[torch.Tensor]):
        """stage2_trainer.py:306-342: BCE(D(real),1) + BCE(D(fake.detach()),0) over the levels, backward, D optimizer step."""
        if not self.D.training:
            raise AssertionError("[Stage2Adversarial] D was changed to eval mode!")
        h = self._helper
        dev = fpn_feats[0].device
        names = [f"d_loss_p{self.first_level + i}" for i in range(len(fpn_feats))]
        if self.losses is None or self._names != names:
            self.losses, self._names = torch.zeros(len(names), device=dev), names
        self.losses.zero_()
        self.opt.zero_grad()
        overlap = self.overlap_d
        with _wino_weight_cache(h, dev):                   # D's weights are fixed until the optimizer step below
            try:
                self._d_levels(h, guide_feats, fpn_feats, dev, overlap)
            finally:                                       # also on an error: the second stream's kernels read tensors of the caller's stream
                if self._bstream is not None:
                    torch.cuda.current_stream().wait_stream(self._bstream)
        if self.distributed:
            allreduce_sum_(self.opt.flat_grad, self.pg)
        lr = warmup_multistep_lr(self.base_lr, self.iter, *self.sched)
        self.opt.step(lr, self.momentum, gscale=1.0 / self.world)
        self.iter += 1

    def _d_levels(self, h, guide_feats, fpn_feats, dev, overlap):
        for i, (g, f) in enumerate(zip(guide_feats, fpn_feats)):
            real = ops.pixel_major(nearest_half(ops.pixel_major(g.detach())))
            fake = ops.pixel_major(f.detach())
            hh, ww = min(real.shape[2], fake.shape[2]), min(real.shape[3], fake.shape[3])
            rc, fc = real[:, :, :hh, :ww], fake[:, :, :hh, :ww]
            paired = h._paired(rc)
            calls = ((h._pair(i, rc, fc), (1.0, 0.0)),) if paired else ((rc, (1.0,)), (fc, (0.0,)))      # :306-318 (real, then fake)
            for x, targets in calls:
                key = f"d_ws_{i}_{'p' if paired else int(targets[0])}" if overlap else "d_ws"       # (overlap: a workspace lives until its backward has run)
                logits, dws = h._d_forward(x, key, paired=paired)
                half = rc.shape[0] * hh * ww
                dz = h._scratch("dlogits" + (key if overlap else ""), half * len(targets), dev)
                for k, target in enumerate(targets):
                    call("afi_bce_logits_fwd_bwd", C.c_void_p(logits.data_ptr() + 4 * k * half), half, target, 1.0,
                         C.c_void_p(self.losses.data_ptr() + 4 * i), 1.0, C.c_void_p(dz.data_ptr() + 4 * k * half), ops.stream_ptr())
                if overlap:
                    if self._bstream is None:
                        self._bstream = torch.cuda.Stream(device=dev)
                    self._bstream.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(self._bstream):
                        x.record_stream(self._bstream)
                        h._d_backward(x, dws, dz, paired=paired)
                else:
                    h._d_backward(x, dws, dz, paired=paired)

    def g_losses(self, guide_feats: Sequence[torch.Tensor], fpn_feats: Sequence[torch.Tensor]) -> Dict[str, torch.Tensor]:
        """stage2_trainer.py:344-364: {g_loss_p{lv}: 1e-3 * adv + content}; `content` carries gradient into fpn_feats."""
        h = self._helper
        out = {}
        with _wino_weight_cache(h, fpn_feats[0].device):
            for i, (g, f) in enumerate(zip(guide_feats, fpn_feats)):
                lv = self.first_level + i
                real = nearest_half(ops.pixel_major(g.detach()))
                fake = ops.pixel_major(f.detach())
                hh, ww = min(real.shape[2], fake.shape[2]), min(real.shape[3], fake.shape[3])
                adv = torch.zeros(1, device=f.device)
                fc, rc = fake[:, :, :hh, :ww], real[:, :, :hh, :ww]
                if h._paired(fc):                                                            # fake first, then real (:350-354), one call
                    logits, _ = h._d_forward(h._pair(i, fc, rc), "d_ws", backward_follows=False, paired=True)
                else:
                    logits, _ = h._d_forward(fc, "d_ws", backward_follows=False)
                call("afi_bce_logits_fwd_bwd", C.c_void_p(logits.data_ptr()), f.shape[0] * hh * ww, 1.0, 1.0, C.c_void_p(adv.data_ptr()),
                     0.0, C.c_void_p(None), ops.stream_ptr())
                if not h._paired(fc):
                    h._d_forward(rc, "d_ws", backward_follows=False, stats_only=True)       # only its BN side effects matter (Q2)
                content = l1_loss_common(f, real)
                out[f"g_loss_p{lv}"] = adv.reshape(()) * 1e-3 + content
        return out

    def d_metrics(self, reduce: bool = False) -> Dict[str, float]:
        """D losses of the last d_step; ``reduce=True``: mean over the data-parallel ranks by ONE all-reduce (collective; stage1.py)."""
        vec = self.losses.detach()
        if reduce and self.distributed:
            vec = vec.clone()
            torch.distributed.all_reduce(vec, op=torch.distributed.ReduceOp.SUM, group=self.pg)
            vec = vec / self.world
        return dict(zip(self._names, vec.cpu().tolist()))

    def state_dict(self) -> Dict[str, object]:
        """D's optimizer / scheduler / iteration as the reference checkpoints them beside the network (stage2_trainer.py: D_checkpointer
        with optimizer= and scheduler=); same format as Stage1Step.state_dict's D half."""
        steps, gamma, wf, wi = self.sched
        return {"iteration": int(self.iter), "D_optimizer": {"momentum_buffer": self.opt.state_dict()},
                "scheduler": {"last_epoch": int(self.iter), "base_lr": self.base_lr, "steps": list(steps), "gamma": gamma,
                              "warmup_factor": wf, "warmup_iters": wi}}

    def load_state_dict(self, sd: Dict[str, object]):
        """Inverse of ``state_dict``.  The schedule is a constructor argument: a stored one that differs is refused, not ignored."""
        if self._bstream is not None:
            torch.cuda.current_stream().wait_stream(self._bstream)
        sch = sd.get("scheduler")
        if sch is not None:
            mine = self.state_dict()["scheduler"]
            for k in ("base_lr", "steps", "gamma", "warmup_factor", "warmup_iters"):
                if k in sch and (list(sch[k]) != list(mine[k]) if k == "steps" else abs(float(sch[k]) - float(mine[k])) > 1e-12 * max(1.0, abs(float(mine[k])))):
                    raise ValueError(f"checkpoint scheduler {k} = {sch[k]!r} differs from this engine's {mine[k]!r}")
        self.opt.load_state_dict(sd["D_optimizer"]["momentum_buffer"])
        self.iter = int(sd["iteration"])


class Stage2Step:
    """One stage-2 iteration, ``Multi_Scale_AF_Extractor_Trainer.run_step`` (afigan/engine/stage2_trainer.py:279-384), around the
    HIP-backed adversarial terms:

        hr_ = feature_model(data, img_dict_name='image')                 frozen guide network, full-size image            (:291)
        loss_dict, up_ = model(data)                                     AFI detector on `image_x0.5`; its FPN features   (:296)
        D step on (nearest-half(hr_p), up_p.detach()), D optimizer       Stage2Adversarial.d_step                         (:299-342)
        loss_dict.update(g_loss_p{2..6})                                 Stage2Adversarial.g_losses (D already stepped)   (:344-364)
        optimizer.zero_grad(); sum(loss_dict).backward(); optimizer.step()                                                (:366-384)

    `model` is the detector (``GeneralizedRCNN_AFExtractor`` or anything with its contract: training-mode call returns
    ``(loss_dict, [{"features": {"p2".."p6"}}])``), `feature_model` the guide (``RCNN_FPN_only``: returns ``[{"features": ...}]``),
    `optimizer` any torch optimizer over the detector's parameters.  Data-parallel wrapping of `model` is the caller's (the reference
    wraps it in DistributedDataParallel, :75-84; D's gradients are all-reduced inside Stage2Adversarial)."""

    def __init__(self, model, feature_model, D: Discriminator, optimizer, levels: Sequence[int] = (2, 3, 4, 5, 6), **d_kwargs):
        self.model, self.feature_model, self.optimizer = model, feature_model, optimizer
        self.levels = tuple(levels)
        self.adv = Stage2Adversarial(D, first_level=self.levels[0], **d_kwargs)
        self.last_losses: Dict[str, float] = {}

    def run_step(self, data):
        if not self.model.training:
            raise AssertionError("[Stage2Step] model was changed to eval mode!")                       # :283
        with torch.no_grad():
            hr_ = self.feature_model(data, img_dict_name="image")                                     # :291 (guide is frozen, eval mode)
        loss_dict, up_ = self.model(data)                                                             # :296
        hr = [hr_[0]["features"][f"p{d}"].detach() for d in self.levels]                             # :299-303
        up = [up_[0]["features"][f"p{d}"] for d in self.levels]
        self.adv.d_step(hr, up)                                                                       # :305-342
        d_metrics = self.adv.d_metrics()
        if not all(np.isfinite(v) for v in d_metrics.values()):                                       # _detect_anomaly (:324)
            raise FloatingPointError(f"Loss became infinite or NaN at iteration={self.adv.iter}!\nloss_dict = {d_metrics}")
        loss_dict = dict(loss_dict)
        loss_dict.update(self.adv.g_losses(hr, up))                                                   # :344-364
        losses = sum(loss_dict.values())
        if not torch.isfinite(losses).all():                                                          # :367
            raise FloatingPointError(f"Loss became infinite or NaN at iteration={self.adv.iter}!\nloss_dict = {loss_dict}")
        self.optimizer.zero_grad()                                                                    # :377
        losses.backward()
        self.optimizer.step()                                                                         # :384
        self.last_losses = {**d_metrics, **{k: float(v.detach()) for k, v in loss_dict.items()}}
        return self.last_losses
