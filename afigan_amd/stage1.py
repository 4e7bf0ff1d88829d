"""Stage-1 AFI-GAN G+D step on MI355X: the body of ``AFIGAN_Trainer.run_step`` (reference stage1_trainer.py:305-435).

Per iteration, over the pyramid levels p2..p6 (lr_features -> G -> tr_features, compared with hr_features):
  D phase (:334-381): tr = G(lr).detach(); crop both to the common size (:437-443); logits = D(hr), D(tr) as SEPARATE
      calls (separate BatchNorm batch statistics); d_loss = BCE(logit_real, 1) + BCE(logit_fake, 0); backward;
      D optimizer step.
  G phase (:384-433): tr = G(lr); adv = BCE(D(tr).detach(), 1) (no gradient, Q1); content = L1(tr, hr);
      g_loss = 1e-3*adv + content; backward (L1 only reaches G); G optimizer step.  The two extra D forwards still
      advance D's BatchNorm running statistics (Q2: 4 updates per level per iteration).

MI355X-first differences in HOW (results are the reference's):
  * no autograd graph: the engine calls the C-ABI forward/backward entry points directly on persistent workspaces;
  * each D call is back-propagated right after its forward (gradients accumulate in place), so only one D workspace
    is alive at a time;
  * G's forward of the D phase is reused by the G phase (G's weights do not change in between, Q5) unless
    ``reuse_generator_forward=False``;
  * gradients live in one flat buffer per network (``param.grad`` are views), so data-parallel training is ONE RCCL
    all-reduce per network per iteration over xGMI, followed by the fused multi-tensor SGD kernel.  (The reference
    wraps G/D in DDP but calls ``.module`` so its reducer never fires, Q3; the north-star asks for a real all-reduce.)
  * losses stay on the device; ``metrics()`` reads them back (one sync) instead of two ``.item()`` syncs + two pickle
    gathers per iteration (stage1_trainer.py:459,465).
"""
import contextlib
import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import time

import numpy as np
import torch

from . import _lib, ops
from ._lib import SgdDesc, call
from .feature_patch_discriminator import Discriminator
from .generator_rdb import Generator


def warmup_multistep_lr(base_lr: float, it: int, steps: Sequence[int] = (270000,), gamma: float = 0.1,
                        warmup_factor: float = 1e-3, warmup_iters: int = 1000) -> float:
    """detectron2 WarmupMultiStepLR with linear warm-up (v0.1.1 defaults; configs/step1_afigan_training/*.yaml:16-20)."""
    w = 1.0
    if it < warmup_iters:
        a = it / warmup_iters
        w = warmup_factor * (1 - a) + a
    return base_lr * w * (gamma ** sum(1 for s in steps if it >= s))


def allreduce_sum_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """Sum a flat gradient buffer over the data-parallel ranks in place (backend "nccl" == RCCL on ROCm)."""
    torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM, group=group)
    return flat


def broadcast_module_state_(modules, src: int = 0, group=None) -> None:
    """DistributedDataParallel constructor semantics (stage1_trainer.py:80-89): every rank starts from rank `src`'s
    parameters and buffers."""
    for m in modules:
        for t in m.state_dict().values():
            if t.dim() == 4 and not t.is_contiguous() and t.permute(0, 2, 3, 1).is_contiguous():
                t = t.permute(0, 2, 3, 1)                # [O,I,kh,kw] parameters live as [O][kh][kw][I]: hand RCCL the dense view
            torch.distributed.broadcast(t, src=src, group=group)


class GuidePrefetcher:
    """The two forwards of the FROZEN guide network (stage1_trainer.py:316-327: ``feature_model(data, "image")`` and ``(data, "image_x0.5")``
    under no_grad, eval mode) for batch i + 1, issued on a second stream while batch i trains.  The guide has no trainable state and no
    BatchNorm updates, so its features do not depend on the G / D updates of the iteration they overlap: the result is the same tensors one
    iteration earlier, nothing else.  ``submit(fn, *inputs)`` runs ``fn()`` -- any callable returning the feature tensors (lists / dicts of
    them) -- on the prefetch stream behind whatever the caller's stream has queued (its inputs are ready); ``take()`` makes the caller's
    stream wait for them and hands them over (``record_stream`` keeps the allocator from recycling them while the caller's stream still
    reads them).

    Lifetime contract of what ``fn`` READS: those tensors were allocated on the caller's stream and are read by kernels of the
    prefetch stream, so the caching allocator must not hand their blocks to a new caller-stream allocation before the guide forward is done.
    ``submit`` therefore (1) marks every tensor passed as ``inputs`` and every tensor found in ``fn``'s closure cells / ``functools.partial``
    arguments with ``record_stream(prefetch stream)`` and (2) keeps references to them until ``take()``; a caller whose ``fn`` reaches its
    inputs some other way (a global, an attribute) passes them explicitly.  The caller may drop its own references right after ``submit``.

    Library state: ``fn`` runs under the prefetcher's OWN ``afi_ctx_t`` (``self.ctx``, arithmetic ``dtype``), never under the
    context that happens to be current where ``submit`` is called -- a context serves one stream at a time (include/afigan_hip.h), and
    ``submit`` is typically called from ``Stage1Step.on_d_level``, i.e. while the engine's forward context, with the phase's cache of
    transformed weights registered on it, is the active one.  Guide work placed in that cache would be overwritten by the G phase's refill
    on the step's stream while the prefetch stream still reads it."""

    def __init__(self, device=None, dtype=None):
        # (stream priorities measured: the step's streams on the high priority with this one on the default, 82.9 vs 82.0 ms per step -- the
        #  library's own side streams then starve; nothing to gain)
        self.stream = torch.cuda.Stream(device=device)
        self._pending = None
        with torch.cuda.device(self.stream.device):
            self.ctx = _lib.Ctx(dtype)

    @staticmethod
    def _tensors(o):
        if torch.is_tensor(o):
            yield o
        elif isinstance(o, dict):
            for v in o.values():
                yield from GuidePrefetcher._tensors(v)
        elif isinstance(o, (list, tuple)):
            for v in o:
                yield from GuidePrefetcher._tensors(v)

    @staticmethod
    def _closure_tensors(fn):
        """CUDA tensors a callable holds on to: closure cells, default arguments, functools.partial arguments (lists / tuples / dicts
        are walked; a bound method's instance, module attributes and globals are not: pass such inputs to ``submit`` explicitly)."""
        seen = []
        for cell in getattr(fn, "__closure__", None) or ():
            try:
                seen.extend(GuidePrefetcher._tensors(cell.cell_contents))
            except ValueError:                              # empty cell
                pass
        seen.extend(GuidePrefetcher._tensors(list(getattr(fn, "__defaults__", None) or ())))
        seen.extend(GuidePrefetcher._tensors(dict(getattr(fn, "__kwdefaults__", None) or {})))
        seen.extend(GuidePrefetcher._tensors(list(getattr(fn, "args", ()) or ())))
        seen.extend(GuidePrefetcher._tensors(dict(getattr(fn, "keywords", None) or {})))
        return [t for t in seen if t.is_cuda]

    def submit(self, fn, *inputs):
        if self._pending is not None:
            raise RuntimeError("GuidePrefetcher.submit: the previous batch's features were not taken")
        held = [t for t in self._tensors(list(inputs)) if t.is_cuda] + self._closure_tensors(fn)
        self.stream.wait_stream(torch.cuda.current_stream())
        for t in held:
            t.record_stream(self.stream)                    # read by the prefetch stream: not recyclable before its work is done
        with torch.cuda.stream(self.stream), torch.no_grad(), _lib.use_ctx(self.ctx):
            out = fn()
        self._pending = (out, held)                         # the references live until take()

    def take(self):
        if self._pending is None:
            raise RuntimeError("GuidePrefetcher.take: nothing was submitted")
        (out, _held), self._pending = self._pending, None
        cur = torch.cuda.current_stream()
        cur.wait_stream(self.stream)
        for t in self._tensors(out):
            t.record_stream(cur)
        return out

    @property
    def pending(self):
        return self._pending is not None


class _FlatOptim:
    """Flat gradient + momentum buffers for one network and the device-side descriptor table of the fused SGD kernel
    (torch.optim.SGD semantics as configured by detectron2 build_optimizer: momentum 0.9, weight decay 1e-4, 0 for norm
    parameters; stage1_trainer.py:110-114)."""

    def __init__(self, named_params, weight_decay, weight_decay_norm):
        self.params = [p for _, p in named_params]
        self.names = [n for n, _ in named_params]
        dev = self.params[0].device
        sizes = [p.numel() for p in self.params]
        offs = np.concatenate([[0], np.cumsum([(n + 3) // 4 * 4 for n in sizes])])
        self.total = int(offs[-1])
        self.flat_grad = torch.zeros(self.total, device=dev, dtype=torch.float32)
        self.flat_mom = torch.zeros(self.total, device=dev, dtype=torch.float32)
        self.grad_ptrs = []
        self.weight_decays = [weight_decay_norm if ".norm." in n else weight_decay for n in self.names]
        self._offs = [int(o) for o in offs[:-1]]
        descs = (SgdDesc * len(self.params))()
        for i, ((name, p), n) in enumerate(zip(named_params, sizes)):
            o = int(offs[i])
            seg = self.flat_grad[o:o + n]
            if p.dim() == 4 and not p.is_contiguous():           # [O,I,kh,kw] stored as [O][kh][kw][I]
                assert p.permute(0, 2, 3, 1).is_contiguous(), name
                O, I, kh, kw = p.shape
                p.grad = seg.view(O, kh, kw, I).permute(0, 3, 1, 2)
            else:
                assert p.is_contiguous(), name
                p.grad = seg.view(p.shape)
            self.grad_ptrs.append(self.flat_grad.data_ptr() + 4 * o)
            d = descs[i]
            d.p, d.g, d.m, d.n = p.data_ptr(), self.grad_ptrs[-1], self.flat_mom.data_ptr() + 4 * o, n
            d.wd = weight_decay_norm if ".norm." in name else weight_decay
        raw = np.frombuffer(bytes(descs), dtype=np.uint8).copy()
        self.descs = torch.from_numpy(raw).to(dev)
        self.n = len(self.params)
        self.max_n = max(sizes)

    def zero_grad(self):
        self.flat_grad.zero_()

    def _logical(self, flat: torch.Tensor, i: int) -> torch.Tensor:
        """View of parameter i's segment of a flat buffer in the parameter's LOGICAL shape ([O,I,kh,kw] for conv weights, whatever
        their memory order here)."""
        p, o = self.params[i], self._offs[i]
        seg = flat[o:o + p.numel()]
        if p.dim() == 4 and not p.is_contiguous():
            O, I, kh, kw = p.shape
            return seg.view(O, kh, kw, I).permute(0, 3, 1, 2)
        return seg.view(p.shape)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """torch.optim.SGD's per-parameter state, keyed by parameter NAME: {name: momentum_buffer} in the parameter's logical shape
        (what DetectionCheckpointer stores as optimizer.state[p]["momentum_buffer"], stage1_trainer.py:129-148)."""
        return {n: self._logical(self.flat_mom, i).detach().clone(memory_format=torch.contiguous_format) for i, n in enumerate(self.names)}

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        missing = [n for n in self.names if n not in sd]
        extra = [n for n in sd if n not in set(self.names)]
        if missing or extra:
            raise KeyError(f"optimizer state: missing {missing}, unexpected {extra}")
        for i, n in enumerate(self.names):
            v = sd[n]
            if tuple(v.shape) != tuple(self.params[i].shape):
                raise ValueError(f"optimizer state {n}: shape {tuple(v.shape)} != {tuple(self.params[i].shape)}")
            self._logical(self.flat_mom, i).copy_(v.to(self.flat_mom.device, torch.float32))

    def step(self, lr, momentum, gscale=1.0):
        call("afi_sgd_momentum_step", C.c_void_p(self.descs.data_ptr()), self.n, self.max_n, float(lr), float(momentum), float(gscale),
             ops.stream_ptr())


class Stage1Step:
    """One stage-1 iteration (D step then G step) for a Generator / Discriminator pair living on one GPU."""

    def __init__(self, G: Generator, D: Discriminator, base_lr: float = 1e-3, momentum: float = 0.9, weight_decay: float = 1e-4,
                 weight_decay_norm: float = 0.0, lr_steps: Sequence[int] = (270000,), lr_gamma: float = 0.1,
                 warmup_factor: float = 1e-3, warmup_iters: int = 1000, first_level: int = 2,
                 reuse_generator_forward: bool = True, process_group=None, distributed: Optional[bool] = None, dtype: Optional[str] = None,
                 overlap_d: bool = True, overlap_g: bool = True, weight_cache: bool = True, wgrad_accum: bool = True,
                 g_bwd_small_first: bool = True, overlap_comm: Optional[bool] = None, pair_d_max_pixels: int = 40000):
        self.G, self.D = G, D
        self.gnet, self.dnet = G, D.Discriminators[0]
        self.base_lr, self.momentum = base_lr, momentum
        self.lr_steps, self.lr_gamma, self.warmup_factor, self.warmup_iters = tuple(lr_steps), lr_gamma, warmup_factor, warmup_iters
        self.first_level = first_level
        self.reuse_g = reuse_generator_forward
        # D phase on two streams (forwards in order on the caller's, backwards in order on a second one), and in the G phase G's backward
        # beside the two D forwards per level (the adversarial term carries no gradient): -2.7 % and -2 % of a step now that the big GEMMs
        # are power-bound and leave room beside the bandwidth-bound passes (129.5 -> 126.4 -> 123.9 ms); `overlap_d` / `overlap_g` (attributes too)
        self.overlap_d, self.overlap_g = overlap_d, overlap_g
        self.g_bwd_small_first = g_bwd_small_first          # G-phase backward passes on the second stream: smallest level first (see _run_phases)
        # levels whose cropped map has at most this many pixels (N*h*w) run their two D calls of a phase as ONE call over both batches with
        # per-batch BatchNorm statistics (afi_discriminator_fwd_paired): half the launches of the latency-bound small levels; 0 = never.  Same-box
        # A/B at the reference sizes: 0 -> 83.6 ms, 3000 -> 82.3, 10000 -> 82.2, 40000 (levels p3..p6) -> 81.9, every level -> 84.5 ms per step
        self.pair_d_max_pixels = pair_d_max_pixels
        # data-parallel runs: the two gradient all-reduces are issued asynchronously and run beside work that does not need them -- D's
        # beside G's five backward passes (second stream), G's beside the G phase's D forwards (see _run_phases); False = blocking, in place
        # Default (None): OFF, under every backend.  Both placements are exercised (a one-rank RCCL communicator, two-rank gloo groups:
        # tests/test_gpu_stage1.py)
        # and both are measured wherever more than one rank runs (bench.py: comm.overlap_ab) -- the measurements that exist say blocking: two ranks on one
        # GPU under gloo 186 against 213 ms per step (its wait blocks the HOST, which then cannot queue the G phase), a one-rank RCCL group 78.4 against
        # 80.9 ms; no multi-GPU RCCL measurement exists.  Blocking costs at most the exchange itself (92.7 MB: ~1 ms per step over xGMI)
        self.overlap_comm = overlap_comm
        self._pending_work = []                             # collectives issued and not yet waited for (an error path waits for them: see run_step)
        self.comm_exposed_ms = None                         # measure_comm = True: [D, G] time the consuming stream waited for the collective, last step
        self.measure_comm = False
        # per-phase cache of transformed weights / transform-domain sum of the weight gradients of a phase (pure re-orderings; off = per call)
        self.weight_cache, self.wgrad_accum = weight_cache, wgrad_accum
        self._bstream = None
        # callable(level index) or None, called while the step is being ENQUEUED, before level i of the D phase (largest level first): the place to
        # queue independent work of the caller's (the next batch's guide forwards: GuidePrefetcher.submit) beside the small levels, whose
        # latency-bound chains leave most of the chip idle, instead of beside the chip-filling GEMMs of the large ones
        self.on_d_level = None
        self.iter = 0
        self.pg = process_group
        if distributed is None:
            distributed = torch.distributed.is_available() and torch.distributed.is_initialized() and \
                torch.distributed.get_world_size(process_group) > 1
        self.distributed = distributed
        self.world = torch.distributed.get_world_size(process_group) if distributed else 1
        self.backend = torch.distributed.get_backend(process_group) if distributed else None
        if self.overlap_comm is None:
            self.overlap_comm = False
        for p in list(G.parameters()) + list(D.parameters()):
            ops._check_cuda(p)
        if self.distributed:        # DistributedDataParallel(...) ctor semantics: rank 0's weights everywhere (:80-89)
            broadcast_module_state_([G, D], 0, process_group)
        gnames = dict((id(p), n) for n, p in G.named_parameters())
        dnames = dict((id(p), n) for n, p in D.named_parameters())
        self.g_order = G._ordered_params()
        self.d_order = self.dnet._ordered_params()
        self.g_opt = _FlatOptim([(gnames[id(p)], p) for p in self.g_order], weight_decay, weight_decay_norm)
        self.d_opt = _FlatOptim([(dnames[id(p)], p) for p in self.d_order], weight_decay, weight_decay_norm)
        self._lib = _lib.load()
        self._gprm, self._gkeep = G._param_struct(self.g_order)
        self._ggrad, _ = G._param_struct([p.grad for p in self.g_order], already_packed=True)
        self._dprm, self._dkeep = self.dnet._param_struct(self.d_order)
        self._dgrad, _ = self.dnet._param_struct([p.grad for p in self.d_order], already_packed=True, grads=True)
        for p, k in list(zip(self.g_order, self._gkeep)) + list(zip(self.d_order, self._dkeep)):
            if p.data_ptr() != k.data_ptr():
                raise _lib.AfiError("parameters must be stored in the kernels' layout ([O][kh][kw][I]); "
                                    "construct the modules with afigan_amd.Generator / Discriminator")
        self._buf: Dict[str, torch.Tensor] = {}
        # this engine's own library state and the arithmetic of its big convolutions (None = the library's default, bf16x6: fp32 products
        # formed exactly on the bf16 matrix cores; "fp32" = the fp32 MFMA; "bf16x3" / "bf16" are opt-in: afi_ctx_set_compute_dtype).
        # TWO contexts, one per stream the step uses (include/afigan_hip.h: a context serves one stream at a time): `ctx` for everything
        # on the caller's stream (all forwards: its cache holds the forward weight transforms), `bctx` for the backward passes, which run
        # on the engine's second stream when the overlaps are on (its cache holds the data-gradient weight transforms, and it owns the
        # transform-domain weight-gradient accumulator and the library's side stream for the small-map weight gradients)
        self.ctx = _lib.Ctx(dtype)
        self.bctx = _lib.Ctx(dtype)
        self.after_allreduce = None
        self.losses = None
        self._loss_names: List[str] = []

    # ------------------------------------------------------------------------------------------------ helpers
    def _scratch(self, key: str, floats: int, device) -> torch.Tensor:
        t = self._buf.get(key)
        if t is None or t.numel() < floats:
            t = ops.new_workspace(floats, device)
            self._buf[key] = t
        return t

    def set_option(self, name: str, value: int) -> None:
        """A library option (afi_ctx_set_option) on BOTH of the engine's contexts: a backward pass must run under the options of its forward
        (include/afigan_hip.h, afi_discriminator_bwd), and the forwards run under ``ctx``, the backwards under ``bctx``."""
        self.ctx.set_option(name, value)
        self.bctx.set_option(name, value)

    _PAIRED_OPTIONS = ("winograd", "winograd_f4_backward", "winograd_f4_forward", "d_winograd_min_pixels", "d_fold_bn_apply", "f16_presplit",
                       "f16_local_sums", "d_fuse_tail", "d_fuse_bwd_sums")

    def _check_contexts_agree(self) -> None:
        if self.ctx.dtype != self.bctx.dtype:
            raise _lib.AfiError(f"the engine's two contexts run different arithmetic ({self.ctx.dtype} / {self.bctx.dtype})")
        for k in self._PAIRED_OPTIONS:
            if self.ctx.get_option(k) != self.bctx.get_option(k):
                raise _lib.AfiError(f"option {k!r} differs between the forward and the backward context ({self.ctx.get_option(k)} / "
                                    f"{self.bctx.get_option(k)}): set options with Stage1Step.set_option")

    @property
    def dtype(self) -> str:
        return self.ctx.dtype

    def set_dtype(self, dtype: str):
        """Arithmetic of the big convolutions' GEMMs for the following steps (both of the engine's contexts)."""
        self.ctx.set_dtype(dtype)
        self.bctx.set_dtype(dtype)

    def _join_bstream(self):
        if self._bstream is not None:
            torch.cuda.current_stream().wait_stream(self._bstream)

    # ------------------------------------------------------------------------------------------------ engine state (resume)
    def state_dict(self) -> Dict[str, object]:
        """What a resumed run needs beyond the two networks' own ``state_dict()``: the reference checkpoints, per network, the optimizer
        (momentum buffers), the scheduler (``last_epoch``) and the iteration (stage1_trainer.py:129-174, DetectionCheckpointer with
        ``optimizer=`` / ``scheduler=``).  Momentum buffers are keyed by parameter name and stored in the parameter's logical
        [O,I,kh,kw] shape, so the file does not depend on this engine's memory layout."""
        return {"iteration": int(self.iter),
                "G_optimizer": {"momentum_buffer": self.g_opt.state_dict()},
                "D_optimizer": {"momentum_buffer": self.d_opt.state_dict()},
                "scheduler": {"last_epoch": int(self.iter), "base_lr": self.base_lr, "steps": list(self.lr_steps), "gamma": self.lr_gamma,
                              "warmup_factor": self.warmup_factor, "warmup_iters": self.warmup_iters}}

    def load_state_dict(self, sd: Dict[str, object]):
        """Inverse of ``state_dict``; the networks' parameters / BN buffers are loaded by the caller into G and D (in place: the engine
        holds their storage).  The learning-rate schedule is a constructor argument and stays this engine's: a stored schedule that differs
        from it is refused (``ValueError``) rather than silently ignored.  Under data parallelism every rank loads the same file."""
        self._join_bstream()
        sch = sd.get("scheduler")
        if sch is not None:
            mine = self.state_dict()["scheduler"]
            for k in ("base_lr", "steps", "gamma", "warmup_factor", "warmup_iters"):
                if k in sch and (list(sch[k]) != list(mine[k]) if k == "steps" else abs(float(sch[k]) - float(mine[k])) > 1e-12 * max(1.0, abs(float(mine[k])))):
                    raise ValueError(f"checkpoint scheduler {k} = {sch[k]!r} differs from this engine's {mine[k]!r}: construct Stage1Step with the stored schedule")
            if "last_epoch" in sch and int(sch["last_epoch"]) != int(sd["iteration"]):
                raise ValueError(f"checkpoint scheduler.last_epoch {sch['last_epoch']} != iteration {sd['iteration']}")
        self.g_opt.load_state_dict(sd["G_optimizer"]["momentum_buffer"])
        self.d_opt.load_state_dict(sd["D_optimizer"]["momentum_buffer"])
        self.iter = int(sd["iteration"])

    # ---- the reference's own checkpoint layout (stage1_trainer.py:129-174: one DetectionCheckpointer per network, each saving
    #      {"model", "optimizer", "scheduler", "iteration"} with torch.optim.SGD / WarmupMultiStepLR state dicts; resume_or_load returns
    #      checkpoint["iteration"] -- the iteration that FINISHED -- and the loop restarts at + 1, :167-172)
    def to_reference_checkpoints(self) -> Dict[str, Dict[str, object]]:
        """{"G": ..., "D": ...}: what ``G_checkpointer.save`` / ``D_checkpointer.save`` write besides "model" after this engine's last step:
        "optimizer" = a torch.optim.SGD state_dict (state keyed by the parameter's index in ``model.parameters()`` order, momentum buffers in
        the parameter's logical shape; param_groups as detectron2 v0.1.1's build_optimizer makes them: ONE group per parameter with its own
        weight decay), "scheduler" = WarmupMultiStepLR.state_dict() essentials, "iteration" = the index of the iteration that just finished
        (``self.iter - 1``: this engine counts steps DONE)."""
        out = {}
        for tag, net, opt in (("G", self.G, self.g_opt), ("D", self.D, self.d_opt)):
            names = [n for n, _ in net.named_parameters()]
            mom = opt.state_dict()
            wds = dict(zip(opt.names, opt.weight_decays))
            lr = self.lr_at(self.iter)
            state = {i: {"momentum_buffer": mom[n]} for i, n in enumerate(names)} if self.iter > 0 else {}
            groups = [{"lr": lr, "initial_lr": self.base_lr, "momentum": self.momentum, "dampening": 0, "weight_decay": wds[n], "nesterov": False,
                       "params": [i]} for i, n in enumerate(names)]
            out[tag] = {"optimizer": {"state": state, "param_groups": groups},
                        "scheduler": {"last_epoch": int(self.iter), "_step_count": int(self.iter) + 1, "base_lrs": [self.base_lr] * len(names),
                                      "milestones": list(self.lr_steps), "gamma": self.lr_gamma, "warmup_factor": self.warmup_factor,
                                      "warmup_iters": self.warmup_iters, "warmup_method": "linear"},
                        "iteration": int(self.iter) - 1}
        return out

    def load_reference_checkpoints(self, ckpts: Dict[str, Dict[str, object]]):
        """Inverse of ``to_reference_checkpoints`` (the "model" entries are the caller's: ``G.load_state_dict`` / ``D.load_state_dict``).  Takes
        what the reference's two checkpointers wrote: SGD state keyed by parameter index -- mapped through ``named_parameters()`` order --, and
        resumes at ``iteration + 1`` like ``resume_or_load`` does.  Both files must agree on the iteration; the stored schedule is checked
        against this engine's."""
        self._join_bstream()
        its = {int(ckpts[t]["iteration"]) for t in ("G", "D")}
        if len(its) != 1:
            raise ValueError(f"G and D checkpoints disagree on the iteration: {sorted(its)}")
        it = its.pop() + 1
        for tag, net, opt in (("G", self.G, self.g_opt), ("D", self.D, self.d_opt)):
            ck = ckpts[tag]
            names = [n for n, _ in net.named_parameters()]
            st = ck["optimizer"]["state"]
            if st and sorted(int(k) for k in st) != list(range(len(names))):
                raise KeyError(f"{tag} optimizer state: indices {sorted(st)} do not cover the {len(names)} parameters")
            if st:
                opt.load_state_dict({n: st[i if i in st else str(i)]["momentum_buffer"] for i, n in enumerate(names)})
            else:
                opt.flat_mom.zero_()
            sch = ck.get("scheduler")
            if sch is not None:
                if list(sch.get("milestones", self.lr_steps)) != list(self.lr_steps) or abs(float(sch.get("gamma", self.lr_gamma)) - self.lr_gamma) > 1e-12 or \
                        int(sch.get("warmup_iters", self.warmup_iters)) != self.warmup_iters or abs(float(sch.get("warmup_factor", self.warmup_factor)) - self.warmup_factor) > 1e-12:
                    raise ValueError(f"{tag} checkpoint's schedule differs from this engine's (milestones / gamma / warm-up)")
                if "last_epoch" in sch and int(sch["last_epoch"]) != it:
                    raise ValueError(f"{tag} scheduler.last_epoch {sch['last_epoch']} != iteration + 1 = {it}")
        self.iter = it

    def lr_at(self, it: int) -> float:
        return warmup_multistep_lr(self.base_lr, it, self.lr_steps, self.lr_gamma, self.warmup_factor, self.warmup_iters)

    def _g_forward(self, level: int, lr: torch.Tensor, key: str):
        N, Cc, H, W = lr.shape
        G = self.G
        n = self._lib.afi_generator_fwd_ws_floats(G.in_channels, G.growth_rate, G.n_residual_dense_blocks, N, H, W)
        ws = self._scratch(f"{key}{level}", n, lr.device)
        out = self._buf.get(f"gout{key}{level}")
        if out is None or tuple(out.shape) != (N, Cc, 2 * H, 2 * W):
            out = ops.new_pixel_major(N, Cc, 2 * H, 2 * W, lr.device)
            self._buf[f"gout{key}{level}"] = out
        call("afi_generator_fwd", C.byref(self._gprm), ops.view_of(lr), N, H, W, ops.view_of(out), C.c_void_p(ws.data_ptr()), n,
             ops.stream_ptr())
        return out, ws

    def _paired(self, x: torch.Tensor) -> bool:
        # (a separate threshold for the G phase's two forwards was measured: never / 40,000 / always = 81.0-81.1 / 80.5-81.3 / 81.1-82.1 ms per step:
        # one knob)
        return 0 < x.shape[0] * x.shape[2] * x.shape[3] <= self.pair_d_max_pixels

    def _pair(self, level: int, first: torch.Tensor, second: torch.Tensor) -> torch.Tensor:
        """[first; second] along the batch, pixel-major, in a buffer kept per level (the D phase's backward reads it later)."""
        N, Cc, H, W = first.shape
        key = f"pair{level}"
        t = self._buf.get(key)
        if t is None or tuple(t.shape) != (2 * N, Cc, H, W):
            t = ops.new_pixel_major(2 * N, Cc, H, W, first.device)
            self._buf[key] = t
        t[:N].copy_(first)
        t[N:].copy_(second)
        return t

    def _d_forward(self, x: torch.Tensor, ws_key: str, backward_follows: bool = True, stats_only: bool = False, paired: bool = False):
        """training = 1: train-mode forward whose saved activations feed afi_discriminator_bwd; 2: train-mode statistics and
        logits only (the G phase, stage1_trainer.py:399-403: no gradient ever flows through these two calls, Q1); 3: the BatchNorm
        side effects only (the G phase's D(real) call, :401-403: nothing reads its logits -- the running statistics advance exactly as
        in the reference, the last activation / last conv / logits are not computed)."""
        N, _, H, W = x.shape
        F = (C.c_int * 4)(*self.dnet.F)
        mode = 1 if backward_follows else (3 if stats_only else 2)
        # what THIS context's call of this kind writes (ABI v7): the kept F(4x4) input planes only where a backward reads them
        n = self._lib.afi_discriminator_fwd_ws_floats_ex(_lib.current_ctx().handle, F, N, H, W, mode)
        ws = self._scratch(ws_key, n, x.device)
        logits = self._scratch(ws_key + "_logits", N * H * W, x.device)
        call("afi_discriminator_fwd_paired" if paired else "afi_discriminator_fwd", C.byref(self._dprm), ops.view_of(x), N, H, W,
             C.c_void_p(logits.data_ptr()), mode, C.c_void_p(ws.data_ptr()), n, ops.stream_ptr())
        return logits, ws

    def _d_backward(self, x: torch.Tensor, ws: torch.Tensor, dlogits: torch.Tensor, paired: bool = False):
        N, _, H, W = x.shape
        F = (C.c_int * 4)(*self.dnet.F)
        n = self._lib.afi_discriminator_bwd_ws_floats(F, N, H, W)
        sc = self._scratch("d_bwd", n, x.device)
        with _lib.use_ctx(self.bctx):                      # the backward passes' own context (they may be on the second stream)
            call("afi_discriminator_bwd_paired" if paired else "afi_discriminator_bwd", C.byref(self._dprm), C.byref(self._dgrad), ops.view_of(x), N, H, W,
                 C.c_void_p(ws.data_ptr()), C.c_void_p(dlogits.data_ptr()), C.c_void_p(None), C.c_void_p(sc.data_ptr()), n, ops.stream_ptr())

    @staticmethod
    def _crop_pair(tr: torch.Tensor, hr: torch.Tensor):
        """_reshape_stage1 applied both ways (stage1_trainer.py:345-346,437-443): crop from the origin, never pad."""
        h, w = min(tr.shape[2], hr.shape[2]), min(tr.shape[3], hr.shape[3])
        return tr[:, :, :h, :w], hr[:, :, :h, :w]

    # ONE collective per network per iteration: sum the flat gradient buffer over the ranks (RCCL over xGMI; gloo in the CPU rehearsal).
    # The 1/world averaging is folded into the fused SGD kernel's gradient scale, so between the collective and the optimizer step
    # ``param.grad`` holds the SUM over ranks.
    def _allreduce_start(self, opt: _FlatOptim):
        """Issue the collective behind what the CURRENT stream has queued and return at once (``async_op``: RCCL runs it on its own
        stream); nothing may touch ``opt.flat_grad`` until ``_allreduce_finish``."""
        if self.distributed and self.overlap_comm:
            work = torch.distributed.all_reduce(opt.flat_grad, op=torch.distributed.ReduceOp.SUM, group=self.pg, async_op=True)
            self._pending_work.append(work)
            return work
        if self.distributed:
            if self.measure_comm:                          # blocking: the whole exchange is exposed
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                allreduce_sum_(opt.flat_grad, self.pg)
                e1.record()
                self._comm_events.append(("D" if opt is self.d_opt else "G", e0, e1))
            else:
                allreduce_sum_(opt.flat_grad, self.pg)
        return None

    def _allreduce_finish(self, opt: _FlatOptim, work):
        if work is not None:
            if self.measure_comm:                          # what the consumer's stream really waits: events on it either side of the wait
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                t0 = time.perf_counter()
                work.wait()                                # (nccl: the current stream waits for the collective; gloo: the host does)
                host_ms = (time.perf_counter() - t0) * 1e3
                e1.record()
                self._comm_events.append(("D" if opt is self.d_opt else "G", e0, e1, host_ms))
            else:
                work.wait()
            if work in self._pending_work:
                self._pending_work.remove(work)
        if self.after_allreduce is not None:       # observation point for tests: `opt.flat_grad` holds the SUM over ranks here
            self.after_allreduce("D" if opt is self.d_opt else "G", opt)

    def _drain_pending_comm(self):
        """Error path: a phase raised with a collective in flight.  Wait for it before anything (the next step's zero_grad) touches the
        buffer it is still writing."""
        for w in self._pending_work:
            try:
                w.wait()
            except Exception:
                pass
        self._pending_work = []

    def comm_exposure(self):
        """After a step run with ``measure_comm = True``: {"D": ms, "G": ms, "total": ms} the CONSUMING STREAM spent waiting for each exchange
        (device time between two events either side of the wait), and under ``"host_blocked_ms"`` the time the HOST spent inside the wait.
        Under nccl / RCCL the host does not block (``wait`` only orders streams) and the device figure is the exposure.  Under gloo ``wait``
        blocks the host until the exchange is done, and that exchange cannot start before the device has caught up with everything the
        host had queued ahead of it (the whole D phase): the host figure there measures the host's LEAD over the device plus the exchange
        (182 ms against an 18.8 ms exchange in round 5's two-ranks-one-GPU rehearsal), not an exposure -- it is reported apart and never
        folded into the device figure."""
        torch.cuda.synchronize()
        out, host = {}, {}
        for rec in self._comm_events:
            out[rec[0]] = out.get(rec[0], 0.0) + rec[1].elapsed_time(rec[2])
            if len(rec) > 3:
                host[rec[0]] = host.get(rec[0], 0.0) + rec[3]
        out["total"] = sum(out.values())
        out["host_blocked_ms"] = host
        return out

    # ------------------------------------------------------------------------------------------------ the step
    def run_step(self, lr_features: Sequence[torch.Tensor], hr_features: Sequence[torch.Tensor]):
        """lr_features / hr_features: lists over levels (p2..p6) of [N,C,h,w] fp32 GPU tensors (detached guide features).
        Returns nothing; read ``metrics()`` for the loss values."""
        if not (self.G.training and self.D.training):
            raise AssertionError("[Stage1Step] model was changed to eval mode!")       # stage1_trainer.py:309
        nlev = len(lr_features)
        assert nlev == len(hr_features) and nlev > 0
        dev = lr_features[0].device
        lrs = [ops.pixel_major(t) for t in lr_features]
        hrs = [ops.pixel_major(t) for t in hr_features]
        names = []
        for i in range(nlev):
            lv = self.first_level + i
            names += [f"d_loss_p{lv}", f"adv_loss_p{lv}", f"content_loss_p{lv}"]
        if self.losses is None or self._loss_names != names:
            self.losses = torch.zeros(len(names), device=dev, dtype=torch.float32)
            self._loss_names = names
        self.losses.zero_()
        lptr = self.losses.data_ptr()
        lr_now = self.lr_at(self.iter)
        self._check_contexts_agree()
        self._comm_events = []
        # transformed conv weights are shared by the calls of a phase (weights only change at the two optimizer steps)
        cx, bx = self.ctx, self.bctx
        self._outer_ctx = _lib.current_ctx()               # what the caller's own calls go to: the hooks run under it, not under cx / bx
        with _lib.use_ctx(cx):
            if self.weight_cache:
                for c_, key in ((cx, "wino_wcache"), (bx, "wino_wcache_b")):
                    wcache = self._scratch(key, self.WINO_WCACHE_FLOATS, dev)
                    call("afi_ctx_set_wino_weight_cache", c_.handle, C.c_void_p(wcache.data_ptr()), self.WINO_WCACHE_FLOATS)
                # ... and the transform-domain weight-gradient sums of a phase are transformed back once, before its all-reduce
                if self.wgrad_accum:
                    wgacc = self._scratch("wino_wgacc", self.WINO_WGACC_FLOATS, dev)
                    call("afi_ctx_set_wino_wgrad_accum", bx.handle, C.c_void_p(wgacc.data_ptr()), self.WINO_WGACC_FLOATS)
            try:
                self._run_phases(nlev, lrs, hrs, lptr, lr_now, dev)
            except BaseException:
                # a phase failed.  First the streams: backward kernels still queued on the second stream read lrs / hrs / workspaces that
                # were allocated on the caller's stream -- the caller's stream must not get that memory back before they are done
                # (ADVICE r2).  Then drop the partial transform-domain sums (never add them into param.grad) and let the first error out.
                try:
                    self._join_bstream()
                except Exception:
                    pass
                self._drain_pending_comm()                 # a collective issued before the failure may still be writing flat_grad
                self._lib.afi_ctx_wino_wgrad_discard(bx.handle)
                self._lib.afi_ctx_set_wino_wgrad_accum(bx.handle, None, 0)
                self._lib.afi_ctx_set_wino_weight_cache(cx.handle, None, 0)
                self._lib.afi_ctx_set_wino_weight_cache(bx.handle, None, 0)
                raise
            call("afi_ctx_set_wino_wgrad_accum", bx.handle, C.c_void_p(None), 0)        # (both phases flushed their sums)
            call("afi_ctx_set_wino_weight_cache", cx.handle, C.c_void_p(None), 0)
            call("afi_ctx_set_wino_weight_cache", bx.handle, C.c_void_p(None), 0)
        self.iter += 1

    # per context: the forward one peaks in the G phase (D's three layers at both tilings, 16 + 36 planes of 1.70 M elements = 89 M, + G's
    # forward transforms and packed conv-transpose weight when G is recomputed: 104 M), the backward one in the D phase (the same 89 M);
    # under the bf16 settings an element is cached as three bf16 parts (1.5 floats: the DMA GEMM's pre-split operand) -> 156 M floats
    WINO_WCACHE_FLOATS = 168 * 1024 * 1024
    WINO_WGACC_FLOATS = 100 * 1024 * 1024

    def _run_phases(self, nlev, lrs, hrs, lptr, lr_now, dev):
        # ---------------- D phase (stage1_trainer.py:334-381)
        self.d_opt.zero_grad()                                                       # :374
        trs = []
        for i in range(nlev):
            if self.on_d_level is not None:
                # the caller's hook queues the CALLER's work (possibly on another stream): under the context the caller had, never under the
                # engine's, whose phase weight cache and op scratch belong to the step's own streams
                with _lib.use_ctx(self._outer_ctx):
                    self.on_d_level(i)
            tr, ws = self._g_forward(i, lrs[i], "g_ws")                              # :339-341 (.detach(): no graph anyway)
            trs.append((tr, ws))
            tr_c, hr_c = self._crop_pair(tr, hrs[i])                                  # :345-346
            paired = self._paired(hr_c)
            calls = ((self._pair(i, hr_c, tr_c), (1.0, 0.0), "d_ws"),) if paired else ((hr_c, (1.0,), "d_ws"), (tr_c, (0.0,), "d_ws"))
            for x, targets, key in calls:                                            # :349-353, :355-359 (real, then fake)
                if self.overlap_d:
                    key = f"d_ws_{i}_{'p' if paired else int(targets[0])}"           # lives until its backward has run
                logits, dws = self._d_forward(x, key, paired=paired)
                half = hr_c.shape[0] * hr_c.shape[2] * hr_c.shape[3]                # (scratch tensors may be larger than asked for)
                dz = self._scratch("dlogits" + (key if self.overlap_d else ""), half * len(targets), dev)
                for h, target in enumerate(targets):
                    call("afi_bce_logits_fwd_bwd", C.c_void_p(logits.data_ptr() + 4 * h * half), half, target, 1.0,
                         C.c_void_p(lptr + 4 * (3 * i)), 1.0, C.c_void_p(dz.data_ptr() + 4 * h * half), ops.stream_ptr())
                if self.overlap_d:
                    # all forwards stay in order on the caller's stream (BatchNorm running statistics), all backwards in
                    # order on a second one (gradient accumulation): same results, and a backward's GEMMs run beside the
                    # next forward's bandwidth-bound transforms / BatchNorm passes
                    if self._bstream is None:
                        self._bstream = torch.cuda.Stream(device=dev)
                    self._bstream.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(self._bstream):
                        self._d_backward(x, dws, dz, paired=paired)
                else:
                    self._d_backward(x, dws, dz, paired=paired)                      # :375 (accumulates into the flat grads)
        if self.overlap_d:
            self._join_bstream()
        call("afi_ctx_wino_wgrad_flush", self.bctx.handle, ops.stream_ptr())         # (joined: the backward context's sums, on the caller's stream)
        # D's gradient exchange (61.4 MB at the reference's widths) is issued here and waited for only where D's weights are needed: G's
        # five backward passes need nothing of D (below), so with the second stream they run beside the collective
        d_work = self._allreduce_start(self.d_opt)

        def d_finish():
            self._allreduce_finish(self.d_opt, d_work)
            self.d_opt.step(lr_now, self.momentum, gscale=1.0 / self.world)          # :381
            call("afi_ctx_wino_weight_cache_invalidate", self.ctx.handle)            # D's weights moved
            call("afi_ctx_wino_weight_cache_invalidate", self.bctx.handle)

        # ---------------- G phase (:384-433)
        self.g_opt.zero_grad()                                                       # :426

        def g_backward(i, tr, ws):                                                   # :410 (L1 term) and :427; under the backward context
            N, Cc, Ha, Wa = tr.shape
            tr_c, _ = self._crop_pair(tr, hrs[i])
            da = self._scratch("g_dout", tr.numel(), dev)
            call("afi_l1_fwd_bwd", ops.view_of(tr), ops.view_of(hrs[i]), N, tr_c.shape[2], tr_c.shape[3], Cc, Ha, Wa, 1.0,
                 C.c_void_p(lptr + 4 * (3 * i + 2)), 1.0, C.c_void_p(da.data_ptr()), ops.stream_ptr())
            lrt = lrs[i]
            n = self._lib.afi_generator_bwd_ws_floats(self.G.in_channels, self.G.growth_rate, self.G.n_residual_dense_blocks,
                                                      lrt.shape[0], lrt.shape[2], lrt.shape[3])
            sc = self._scratch("g_bwd", n, dev)
            call("afi_generator_bwd", C.byref(self._gprm), C.byref(self._ggrad), ops.view_of(lrt), lrt.shape[0], lrt.shape[2], lrt.shape[3],
                 C.c_void_p(ws.data_ptr()), C.c_void_p(da.data_ptr()), C.c_void_p(None), C.c_void_p(sc.data_ptr()), n, ops.stream_ptr())

        # The adversarial term carries no gradient (Q1): G's backward needs only the L1 term, i.e. nothing the D forwards of this phase
        # produce -- only the forwards kept from the top of the step.  With the second stream all five backward passes are queued on it
        # first, SMALLEST LEVEL FIRST, and run beside the D forwards, which go largest level first (their order is fixed by the BatchNorm
        # running statistics): latency-bound small-map chains beside chip-filling GEMMs at both ends of the phase, instead of big beside
        # big at its head and small beside small (an idle chip) at its tail.  (The losses land in their own slots; the weight gradients
        # are summed in another order: fp32 rounding only.)
        side = self.overlap_d and self.reuse_g and self.overlap_g
        g_work = None
        if side:
            if self._bstream is None:
                self._bstream = torch.cuda.Stream(device=dev)
            self._bstream.wait_stream(torch.cuda.current_stream())                   # behind the D phase's flush and G's zero_grad (not behind D's all-reduce)
            with torch.cuda.stream(self._bstream), _lib.use_ctx(self.bctx):
                for i in (reversed(range(nlev)) if self.g_bwd_small_first else range(nlev)):
                    g_backward(i, *trs[i])
                # G's gradients are final once its last backward pass has run: its sums are transformed back and its exchange (31.3 MB) is
                # issued from the second stream, beside the D forwards the caller's stream still has to run
                call("afi_ctx_wino_wgrad_flush", self.bctx.handle, ops.stream_ptr())
                g_work = self._allreduce_start(self.g_opt)
        d_finish()                                                                   # D's weights move before the G phase's D forwards read them
        for i in range(nlev):
            if self.reuse_g:
                tr, ws = trs[i]                                                      # Q5: identical to recomputing G(lr)
            else:
                tr, ws = self._g_forward(i, lrs[i], "g_ws")                          # :389-391
            tr_c, hr_c = self._crop_pair(tr, hrs[i])
            if self._paired(tr_c):                                                   # :399-403 (fake first, then real), one call
                x = self._pair(i, tr_c, hr_c)
                logits, _ = self._d_forward(x, "d_ws", backward_follows=False, paired=True)
                call("afi_bce_logits_fwd_bwd", C.c_void_p(logits.data_ptr()), tr_c.shape[0] * tr_c.shape[2] * tr_c.shape[3], 1.0, 1.0,
                     C.c_void_p(lptr + 4 * (3 * i + 1)), 0.0, C.c_void_p(None), ops.stream_ptr())
            else:
                for x, key in ((tr_c, "adv"), (hr_c, None)):                         # :399-403 (fake first, then real)
                    logits, _ = self._d_forward(x, "d_ws", backward_follows=False, stats_only=(key != "adv"))
                    if key == "adv":                                                 # :408, no gradient (Q1)
                        call("afi_bce_logits_fwd_bwd", C.c_void_p(logits.data_ptr()), x.shape[0] * x.shape[2] * x.shape[3], 1.0, 1.0,
                             C.c_void_p(lptr + 4 * (3 * i + 1)), 0.0, C.c_void_p(None), ops.stream_ptr())
            if not side:
                with _lib.use_ctx(self.bctx):
                    g_backward(i, tr, ws)
        if self.overlap_d and self.reuse_g:
            self._join_bstream()
        if not side:
            call("afi_ctx_wino_wgrad_flush", self.bctx.handle, ops.stream_ptr())
            g_work = self._allreduce_start(self.g_opt)
        self._allreduce_finish(self.g_opt, g_work)
        self.g_opt.step(lr_now, self.momentum, gscale=1.0 / self.world)              # :433

    def metrics(self, check_finite: bool = True, reduce: bool = False, data_time: Optional[float] = None) -> Dict[str, float]:
        """Loss values of the last step (one device sync).  g_loss_p = 1e-3*adv + content (stage1_trainer.py:411).
        ``reduce=True`` in a data-parallel run: the MEAN over the ranks -- the reference's ``_write_metrics`` gathers every rank's dict by
        pickle on every iteration and averages on rank 0 (stage1_trainer.py:453-492); here it is ONE all-reduce of the 3-per-level loss
        vector, whenever the caller asks (every logging period, not every iteration), and every rank gets the averages.  Collective:
        all ranks must call it together.

        The names ``_write_metrics`` puts into the event storage are all here: the per-level losses, ``total_loss`` -- the reference writes
        it twice per iteration, the sum of the d_loss_p* after the D phase (:363-368) and the sum of the g_loss_p* after the G phase
        (:413-420); the storage keeps the later one, so ``total_loss`` is the G-phase sum and the D-phase sum is ``d_total_loss`` -- and, when
        the caller passes the seconds its loader took (``data_time``, :315), ``data_time`` / ``G_data_time`` / ``D_data_time``: the MAXIMUM over
        the ranks under ``reduce`` (:468-483), not the mean."""
        vec = self.losses.detach()
        if reduce and self.distributed:
            vec = vec.clone()
            torch.distributed.all_reduce(vec, op=torch.distributed.ReduceOp.SUM, group=self.pg)
            vec = vec / self.world
            if data_time is not None:
                dt = torch.tensor([float(data_time)], device=vec.device)
                torch.distributed.all_reduce(dt, op=torch.distributed.ReduceOp.MAX, group=self.pg)
                data_time = float(dt.item())
        vals = vec.cpu().tolist()
        out = dict(zip(self._loss_names, vals))
        for k in list(out):
            if k.startswith("adv_loss_p"):
                lv = k[len("adv_loss_p"):]
                out[f"g_loss_p{lv}"] = out[k] * 1e-3 + out[f"content_loss_p{lv}"]
        if check_finite and not all(np.isfinite(v) for v in out.values()):          # _detect_anomaly (:445-451)
            raise FloatingPointError(f"Loss became infinite or NaN at iteration={self.iter}!\nloss_dict = {out}")
        out["d_total_loss"] = sum(v for k, v in out.items() if k.startswith("d_loss_p"))
        out["total_loss"] = sum(v for k, v in out.items() if k.startswith("g_loss_p"))
        if data_time is not None:
            out["data_time"] = out["G_data_time"] = out["D_data_time"] = float(data_time)
        return out
