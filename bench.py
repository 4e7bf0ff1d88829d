#!/usr/bin/env python3
"""bench.py -- stage-1 AFI-GAN G+D step on MI355X (BASELINE.json configs[1]) + AF-interpolator fwd+bwd (configs[0] shape).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one full stage-1 iteration on a per-GPU batch of 2 synthetic 3x800x1333 images (stage1_trainer.py:305-435):
two frozen R-50-FPN guide forwards (image and image_x0.5), the D step and the G step over P2..P6 (hand-written HIP kernels),
the RCCL gradient all-reduce (N > 1) and both fused SGD updates.  Inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line.  `value` = images/s of the whole job; weak scaling (2 images per GPU).

Extra objects on the line:
  roofline      -- the dominant kernel of the timed region (by summed HIP-event time, measured on the launch stream by the
                   library's own event brackets): algorithmic FLOP / time vs the dense fp32 MFMA peak (157.3 TFLOP/s).
  cpu_baseline  -- the CPU oracle (kind "port": oracle/afigan_oracle.py, a PyTorch-CPU restatement pinned to the reference's
                   outputs) timed on this host on a bounded sample of the same workload (levels P3..P6 only), scaled to images/s.
  af_interpolator -- BASELINE metric 1: Generator fwd+bwd feature-Mpix/s on 1x256x25x34 -> 1x256x50x68 (and batch 16).
  fpn_topdown / pafpn / bifpn_inference / stage2_adversarial / dual_scale_data_path -- the SURVEY 8(f) rows, N = 1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# the legs beside the headline step, and the N-rank launch / cross-rank checks (tools/interp_sweep.py etc. reach them through this module too)
from tools.bench_legs import (D_FWD_FLOP_PER_PX, D_FWDBWD_DETACHED_FLOP_PER_PX, G_FWD_FLOP_PER_INPX, PEAK_BF16_MFMA_TFLOPS,  # noqa: E402,F401
                              PEAK_FP32_MFMA_TFLOPS, bifpn_bench, bifpn_train_bench, committed_traffic, cpu_baseline, file_sha256, fpn_bench,
                              gemm_peak, host_cores, interp_bench, kind_peak, log, stage2_bench)
from tools.bench_launch import allreduce_alone, identical_across_ranks, per_rank_rates, rehearse_launch, spawn_ranks  # noqa: E402,F401

# Descriptive strings of the JSON line, kept short: the driver's record truncates long values (round 5: `workload` was cut mid-word).
DTYPE_NOTE = {
    "fp32": "f32",
    "f16x3": "f32 via f16x3: two scaled fp16 pieces per operand, three f16 MFMAs per product, fp32 accumulate (tensors fp32)",
    "bf16x6": "f32 via bf16x6: three exact bf16 pieces per operand, six bf16 MFMAs per product, fp32 accumulate (tensors fp32)",
    "bf16x3": "bf16x3: two bf16 pieces per operand, three bf16 MFMAs per product, fp32 accumulate (tensors fp32)",
    "bf16": "bf16 operands, fp32 accumulate (tensors fp32)"}
PEAK_NOTE = ("fp32-equivalent TFLOP/s (2*M*N*K once per product); peak = dense f16 / bf16 MFMA 2500 / MFMAs per product (3, 6 or 1), or the "
             "fp32 MFMA 157.3.  `achieved`: two-stream step (launches share the chip); `kernel_alone`: one stream.  hipBLASLt's plain fp16 "
             "GEMM on this step's largest shape: 922 TFLOP/s = 307 at three products (profiles/r05/hipblaslt_f16_ceiling.txt); the chip is "
             "power-limited under dense MFMA work (traffic.held_clock_ghz, traffic.mfma_busy_at_held_clock): DESIGN.md 4b")
CONV_NOTE = ("3x3 convs with >= 128 channels both sides and >= 1024 pixels: Winograd.  F(4x4,3x3) (points {0,1,-1,1/2,-2,inf}) for data and "
             "weight gradients, forward-only passes and D's blocks 1 and 2 (winograd_f4_forward = 12 with k-step-local sums, f16_local_sums = 12: "
             "below torch fp32's gradient deviation at P2 and P3, profiles/r06/dflip_*); F(2x2,3x3) for the other forwards a backward follows.  "
             "Direct implicit GEMM on the fp32 MFMA elsewhere")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-per-gpu", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-interp", action="store_true", help="skip the AF-interpolator micro-benchmark")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl == RCCL; gloo only to rehearse "
                    "the multi-process path with several ranks sharing one GPU)")
    ap.add_argument("--dtype", default=None, choices=["fp32", "f16x3", "bf16x6", "bf16x3", "bf16"],
                    help="how the big convolutions' GEMMs form their fp32 products (afi_ctx_set_compute_dtype).  Default: the library's "
                         "(f16x3: two scaled fp16 pieces per operand, three products, fp32-grade); the default run also times the other settings on the same "
                         "engine afterwards and reports them under other_dtypes")
    ap.add_argument("--overlap-comm", type=int, default=None, choices=[0, 1], help="N > 1: issue the two gradient all-reduces asynchronously beside independent work "
                    "(Stage1Step(overlap_comm=...)); default: the engine's (on under gloo, off under nccl until an RCCL run has exercised it)")
    ap.add_argument("--profile-timed", action="store_true", help="record the library's per-launch HIP events INSIDE the timed region (default: over a repeat of "
                    "the same K steps right behind it, so that the headline number carries no event records); the rocprofv3 passes use it: their traces "
                    "then hold exactly the timed launches")
    ap.add_argument("--synthetic-pyramid", action="store_true",
                    help="feed seeded randn pyramids instead of running the R-50-FPN guide (debug only; not the headline config)")
    ap.add_argument("--pair-d-max-pixels", type=int, default=None, help="Stage1Step(pair_d_max_pixels=...): levels up to this many pixels run D(real) and D(fake) "
                    "of a phase as one call (per-batch BatchNorm statistics); default: the engine's")
    ap.add_argument("--one-stream", action="store_true", help="Stage1Step(overlap_d=False, overlap_g=False): every kernel alone on the chip (the "
                    "profiling passes of tools/prof_r03.sh: per-kernel durations comparable across rounds)")
    ap.add_argument("--no-guide-prefetch", action="store_true", help="run the frozen guide network's two forwards at the head of every step on the step's own "
                    "stream (default: the NEXT batch's guide forwards are issued on a second stream while the current batch trains -- "
                    "afigan_amd.GuidePrefetcher; one guide pair and one training step per timed step either way)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE", help="afi_ctx_set_option on the engine's contexts (A/B runs), "
                    "e.g. --option g_batch_growth_grads=0; names: afigan_amd._lib.OPTIONS")
    ap.add_argument("--dist-world-1", action="store_true", help="--gpus 1 only: run the distributed code path anyway, on a ONE-rank process group of "
                    "--backend (a real RCCL communicator under nccl): broadcast, both all-reduces, the `comm` object and the overlap_comm A/B leg "
                    "execute on a one-GPU box.  Not the headline configuration")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="launch path only (no GPU work, no metric): spawn / rendezvous / all-reduce / invariant checks of the N-rank job with "
                         "CPU tensors; what tests/test_host_logic.py runs with --gpus 2 --backend gloo in a container without a GPU")
    return ap.parse_args()




def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:          # no launcher around us: be the launcher (before anything touches the GPU)
        os.environ["AFI_BENCH_SPAWNED"] = "1"
        spawn_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch N ranks for --gpus N "
                         f"(python bench.py --gpus N spawns them itself; torch.distributed.run must use --nproc-per-node N)")
    if args.rehearse_launch:
        rehearse_launch(args, world, rank)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the AFI-GAN hot path has no CPU fallback)")
    local_rank = local_rank % torch.cuda.device_count()      # (gloo rehearsal: several ranks may share one GPU)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.dist_world_1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29611")         # (set by every launcher; only --dist-world-1 without one gets here unset)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    import __graft_entry__ as ge
    if rank == 0:
        ge.build(verbose=False)
    if dist is not None:
        dist.barrier()
    import afigan_amd as amd
    from afigan_amd import _lib
    from afigan_amd.guide import GuideR50FPN
    lib = _lib.load()

    B = args.batch_per_gpu
    torch.manual_seed(1234)                       # same init on every rank (and rank 0's weights are broadcast anyway)
    G = amd.Generator(n_residual_dense_blocks=3).to(dev)
    D = amd.Discriminator().to(dev)
    G.train(); D.train()
    step = amd.Stage1Step(G, D, base_lr=1e-3, dtype=args.dtype, overlap_d=not args.one_stream, overlap_g=not args.one_stream,
                          distributed=(True if args.dist_world_1 else None),
                          g_bwd_small_first=os.environ.get("AFI_BENCH_G_BWD_ORDER", "small-first") != "level-order")   # (A/B of the G-phase schedule)
    if args.pair_d_max_pixels is not None:
        step.pair_d_max_pixels = args.pair_d_max_pixels
    for kv in args.option:
        name, _, val = kv.partition("=")
        step.set_option(name, int(val))
    run_dtype = step.dtype                             # the library's default when --dtype is not given
    # data-parallel runs: what the process group says it is, and what the two gradient exchanges of a step cost with nothing beside them
    # (the engine's own flat gradient buffers: D 61.4 MB, G 31.3 MB), before any step runs -- SURVEY 8e (2), (3)
    comm = None
    if dist is not None:
        if args.overlap_comm is not None:
            step.overlap_comm = bool(args.overlap_comm)
        comm = {"backend": dist.get_backend(), "world_size_reported": dist.get_world_size(), "rank0_device": torch.cuda.get_device_name(dev),
                "overlap_comm": bool(step.overlap_comm),
                "allreduce_alone": allreduce_alone(dist, torch, {"D": step.d_opt.flat_grad, "G": step.g_opt.flat_grad}, dev, world)}
        log(f"process group: backend {comm['backend']}, world {comm['world_size_reported']}; all-reduce alone: " +
            ", ".join(f"{k} {v['ms']:.3f} ms ({v['bus_gb_per_s']} GB/s bus)" for k, v in comm["allreduce_alone"].items()))
    guide = None if args.synthetic_pyramid else GuideR50FPN().to(dev)
    gen = torch.Generator(device=dev).manual_seed(100 + rank)     # each rank owns a different shard of the global batch
    images = torch.rand((B, 3, 800, 1333), device=dev, generator=gen) * 255.0
    images_half = torch.nn.functional.interpolate(images, size=(400, 666), mode="bilinear", align_corners=False)
    hr_shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    lr_shapes = [(104, 168), (52, 84), (26, 42), (13, 21), (7, 11)]
    if guide is None:
        syn_hr = [torch.randn((B, 256, h, w), device=dev, generator=gen).contiguous(memory_format=torch.channels_last) for h, w in hr_shapes]
        syn_lr = [torch.randn((B, 256, h, w), device=dev, generator=gen).contiguous(memory_format=torch.channels_last) for h, w in lr_shapes]

    def guide_pair():
        hr_ = guide(images)                                        # stage1_trainer.py:320
        lr_ = guide(images_half)                                   # :321
        return [hr_[f"p{d}"] for d in range(2, 7)], [lr_[f"p{d}"] for d in range(2, 7)]      # :325-327

    # The guide is frozen (eval, no_grad, no BatchNorm updates): the features of batch i + 1 do not depend on the G / D updates of
    # iteration i, so its two forwards are issued on a second stream while iteration i trains.  Every timed step still contains ONE guide
    # pair and ONE training step (the pair it launches is consumed by the next step; the first one is launched by the warm-up).
    prefetch = amd.GuidePrefetcher(dev) if (guide is not None and not args.no_guide_prefetch and not args.one_stream) else None

    # where in the step the NEXT batch's guide forwards are queued: before level index 3 (P5) of the D phase, beside the latency-bound small
    # levels (same-box A/B, ms per step: at the head of the step 81.8 / 81.9, level 1 82.5, level 2 81.4 / 81.3, level 3 81.0, level 4 81.2)
    guide_at_level = int(os.environ.get("AFI_BENCH_GUIDE_AT_LEVEL", "3"))
    host_ms = {"guide_enqueue": [], "step_enqueue": []}     # host time spent enqueueing (no sync inside): the last steps' values are reported

    def one_step(last=False, solo=False):
        """solo: nothing runs beside the step (the `kernel_alone` leg): a pending prefetched pair is used up, none is launched."""
        if guide is None:
            hr, lr = syn_hr, syn_lr
        elif prefetch is None or (solo and not prefetch.pending):
            hr, lr = guide_pair()
        else:
            if not prefetch.pending:
                prefetch.submit(guide_pair)                        # (first call only)
            hr, lr = prefetch.take()
            th0 = time.perf_counter()
            step.on_d_level = None
            if not (last or solo):
                if guide_at_level > 0:                             # ... queued from inside the step, before that level of the D phase
                    step.on_d_level = lambda i: prefetch.submit(guide_pair) if i == guide_at_level else None
                else:
                    prefetch.submit(guide_pair)                    # the next batch's features, beside this step
            host_ms["guide_enqueue"].append((time.perf_counter() - th0) * 1e3)
        th1 = time.perf_counter()
        step.run_step(lr, hr)
        host_ms["step_enqueue"].append((time.perf_counter() - th1) * 1e3)
        return hr, lr

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    log("models built; warm-up (the first guide forward lets MIOpen pick/compile its kernels)")
    if guide is not None and args.warmup:
        guide(images_half)
        torch.cuda.synchronize()
        log("guide forward (half-size image) done")
        guide(images)
        torch.cuda.synchronize()
        log("guide forward (full-size image) done")
    for i in range(args.warmup):
        hr, lr = one_step()
        torch.cuda.synchronize()
        log(f"warm-up step {i + 1}/{args.warmup} done")
    if args.warmup:
        assert [tuple(t.shape[2:]) for t in hr] == hr_shapes and [tuple(t.shape[2:]) for t in lr] == lr_shapes, \
            ([t.shape for t in hr], [t.shape for t in lr])
    sync()
    # The timed region carries no event records unless --profile-timed is given: the per-launch HIP events of the roofline leg (two per GEMM
    # launch on the launch stream) are taken over a REPEAT of the same K steps right behind it, same inputs, same schedule.
    if rank == 0 and args.profile_timed:
        lib.afi_profile_enable(1)
    t0 = time.perf_counter()
    # K guide pairs inside the K timed steps either way: after a warm-up the first step consumes the pair the warm-up launched and every
    # step, the last included, launches one; with --warmup 0 (the profiled runs: rocprofv3 then sees exactly the timed launches) the first
    # step launches its own pair first and the last one launches none
    for i in range(args.steps):
        one_step(last=(args.warmup == 0 and i == args.steps - 1))
    elapsed_rank = time.perf_counter() - t0               # this rank's own steps (before the closing barrier): per_rank_images_per_s
    sync()
    elapsed = time.perf_counter() - t0
    log(f"timed region done: {args.steps} steps in {elapsed:.3f} s")
    host_timed = {k: list(v[-args.steps:]) for k, v in host_ms.items()}      # (the profiled repeat below appends its own)
    profiled_ms_per_step = None
    if not args.profile_timed:
        if rank == 0:
            lib.afi_profile_enable(1)
        step.measure_comm = dist is not None               # events either side of each exchange's wait, on the stream that waits (last step's are kept)
        t1 = time.perf_counter()
        for i in range(args.steps):
            one_step(last=(args.warmup == 0 and i == args.steps - 1))
        sync()
        profiled_ms_per_step = (time.perf_counter() - t1) / args.steps * 1e3
        if dist is not None:
            ex = step.comm_exposure()
            hb = ex.get("host_blocked_ms", {})
            te = torch.tensor([ex.get("D", 0.0), ex.get("G", 0.0), hb.get("D", 0.0), hb.get("G", 0.0)], device=dev, dtype=torch.float64)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            comm["comm_exposed_ms"] = {"D": round(float(te[0]), 4), "G": round(float(te[1]), 4), "total": round(float(te[:2].sum()), 4),
                                       "note": "device time the consuming stream waited for each gradient exchange in the last step of the "
                                               "profiled repeat (events either side of the wait; max over ranks); blocking exchanges "
                                               "(overlap_comm false) are exposed whole"}
            comm["host_blocked_ms"] = {"D": round(float(te[2]), 4), "G": round(float(te[3]), 4),
                                       "note": "host time inside the wait: ~0 under nccl (the wait orders streams); under gloo the host's lead "
                                               "over the device plus the exchange -- not an exposure"}
        step.measure_comm = False
        log(f"profiled repeat done: {profiled_ms_per_step:.2f} ms/step with the event brackets on")
    def overlap_ab_leg(emit_on_timeout=None):
        """The other setting of overlap_comm on the same engine, same inputs, same K (never the headline): the first multi-GPU run measures both.
        It is the one leg of an N > 1 run that issues collectives no multi-GPU box has run yet, so for world > 1 it is the LAST thing every rank does,
        behind a watchdog: if it has not finished within AFI_BENCH_AB_TIMEOUT_S (default 90) the rank leaves -- rank 0 after printing the line it
        had already completed (`emit_on_timeout`), with the time-out recorded in comm.overlap_ab -- instead of taking the measured headline down."""
        if dist is None or os.environ.get("AFI_BENCH_OVERLAP_AB", "1") == "0":
            return
        import threading
        other = not step.overlap_comm
        ab = {("overlapped" if step.overlap_comm else "blocking") + "_ms_per_step": round(elapsed / args.steps * 1e3, 3)}
        comm["overlap_ab"] = ab

        def bail():
            ab["error"] = "timed out: abandoned (the headline above was measured before this leg started)"
            if emit_on_timeout is not None:
                emit_on_timeout()
            os._exit(0)
        dog = threading.Timer(float(os.environ.get("AFI_BENCH_AB_TIMEOUT_S", "90")), bail) if world > 1 else None
        if dog is not None:
            dog.daemon = True
            dog.start()
        try:
            step.overlap_comm = other
            one_step(); sync()
            t2 = time.perf_counter()
            for i in range(args.steps):
                one_step(last=(args.warmup == 0 and i == args.steps - 1))
            sync()
            tm = torch.tensor([time.perf_counter() - t2], device=dev, dtype=torch.float64)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            ab[("overlapped" if other else "blocking") + "_ms_per_step"] = round(float(tm.item()) / args.steps * 1e3, 3)
        except Exception as e:                             # (recorded, not fatal: the headline is already measured)
            ab["error"] = f"{type(e).__name__}: {e}"[:300]
        finally:
            step.overlap_comm = not other
            if dog is not None:
                dog.cancel()
        log(f"overlap_comm A/B: {ab}")

    if world == 1:
        overlap_ab_leg()                                   # (--dist-world-1: the engine is released before the single-GPU legs below)
    if rank == 0:
        lib.afi_profile_enable(0)
        if os.environ.get("AFI_PROFILE_DUMP"):            # per-launch CSV (shape, split, ms) for offline analysis
            lib.afi_profile_dump(os.environ["AFI_PROFILE_DUMP"].encode())
    try:
        metrics = step.metrics()                  # also the finite-loss check (_detect_anomaly)
        losses_finite = True
    except FloatingPointError as e:               # the reference aborts here too; the timing is still reported, flagged
        log(f"WARNING: {e}")
        metrics, losses_finite = step.metrics(check_finite=False), False
    params_identical = None
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # data-parallel invariant (stage1_trainer.py:80-89 + the all-reduce): after K steps every rank holds the same G and D weights
        params_identical = identical_across_ranks(dist, torch, [torch.cat([p.detach().reshape(-1) for p in m.parameters()]) for m in (G, D)])
        comm["per_rank_images_per_s"] = per_rank_rates(dist, torch, B * args.steps, elapsed_rank, dev, world)
        assert params_identical, "parameters differ across ranks after the timed steps (all-reduce / broadcast broken)"
    if rank != 0:
        if dist is not None:
            overlap_ab_leg()                               # (rank 0 joins it once its line is complete, below)
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (HIP events recorded by the library on the launch stream)
    kinds = []
    for k in range(lib.afi_profile_num_kinds()):
        out3 = (C.c_double * 3)()
        _lib.check(lib.afi_profile_get(k, out3), "afi_profile_get")
        if out3[0] > 0:
            kinds.append({"kernel": lib.afi_profile_kind_name(k).decode(), "launches": int(out3[0]), "ms_total": out3[1],
                          "avg_us": out3[1] / out3[0] * 1e3, "tflops": out3[2] / (out3[1] * 1e-3) / 1e12 if out3[1] > 0 else 0.0,
                          "flop_total": out3[2]})
    kinds.sort(key=lambda r: -r["ms_total"])
    dom = kinds[0]
    # The timed steps run the D phase (and G's backward) on two streams, so the HIP-event duration of a launch above includes the time it
    # shares the chip with the other stream's kernels.  The same kernel with the chip to itself: two extra steps with the overlaps off.
    kernel_alone = None
    if world == 1 and step.overlap_d and os.environ.get("AFI_BENCH_OTHER_DTYPES", "1") != "0":     # (the profiled runs of tools/prof_r02.sh switch the extra legs off)
        og = step.overlap_g
        step.overlap_d = step.overlap_g = False
        one_step(solo=True); torch.cuda.synchronize()
        lib.afi_profile_enable(1)
        for _ in range(2):
            one_step(solo=True)
        torch.cuda.synchronize()
        lib.afi_profile_enable(0)
        for k in range(lib.afi_profile_num_kinds()):
            if lib.afi_profile_kind_name(k).decode() == dom["kernel"]:
                o3 = (C.c_double * 3)()
                _lib.check(lib.afi_profile_get(k, o3), "afi_profile_get")
                if o3[0] > 0:
                    kernel_alone = {"launches": int(o3[0]), "avg_launch_us": o3[1] / o3[0] * 1e3, "achieved": o3[2] / (o3[1] * 1e-3) / 1e12}
        step.overlap_d, step.overlap_g = True, og
    # the opt-in bf16 arithmetic on the same engine, same inputs (N = 1 only; never the headline value): 1 warm-up + the same K steps each
    other_dtypes = None
    if world == 1 and args.dtype is None and os.environ.get("AFI_BENCH_OTHER_DTYPES", "1") != "0":
        other_dtypes = {}
        for dt in [d for d in ("fp32", "f16x3", "bf16x6", "bf16x3", "bf16") if d != run_dtype]:
            step.set_dtype(dt)
            one_step()
            torch.cuda.synchronize()
            lib.afi_profile_enable(1)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                one_step()
            torch.cuda.synchronize()
            el = time.perf_counter() - t1
            lib.afi_profile_enable(0)
            gk = []
            for k in range(lib.afi_profile_num_kinds()):
                o3 = (C.c_double * 3)()
                _lib.check(lib.afi_profile_get(k, o3), "afi_profile_get")
                name = lib.afi_profile_kind_name(k).decode()
                if o3[0] > 0 and ("gemm_nt" in name or "gemm_tn" in name):
                    tf = o3[2] / (o3[1] * 1e-3) / 1e12
                    gk.append({"kernel": name, "launches": int(o3[0]), "ms_total": round(o3[1], 3), "tflops": round(tf, 1),
                               "frac_of_peak": round(tf / gemm_peak(dt), 4), "peak": round(gemm_peak(dt), 1)})
            try:
                fin = all(v == v and abs(v) != float("inf") for v in step.metrics().values())
            except FloatingPointError:
                fin = False
            other_dtypes[dt] = {"ms_per_step": el / args.steps * 1e3, "images_per_s": B * args.steps / el, "losses_finite": fin, "winograd_gemm_kernels": gk}
            log(f"dtype {dt}: {el / args.steps * 1e3:.1f} ms/step")
        step.set_dtype(run_dtype)

    gemm_ms = sum(r["ms_total"] for r in kinds)
    gemm_flop = sum(r["flop_total"] for r in kinds)
    traffic = committed_traffic("traffic_dominant_kernel.json", dom["kernel"])
    # the Winograd GEMMs' own roof: the dense bf16 MFMA peak over the bf16 MFMAs issued per fp32-equivalent product (6 / 3 / 1), or the
    # fp32 MFMA peak; every other kernel multiplies on the fp32 MFMA
    dom_peak = kind_peak(dom["kernel"], run_dtype)
    peak_s = sum(r["flop_total"] / (kind_peak(r["kernel"], run_dtype) * 1e12) for r in kinds)        # seconds the step's products take at each kernel's own peak
    roofline = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["tflops"], "peak": dom_peak, "unit": "TFLOP/s",
                "frac": dom["tflops"] / dom_peak, "traffic": traffic, "launches": dom["launches"], "avg_launch_us": dom["avg_us"],
                "share_of_step_time": dom["ms_total"] / (elapsed * 1e3),
                # `achieved` above is over the timed region, where the D phase runs on two streams: a launch's duration includes the time it
                # shares the chip with the other stream's kernels (sum of durations > wall time).  The kernel with the chip to itself:
                "kernel_alone": (dict(kernel_alone, frac=kernel_alone["achieved"] / dom_peak,
                                      note="two extra steps on ONE stream (overlap_d / overlap_g / guide prefetch off), same HIP-event "
                                           "brackets; profiles/r06 holds both traces")
                                 if kernel_alone else None),
                # fp32-equivalent rate of the dominant kernel against the fp32 MFMA roof it replaces (> 1 is the point of the emulated forms)
                "achieved_over_fp32_mfma_peak": dom["tflops"] / PEAK_FP32_MFMA_TFLOPS,
                "peak_note": PEAK_NOTE,
                # the whole step in EXECUTED matrix-core products (what the GEMM launches multiplied, Winograd-domain for the big convs): the time
                # they need at each kernel's own peak over the wall time -- the one <= 1 "achieved roofline" figure of the step
                "frac_step_executed": peak_s / elapsed,
                "all_gemm_kernels": {"tflops": gemm_flop / (gemm_ms * 1e-3) / 1e12, "frac": peak_s / (gemm_ms * 1e-3),
                                     "share_of_step_time": gemm_ms / (elapsed * 1e3)},
                "per_kernel": [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k != "flop_total"} for r in kinds],
                "measured_over": ("the timed region itself (--profile-timed)" if args.profile_timed else
                                  f"a repeat of the timed region's {args.steps} steps right behind it with the library's per-launch HIP events on "
                                  f"({profiled_ms_per_step:.2f} ms/step there): the timed region carries no event records")}

    # algorithmic work of one step per image: D fwd 4x hr px + D bwd on 2x hr px; G fwd 2x lr px + G bwd 1x lr px
    hr_px = sum(h * w for h, w in hr_shapes)
    lr_px = sum(h * w for h, w in lr_shapes)
    flop_img = (2 * D_FWD_FLOP_PER_PX + 2 * D_FWDBWD_DETACHED_FLOP_PER_PX) * hr_px + \
        (2 * G_FWD_FLOP_PER_INPX + 2 * G_FWD_FLOP_PER_INPX - 1_179_648) * lr_px
    n_img = world * B * args.steps
    # SURVEY 8(d)'s algorithmic (direct-convolution) FLOP count of the step over wall time and peak.  It can pass 1: the big 3x3 convs
    # run in Winograd form, which executes 2.25x (F(2x2,3x3)) / 4x (F(4x4,3x3)) fewer multiplies than the count assumes
    roofline["algorithmic_over_peak"] = flop_img * B * args.steps / elapsed / 1e12 / PEAK_FP32_MFMA_TFLOPS
    roofline["winograd_multiply_reduction"] = ("4x F(4x4,3x3): data / weight gradients, forwards without a backward, D's blocks 1 and 2; "
                                               "2.25x F(2x2,3x3): the other forwards a backward follows")
    line = {
        "metric": "stage1_G+D_step_images_per_s", "value": n_img / elapsed, "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE_NOTE[run_dtype], "data": "synthetic",
        "backend": (args.backend if dist is not None else None), "comm": comm,
        "config": {"workload": f"configs[1]: stage-1 G+D step, R-50-FPN guide, {B}x3x800x1333 synthetic images/GPU, P2..P6",       # (< 100 chars)
                   "global_batch": world * B, "parallelism": f"dp{world}",
                   "guide": "r50fpn random-init eval (this library's conv kernels; stem GEMM via hipBLASLt)" if guide is not None else "synthetic-pyramid",
                   "generator": "n_rdb=3",
                   "guide_prefetch": prefetch is not None, "options": args.option or None,
                   "host_enqueue_ms": {k: round(sum(v) / max(1, len(v)), 2) for k, v in host_timed.items() if v},
                   "reuse_generator_forward": True},
        "algorithmic_tflop_per_image": flop_img / 1e12,
        "step_tflops_per_gpu": flop_img * B * args.steps / elapsed / 1e12,
        # direct-convolution FLOP count of the step over the dense fp32 MFMA peak.  The big 3x3 convs run in Winograd form,
        # which EXECUTES 2.25x / 4x fewer products than this count, so the ratio can pass 1; `roofline` below is in executed products of
        # the dominant kernel (the batched Winograd GEMM) against that kernel's own roof.
        "step_algorithmic_tflops_over_fp32_mfma_peak": flop_img * B * args.steps / elapsed / 1e12 / PEAK_FP32_MFMA_TFLOPS,
        "conv_algorithm": CONV_NOTE,
        "roofline": roofline,
        # opt-in arithmetic of the big convolutions, same engine / inputs / K (afi_ctx_set_compute_dtype; tolerances: tests/test_gpu_bf16.py).
        # Not the headline: the reference is fp32-only.  Their GEMM kernels are priced against the dense bf16 MFMA peak.
        "other_dtypes": other_dtypes,
        "params_identical_across_ranks": params_identical,
        "losses_last_step": {k: round(v, 5) for k, v in metrics.items()}, "losses_finite": losses_finite,
    }
    log("AF-interpolator micro-benchmark")
    micro = not args.no_interp and world == 1             # single-GPU metrics: reported on the N=1 line only
    if micro:
        # release the stage-1 engine first: with its tens of GB of workspaces mapped, every kernel launch of the small-map
        # legs cost the host ~16 us instead of ~5 us and the (launch-bound) config-1 leg measured the host, not the GPU
        del step, G, D
        if guide is not None:
            del guide
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
        line["af_interpolator"] = {"metric": "AF-interpolator fwd+bwd feature-Mpix/s (256ch P5->P4)",
                                   "cfg1": interp_bench(amd, torch, 1, 25, 34), "batch16": interp_bench(amd, torch, 16, 25, 34, iters=20, warmup=5)}
        # SURVEY 8(d) sweep: batch sizes at the config-1 map, and the stage-3 FPN call-site maps (25x42, 50x84, 100x168) at N = 1
        sweep = {}
        for (n_, h_, w_) in ((2, 25, 34), (8, 25, 34), (1, 25, 42), (1, 50, 84), (1, 100, 168)):
            r_ = interp_bench(amd, torch, n_, h_, w_, iters=20, warmup=5)          # (eager or hipGraph replay, whichever is faster: as cfg1)
            sweep[f"{n_}x256x{h_}x{w_}"] = {"ms": round(r_["ms"], 4), "out_mpix_per_s": round(r_["out_mpix_per_s"], 3), "tflops": round(r_["tflops"], 1)}
        line["af_interpolator"]["sweep"] = sweep
        # BASELINE.json's FIRST metric where the driver's record keeps it (it preserves the `roofline` and `cpu_baseline` objects whole)
        c1 = line["af_interpolator"]["cfg1"]
        line["roofline"]["af_interpolator_cfg1"] = {
            "ms": round(c1["ms"], 4), "out_mpix_per_s": round(c1["out_mpix_per_s"], 3), "in_mpix_per_s": round(c1["in_mpix_per_s"], 3), "tflops": round(c1["tflops"], 2),
            "frac_of_fp32_mfma_peak": round(c1["frac_of_fp32_mfma_peak"], 4), "launch": c1["launch"],
            "ms_weights_cached": None if c1.get("ms_weights_cached") is None else round(c1["ms_weights_cached"], 4),
            "dominant_kernel": None if not c1["roofline"] else {k: c1["roofline"][k] for k in ("kernel", "achieved", "peak", "frac", "launches", "avg_launch_us")},
            "note": "AF interpolator forward + full backward, 1x256x25x34 -> 1x256x50x68 through the C-ABI, SURVEY 8(d) metric 1: 48.98 GFLOP algorithmic per call.  "
                    "ms = the stand-alone call (weights transformed inside it); ms_weights_cached = the same call as the stage-1 engine issues it inside a step, where "
                    "the context's per-phase cache already holds the weight images (built once per optimizer step, shared by the five levels)"}
    if micro:
        log("FPN_AFIGAN top-down merge (SURVEY 8f row 1)")
        line["fpn_topdown"] = fpn_bench(amd, torch)
        line["pafpn"] = fpn_bench(amd, torch, pafpn=True)
        # BASELINE configs[4] names this pyramid "bf16": the same forward + backward under each arithmetic setting of the big GEMMs
        line["pafpn"]["ms_by_dtype"] = {}
        for dt in ("fp32", "f16x3", "bf16x6", "bf16x3", "bf16"):
            with amd.compute_dtype(dt):
                line["pafpn"]["ms_by_dtype"][dt] = round(fpn_bench(amd, torch, iters=5, warmup=2, pafpn=True)["ms"], 3)
        line["bifpn_inference"] = bifpn_bench(amd, torch)
        line["bifpn_training"] = bifpn_train_bench(amd, torch)
        line["stage2_adversarial"] = stage2_bench(amd, torch)
        from tools import dual_scale_bench                       # the mapper's uint8 resize pair + normalise/pad
        line["dual_scale_data_path"] = dual_scale_bench.run(iters=100, warm=10, cpu_iters=3)
    log("CPU baseline (oracle) on the host cores")
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(torch, B)
        line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
        if "af_interpolator" in line:                      # metric 1, GPU beside CPU, in the object the driver keeps
            c1 = line["af_interpolator"]["cfg1"]
            cb = line["cpu_baseline"]
            cb["af_interpolator_gpu_ms"] = round(c1["ms"], 4)
            cb["af_interpolator_gpu_out_mpix_per_s"] = round(c1["out_mpix_per_s"], 3)
            cb["af_interpolator_gpu_over_cpu"] = round(c1["out_mpix_per_s"] / cb["af_interpolator_out_mpix_per_s"], 1)
    if world > 1:
        overlap_ab_leg(emit_on_timeout=lambda: print(json.dumps(line), flush=True))
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
