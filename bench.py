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

PEAK_FP32_MFMA_TFLOPS = 157.3           # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0          # MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA", dense


def gemm_peak(dtype):
    """Roof of the Winograd-domain GEMM kernels in fp32-equivalent TFLOP/s (2*M*N*K counted once per product)."""
    return {"fp32": PEAK_FP32_MFMA_TFLOPS, "bf16x6": PEAK_BF16_MFMA_TFLOPS / 6, "f16x3": PEAK_BF16_MFMA_TFLOPS / 3, "bf16x3": PEAK_BF16_MFMA_TFLOPS / 3,
            "bf16": PEAK_BF16_MFMA_TFLOPS}[dtype]


def kind_peak(name, run_dtype):
    """Roof of one profiled kernel kind (afi_profile_kind_name): the dense bf16 / f16 MFMA peak over the MFMAs issued per fp32-equivalent product,
    or the fp32 MFMA peak.  The f16x3 GEMMs issue three; the small-map kernels ("bf16x6 operands") always six; the bf16 Winograd GEMMs what
    the run's dtype says."""
    if "f16x3" in name:
        return gemm_peak("f16x3")
    if "bf16x6 operands" in name:
        return gemm_peak("bf16x6")
    if "bf16" in name:
        return gemm_peak(run_dtype if run_dtype in ("bf16x6", "bf16x3", "bf16") else "bf16x6")
    return PEAK_FP32_MFMA_TFLOPS
G_FWD_FLOP_PER_INPX = 19_206_144        # SURVEY.md 8(d) / BASELINE.md section 3
D_FWD_FLOP_PER_PX = 30_689_280
D_FWDBWD_DETACHED_FLOP_PER_PX = 89_708_544


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
    ap.add_argument("--debug-nt-ablation", type=int, default=0, help="afi_debug_set_nt_ablation(N): kernel A/B switches of csrc/igemm.hip (tools only)")
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


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher (reference: stage1_train.py:52-59, detectron2 `launch(main, num_gpus, ...)`): THIS
    process never touches the GPU -- no HIP call, no torch.cuda query -- it starts N children, one rank per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (what torch.distributed.run would set), waits for them and relays
    rank 0's stdout (the ONE JSON line).  A child that fails takes the job down: the others are terminated (by their exact PIDs) and the
    exit code is non-zero."""
    import socket
    import subprocess
    n = args.gpus
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL between processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    rc = 0
    pending = set(range(n))
    while pending and rc == 0:                                     # (rank 0 prints one short line at the very end: its pipe cannot fill up)
        for r in sorted(pending):
            c = procs[r].poll()
            if c is not None:
                pending.discard(r)
                if c != 0:
                    rc = c if c > 0 else 1
                    print(f"[bench] rank {r} exited with code {c}: stopping the other ranks", file=sys.stderr, flush=True)
        if pending and rc == 0:
            time.sleep(0.2)
    for pr in procs:
        if pr.poll() is None:
            pr.terminate()                                         # only reached when a rank failed: the rest would wait at a barrier forever
    out0 = procs[0].communicate()[0]
    for pr in procs[1:]:
        try:
            pr.wait(timeout=30)
        except subprocess.TimeoutExpired:
            pr.kill()
    sys.stdout.write(out0 or "")
    sys.stdout.flush()
    raise SystemExit(rc)


def allreduce_alone(dist, torch, bufs, dev, world, reps=5):
    """Each gradient exchange of a step with nothing beside it: {tag: {bytes, ms (max over ranks), bus_gb_per_s}} for the flat buffers
    in `bufs`.  Zeroes the buffers afterwards.  Used by the real run (device buffers) and by --rehearse-launch
    (CPU buffers of the same sizes)."""
    cuda = dev is not None and dev.type == "cuda"
    out = {}
    for tag, buf in bufs.items():
        for _ in range(2):
            dist.all_reduce(buf)
        if cuda:
            torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            dist.all_reduce(buf)
        if cuda:
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        tm = torch.tensor([ms], device=dev, dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        nbytes = buf.numel() * 4
        out[tag] = {"bytes": nbytes, "ms": round(float(tm.item()), 4),
                    "bus_gb_per_s": round(2.0 * (world - 1) / world * nbytes / (float(tm.item()) * 1e-3) / 1e9, 2)}
        buf.zero_()
    return out


def identical_across_ranks(dist, torch, tensors):
    """Data-parallel invariant (stage1_trainer.py:80-89 + the all-reduce): the same values on every rank.  One fp64 checksum per
    tensor, MIN and MAX over the ranks compared bit for bit."""
    chk = torch.stack([t.detach().reshape(-1).double().sum() for t in tensors])
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool(torch.equal(lo, hi))


def per_rank_rates(dist, torch, images, seconds, dev, world):
    """[images/s of rank 0, rank 1, ...] from every rank's own clock around the timed steps (the job's `value` uses the MAX time)."""
    mine = torch.tensor([images / seconds], device=dev, dtype=torch.float64)
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    return [round(float(t.item()), 3) for t in got]


D_GRAD_FLOATS, G_GRAD_FLOATS = 15_352_324, 7_834_624        # flat gradient buffers of the reference-width D and G (61.4 MB, 31.3 MB)


def rehearse_launch(args, world, rank):
    """--rehearse-launch: everything bench.py does AROUND the GPU work for an N-rank job, on CPU tensors over gloo: rendezvous, the `comm`
    object (backend, world size as the group reports it, the two exchanges alone on buffers of the real sizes: the same
    `allreduce_alone` the real run calls), K "steps" whose only content is the two all-reduces of a step in the engine's order (blocking,
    or asynchronous and waited for where the engine waits: `--overlap-comm`), barrier + MAX-over-ranks timing, per-rank images/s, the
    cross-rank identity check (`identical_across_ranks`, as the real run), ONE JSON line from rank 0.  Reports no metric."""
    import torch
    import torch.distributed as dist
    if os.environ.get("AFI_BENCH_REHEARSE_FAIL_RANK") == str(rank):   # fault injection for the test of the failure path
        raise SystemExit(3)
    B = args.batch_per_gpu
    comm, same, rates, ok = None, None, None, True
    params = [torch.full((1 << 12,), 1.0), torch.full((1 << 10,), 2.0)]        # "G" and "D": rank 0's values everywhere after the broadcast
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        for p_ in params:
            p_.add_(float(rank))                                   # different per rank before the broadcast (DDP ctor semantics: rank 0's win)
            dist.broadcast(p_, src=0)
        small = os.environ.get("AFI_BENCH_REHEARSE_SMALL", "1") != "0"       # 1/64 of the real sizes: 8 ranks on this container's 8 cores
        bufs = {"D": torch.zeros(D_GRAD_FLOATS // (64 if small else 1)), "G": torch.zeros(G_GRAD_FLOATS // (64 if small else 1))}
        overlap = bool(args.overlap_comm) if args.overlap_comm is not None else True        # (gloo: the engine's default is on)
        comm = {"backend": dist.get_backend(), "world_size_reported": dist.get_world_size(), "rank0_device": "cpu (rehearsal)",
                "overlap_comm": overlap, "allreduce_alone": allreduce_alone(dist, torch, bufs, None, world, reps=2)}
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            for tag in ("D", "G"):
                bufs[tag].fill_(float(rank + 1))
            wd = dist.all_reduce(bufs["D"], async_op=overlap)      # D's exchange behind the D phase ...
            wg = dist.all_reduce(bufs["G"], async_op=overlap)      # ... G's behind G's last backward pass, issued before D's is waited for
            for w_ in (wd, wg):
                if w_ is not None:
                    w_.wait()
            ok = ok and all(bool((bufs[tag] == world * (world + 1) / 2).all()) for tag in ("D", "G"))
            for p_, tag in zip(params, ("G", "D")):
                p_.sub_(1e-3 / world * bufs[tag][:p_.numel()])     # "SGD" on the averaged gradient: identical on every rank
        mine = time.perf_counter() - t0
        dist.barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        rates = per_rank_rates(dist, torch, B * args.steps, mine, None, world)
        same = identical_across_ranks(dist, torch, params)
        dist.barrier()
        dist.destroy_process_group()
        elapsed = float(el.item())
    else:
        elapsed = 0.0
    if rank == 0:
        print(json.dumps({"metric": "launch_rehearsal (no GPU work, no measurement)", "value": None, "n_gpus": world, "steps": args.steps,
                          "backend": "gloo" if world > 1 else None, "comm": comm, "allreduce_sum_ok": ok,
                          "params_identical_across_ranks": same, "per_rank_images_per_s": rates, "max_over_ranks_s": elapsed,
                          "config": {"global_batch": world * B, "parallelism": f"dp{world}"},
                          "spawned_by_bench": os.environ.get("AFI_BENCH_SPAWNED") == "1"}), flush=True)
    raise SystemExit(0 if ok and same is not False else 1)


KERNEL_TOKENS = ("gemm_nt_f16x3", "gemm_tn_f16x3", "gemm_nt", "gemm_tn", "pix_gemm_wk6", "pix_gemm_wk", "pix_gemm", "wgrad6", "wgrad")


def committed_traffic(fname, dom_kernel):
    """HBM traffic of a dominant kernel cannot be read live (PMC counters need their own rocprofv3 passes): report the committed
    measurement of the same command when there is one (the newest profiles/rNN/<fname>), else None.  The record names the kernel it was
    taken on -- another dominant kernel nulls it -- and is tied to the kernel SOURCE it was measured on by a sha256: a later edit of that
    file nulls it too."""
    for rnd in ("r05", "r04", "r03", "r02", "r01"):
        tpath = os.path.join(ROOT, "profiles", rnd, fname)
        if not os.path.exists(tpath):
            continue
        try:
            tj = json.load(open(tpath))
            same = [t for t in KERNEL_TOKENS if t in dom_kernel][:1] == [t for t in KERNEL_TOKENS if t in str(tj.get("kernel", ""))][:1]
            src = tj.get("kernel_source")
            fresh = bool(src) and file_sha256(os.path.join(ROOT, src)) == tj.get("kernel_source_sha256")
            if not same or not fresh:
                return None                                # measured on another kernel, or on another version of this one: stale
            return {"hbm_bytes_per_launch": tj["hbm_bytes_per_launch"], "algorithmic_bytes_per_launch": tj.get("algorithmic_bytes_per_launch"),
                    "measured_at": tj.get("measured_at"), "kernel_source": src, "kernel_source_sha256": tj.get("kernel_source_sha256"),
                    "held_clock_ghz": tj.get("held_clock_ghz"), "mfma_busy_at_held_clock": tj.get("mfma_busy_at_held_clock"),
                    "source": f"profiles/{rnd}/{fname} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command; FETCH x2 per the gfx950 correction)"}
        except (OSError, ValueError, KeyError):
            return None
    return None


def interp_bench(amd, torch, N, H, W, iters=50, warmup=10, graph=True):
    """Generator(n_rdb=3) forward + full backward (input grad + all weight grads, loss = out.sum()) through the C-ABI.
    Timed twice: eager launches, and the same call sequence captured once into a hipGraph and replayed (no host launch cost;
    the library's fork/join onto its side stream is plain event record/wait, so it captures)."""
    from afigan_amd import _lib, ops
    lib = _lib.load()
    torch.manual_seed(0)
    G = amd.Generator(n_residual_dense_blocks=3).cuda()
    x = ops.pixel_major(torch.randn(N, 256, H, W, generator=torch.Generator().manual_seed(0)).cuda())
    params = G._ordered_params()
    prm, keep = G._param_struct(params)
    grads = [torch.zeros_like(p) for p in params]
    gst, _ = G._param_struct(grads, already_packed=True)
    nf = lib.afi_generator_fwd_ws_floats(256, 32, 3, N, H, W)
    nb = lib.afi_generator_bwd_ws_floats(256, 32, 3, N, H, W)
    ws = torch.empty(nf, device="cuda")
    sc = torch.empty(nb, device="cuda")
    out = ops.new_pixel_major(N, 256, 2 * H, 2 * W, "cuda")
    dout = ops.new_pixel_major(N, 256, 2 * H, 2 * W, "cuda")
    dout.fill_(1.0)
    dx = ops.new_pixel_major(N, 256, H, W, "cuda")
    def one():
        st = ops.stream_ptr()
        _lib.call("afi_generator_fwd", C.byref(prm), ops.view_of(x), N, H, W, ops.view_of(out), C.c_void_p(ws.data_ptr()), nf, st)
        _lib.call("afi_generator_bwd", C.byref(prm), C.byref(gst), ops.view_of(x), N, H, W, C.c_void_p(ws.data_ptr()),
                  C.c_void_p(dout.data_ptr()), C.c_void_p(dx.data_ptr()), C.c_void_p(sc.data_ptr()), nb, st)

    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        one()
    t_enq = (time.perf_counter() - t0) / iters             # host time to enqueue one iteration (no sync)
    torch.cuda.synchronize()
    dt_eager = dt = (time.perf_counter() - t0) / iters
    mode = "eager"
    dt_graph = None
    if graph:
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                one()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                g.replay()
            torch.cuda.synchronize()
            dt_graph = (time.perf_counter() - t0) / iters
            if dt_graph < dt:
                dt, mode = dt_graph, "hipGraph replay"
        except Exception as e:      # capture is an optimisation of the launch path only
            log(f"  hipGraph capture unavailable: {type(e).__name__}: {e}")
    # the same call as the stage-1 engine issues it INSIDE a step: the context holds the per-phase cache of transformed weights / small-map
    # weight images (afi_ctx_set_wino_weight_cache: built by the first call after an optimizer step, shared by every later call of the phase
    # -- five levels, forward and backward), so a call past the first finds its images built.  Reported beside the stand-alone figure,
    # never instead of it.
    dt_cached = None
    if graph:
        try:
            cx = _lib.current_ctx()
            nfl = 32 * 1024 * 1024
            wc = torch.empty(nfl, device="cuda")
            _lib.call("afi_ctx_set_wino_weight_cache", cx.handle, C.c_void_p(wc.data_ptr()), nfl)
            try:
                for _ in range(3):
                    one()
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2):
                    one()
                for _ in range(3):
                    g2.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(iters):
                    g2.replay()
                torch.cuda.synchronize()
                dt_cached = (time.perf_counter() - t0) / iters
            finally:
                torch.cuda.synchronize()
                _lib.call("afi_ctx_set_wino_weight_cache", cx.handle, C.c_void_p(None), 0)
        except Exception as e:
            log(f"  weight-cache timing unavailable: {type(e).__name__}: {e}")
    out_px = N * 4 * H * W
    flop = 3 * G_FWD_FLOP_PER_INPX * N * H * W
    # live roofline of this workload's dominant kernel: a short run with the library's HIP-event brackets on (not the timed run above:
    # the brackets add two event records per launch)
    lib.afi_profile_enable(1)
    prof_iters = 20
    for _ in range(prof_iters):
        one()
    torch.cuda.synchronize()
    lib.afi_profile_enable(0)
    kinds = []
    for k in range(lib.afi_profile_num_kinds()):
        out3 = (C.c_double * 3)()
        _lib.check(lib.afi_profile_get(k, out3), "afi_profile_get")
        if out3[0] > 0:
            kinds.append({"kernel": lib.afi_profile_kind_name(k).decode(), "launches_per_iter": out3[0] / prof_iters, "us_per_iter": out3[1] * 1e3 / prof_iters,
                          "avg_launch_us": out3[1] * 1e3 / out3[0], "tflops": out3[2] / (out3[1] * 1e-3) / 1e12 if out3[1] > 0 else 0.0})
    kinds.sort(key=lambda r: -r["us_per_iter"])
    # a kernel's own roof (kind_peak): the small-map kernels multiply six bf16 MFMAs per fp32-equivalent product, the f16x3 GEMMs three, the others use the fp32 MFMA
    for r in kinds:
        r["peak"] = kind_peak(r["kernel"], _lib.current_ctx().dtype)
        r["frac"] = r["tflops"] / r["peak"]
    roof = None
    if kinds:
        d0 = kinds[0]
        roof = {"bound": "mfma", "kernel": d0["kernel"], "achieved": d0["tflops"], "peak": d0["peak"], "unit": "TFLOP/s",
                "frac": d0["frac"], "launches": d0["launches_per_iter"], "avg_launch_us": d0["avg_launch_us"],
                "achieved_over_fp32_mfma_peak": d0["tflops"] / PEAK_FP32_MFMA_TFLOPS,
                "gemm_launches_per_iter": sum(r["launches_per_iter"] for r in kinds), "gemm_us_per_iter": sum(r["us_per_iter"] for r in kinds),
                "per_kernel": [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()} for r in kinds],
                "traffic": committed_traffic("traffic_cfg1_dominant_kernel.json", d0["kernel"]) if (N, H, W) == (1, 25, 34) else None}
    return {"roofline": roof, "shape": f"{N}x256x{H}x{W}->{N}x256x{2 * H}x{2 * W}", "launch": mode, "ms": dt * 1e3, "ms_eager": dt_eager * 1e3,
            "ms_graph": None if dt_graph is None else dt_graph * 1e3, "ms_host_enqueue": t_enq * 1e3,
            "ms_weights_cached": None if dt_cached is None else dt_cached * 1e3,
            "out_mpix_per_s": out_px / dt / 1e6, "in_mpix_per_s": out_px / 4 / dt / 1e6, "tflops": flop / dt / 1e12,
            "frac_of_fp32_mfma_peak": flop / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS}


def fpn_bench(amd, torch, iters=10, warmup=3, pafpn=False):
    """the AFI top-down merge of FPN_AFIGAN (fpn_sr.py:127-165) at stage-3 size, one 800x1344 image:
    res2..res5 = 200x336x256, 100x168x512, 50x84x1024, 25x42x2048 -> p2..p6, forward + backward through the module
    (autograd path: three interpolator calls, fused lateral+add GEMMs, 3x3 output convs, all channels_last).
    pafpn=True: PAFPN_AFIGAN (pafpn_sr.py:147-193), i.e. the same plus the three stride-2 downsample+merge GEMMs."""
    from afigan_amd.fpn_sr import ShapeSpec

    class BottomUp(torch.nn.Module):
        def output_shape(self):
            return {f"res{i + 2}": ShapeSpec(c, s) for i, (c, s) in enumerate(zip([256, 512, 1024, 2048], [4, 8, 16, 32]))}

        def forward(self, feats):
            return feats

    torch.manual_seed(0)
    cls = amd.PAFPN_AFIGAN if pafpn else amd.FPN_AFIGAN
    fpn = cls(BottomUp(), ["res2", "res3", "res4", "res5"], 256, top_block=amd.LastLevelMaxPool()).cuda()
    shapes = [(256, 200, 336), (512, 100, 168), (1024, 50, 84), (2048, 25, 42)]
    feats = {f"res{i + 2}": torch.randn((1, c, h, w), device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
             for i, (c, h, w) in enumerate(shapes)}

    def one():
        out = fpn(feats)
        loss = sum(o.sum() for o in out.values())
        loss.backward()
        for q in list(fpn.parameters()) + list(feats.values()):
            q.grad = None

    for _ in range(warmup):
        one()
    dt = None
    for _rep in range(2):                                   # best of two timed batches (the first one sometimes still pays allocator growth)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            one()
        torch.cuda.synchronize()
        d_ = (time.perf_counter() - t0) / iters
        dt = d_ if dt is None else min(dt, d_)
    g_px = 25 * 42 + 50 * 84 + 100 * 168
    lat = sum(h * w * c for c, h, w in shapes) * 256 * 2
    outc = sum(h * w for _, h, w in shapes) * 256 * 2304 * 2
    down = sum(h * w for _, h, w in shapes[1:]) * 256 * 2304 * 2 if pafpn else 0
    flop = 3 * (g_px * G_FWD_FLOP_PER_INPX + lat + outc + down)
    return {"workload": ("PAFPN_AFIGAN top-down + bottom-up" if pafpn else "FPN_AFIGAN top-down merge") + " fwd+bwd, 1 image 800x1344, R-50 feature shapes", "ms": dt * 1e3, "images_per_s": 1.0 / dt,
            "algorithmic_tflop": flop / 1e12, "tflops": flop / dt / 1e12, "frac_of_fp32_mfma_peak": flop / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS}


def bifpn_bench(amd, torch, iters=10, warmup=3):
    """BiFPN_AFIGAN inference forward (bifpn_sr.py:569-733) for one 896x1408 image (size_divisibility 128),
    Swin-L stage3..5 feature shapes: 7 BiFPN layers, 56 fused separable-conv nodes, 28 interpolator forwards on 7x11 .. 56x88
    maps -- the launch-bound regime; timed eagerly and as one hipGraph replay."""
    class BottomUp(torch.nn.Module):
        _out_feature_strides = {"stage3": 8, "stage4": 16, "stage5": 32}
        _out_feature_channels = {"stage3": 384, "stage4": 768, "stage5": 1536}

        def forward(self, feats):
            return feats

    torch.manual_seed(0)
    net = amd.BiFPN_AFIGAN(BottomUp(), ["stage3", "stage4", "stage5"], 256, 7, norm="SyncBN", top_block=amd.LastLevelP6P7(1536, 256, "SyncBN")).cuda().eval()
    feats = {f"stage{i + 3}": torch.randn((1, c, 112 // 2 ** i, 176 // 2 ** i), device="cuda").contiguous(memory_format=torch.channels_last)
             for i, c in enumerate([384, 768, 1536])}

    def timed(fn):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters

    with torch.no_grad():                                   # inference (with grad mode on the module would build its autograd graph)
        dt_eager = timed(lambda: net(feats))
        dt_graph = None
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                net(feats)
            dt_graph = timed(g.replay)
        except Exception as e:
            log(f"  hipGraph capture unavailable: {type(e).__name__}: {e}")
    g_px = sum(7 * (7 * 2 ** i) * (11 * 2 ** i) for i in range(4))               # 7 layers x (p7, p6, p5, p4 inputs)
    flop = g_px * G_FWD_FLOP_PER_INPX
    best = min(dt_eager, dt_graph) if dt_graph else dt_eager
    return {"workload": "BiFPN_AFIGAN inference forward, 1 image 896x1408, Swin-L stage3..5 shapes (28 interpolator calls)",
            "ms_eager": dt_eager * 1e3, "ms_hipgraph": None if dt_graph is None else dt_graph * 1e3, "images_per_s": 1.0 / best,
            "interpolator_tflop": flop / 1e12, "interpolator_tflops_lower_bound": flop / best / 1e12}


def bifpn_train_bench(amd, torch, iters=5, warmup=2):
    """SURVEY 8(f) row 4, the TRAINING path (bifpn_sr.py:569-733 with batch-statistics norms): forward + backward of BiFPN_AFIGAN in train
    mode for one 896x1408 image, Swin-L stage3..5 feature shapes, loss = sum of the five outputs: 28 interpolator forwards AND backwards
    (input gradients and all weight gradients), 56 separable-conv nodes and 61 training-mode norms, every piece a HIP forward + backward
    behind torch autograd.  Reported: wall time per iteration, the forward / backward split (events), and the GEMM launches of one iteration by
    kernel family (the library's own HIP-event brackets); the per-kernel table of the whole pass is profiles/r06/kernel_stats_bifpn_train_*.csv
    (rocprofv3 over tools/bifpn_train_loop.py).  norm "SyncBN" with one rank is plain batch statistics."""
    class BottomUp(torch.nn.Module):
        _out_feature_strides = {"stage3": 8, "stage4": 16, "stage5": 32}
        _out_feature_channels = {"stage3": 384, "stage4": 768, "stage5": 1536}

        def forward(self, feats):
            return feats

    torch.manual_seed(0)
    net = amd.BiFPN_AFIGAN(BottomUp(), ["stage3", "stage4", "stage5"], 256, 7, norm="SyncBN", top_block=amd.LastLevelP6P7(1536, 256, "SyncBN")).cuda().train()
    feats = {f"stage{i + 3}": torch.randn((1, c, 112 // 2 ** i, 176 // 2 ** i), device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
             for i, c in enumerate([384, 768, 1536])}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]

    def one(timed=False):
        for p_ in net.parameters():
            p_.grad = None
        for f_ in feats.values():
            f_.grad = None
        if timed:
            ev[0].record()
        out = net(feats)
        loss = sum(v.sum() for v in out.values())
        if timed:
            ev[1].record()
        loss.backward()
        if timed:
            ev[2].record()

    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    one(timed=True)
    torch.cuda.synchronize()
    fwd_ms, bwd_ms = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
    # GEMM launches of one iteration by kernel family (the library's HIP-event brackets)
    lib = amd._lib.load()
    lib.afi_profile_enable(1)
    one()
    torch.cuda.synchronize()
    lib.afi_profile_enable(0)
    fam = []
    for k in range(lib.afi_profile_num_kinds()):
        o3 = (C.c_double * 3)()
        amd._lib.check(lib.afi_profile_get(k, o3), "afi_profile_get")
        if o3[0] > 0:
            fam.append({"kernel": lib.afi_profile_kind_name(k).decode(), "launches": int(o3[0]), "ms_total": round(o3[1], 3),
                        "tflops": round(o3[2] / (o3[1] * 1e-3) / 1e12, 1) if o3[1] > 0 else 0.0})
    fam.sort(key=lambda r: -r["ms_total"])
    g_px = sum(7 * (7 * 2 ** i) * (11 * 2 ** i) for i in range(4))               # 7 layers x (p7, p6, p5, p4 inputs)
    flop = 3 * g_px * G_FWD_FLOP_PER_INPX                                         # fwd + dgrad + wgrad of the 28 interpolator calls
    return {"workload": "BiFPN_AFIGAN TRAINING forward + backward, 1 image 896x1408, Swin-L stage3..5 shapes (28 interpolator fwd+bwd, 61 batch-statistics norms)",
            "ms": dt * 1e3, "ms_forward": fwd_ms, "ms_backward": bwd_ms, "images_per_s": 1.0 / dt, "norm": "SyncBN (one rank: plain batch statistics)",
            "interpolator_tflop": flop / 1e12, "interpolator_tflops_lower_bound": flop / dt / 1e12,
            "gemm_ms_per_iteration": round(sum(r["ms_total"] for r in fam), 3), "gemm_kernel_families": fam[:8]}


def stage2_bench(amd, torch, iters=5, warmup=2):
    """the AFI-specific part of one stage-2 iteration (stage2_trainer.py:299-364) for a per-GPU batch of two
    images: guide features at full size (P2..P6 of 800x1344), the AFI detector's FPN features at half size (416x672 input);
    D step (real = nearest-half of the guide feature, fake = FPN feature) + generator-side losses with their backward into the
    FPN features.  The detector itself is detectron2 glue and not part of it."""
    D = amd.Discriminator().cuda()
    adv = amd.Stage2Adversarial(D, base_lr=1e-3)
    g = torch.Generator(device="cuda").manual_seed(0)
    guide = [torch.randn((2, 256, h, w), device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
             for h, w in [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]]
    fpn = [torch.randn((2, 256, h, w), device="cuda", generator=g).contiguous(memory_format=torch.channels_last).requires_grad_(True)
           for h, w in [(104, 168), (52, 84), (26, 42), (13, 21), (7, 11)]]

    def one():
        adv.d_step(guide, fpn)
        losses = adv.g_losses(guide, fpn)
        sum(v for k, v in losses.items() if k.startswith("g_loss")).backward()
        for f in fpn:
            f.grad = None

    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    px = sum(min(gh // 2, fh) * min(gw // 2, fw) for (gh, gw), (fh, fw) in zip([(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)],
                                                                                [(104, 168), (52, 84), (26, 42), (13, 21), (7, 11)]))
    flop = 2 * px * (2 * D_FWDBWD_DETACHED_FLOP_PER_PX + 2 * D_FWD_FLOP_PER_PX)      # D step: 2 fwd+bwd; G side: 2 fwd (no D gradient: Q1)
    return {"workload": "stage-2 adversarial terms (D step + generator-side losses), batch 2, FPN features of 416x672 inputs",
            "ms": dt * 1e3, "images_per_s": 2.0 / dt, "algorithmic_tflop": flop / 1e12, "tflops": flop / dt / 1e12}


def host_cores():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU box exposes 256 logical
    CPUs but grants 16; running 256 OpenMP threads against a 16-CPU quota throttles to a crawl)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(torch, batch):
    """Oracle (CPU restatement, kind "port") timed on the host cores on the SAME workload as the GPU step: the stage-1 D phase + G phase
    over the full P2..P6 pyramid of one per-GPU batch (about 40 s on 16 cores) -- measured, not extrapolated.  The frozen guide network's
    two forwards are not part of the oracle (bench harness on the GPU side; < 7 % of the GPU step), which makes this baseline slightly
    optimistic for the CPU."""
    from oracle import afigan_oracle as orc
    ncores = host_cores()
    torch.set_num_threads(ncores)
    gen = torch.Generator().manual_seed(0)
    gp = orc.reference_init_generator_params(generator=gen)
    dp = orc.reference_init_discriminator_params(generator=gen)
    hr_shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    lr_shapes = [(104, 168), (52, 84), (26, 42), (13, 21), (7, 11)]
    lr_f = [torch.randn((batch, 256, h, w), generator=gen) for h, w in lr_shapes]
    hr_f = [torch.randn((batch, 256, h, w), generator=gen) for h, w in hr_shapes]
    t0 = time.perf_counter()
    orc.stage1_d_phase(gp, dp, lr_f, hr_f, first_level=2)
    log(f"  oracle D phase done ({time.perf_counter() - t0:.1f} s)")
    orc.stage1_g_phase(gp, dp, lr_f, hr_f, first_level=2)
    dt = time.perf_counter() - t0
    log(f"  oracle G phase done ({dt:.1f} s)")
    # G fwd+bwd on the config-1 tensor as well (metric 1)
    x = torch.randn((1, 256, 25, 34), generator=gen).requires_grad_(True)
    gq = {k: v.clone().requires_grad_(True) for k, v in gp.items()}
    orc.generator_forward(x, gq).sum().backward()           # warm-up
    t1 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        orc.generator_forward(x, gq).sum().backward()
    tg = (time.perf_counter() - t1) / reps
    return {"value": batch / dt, "unit": "images/s", "cores": ncores, "kind": "port", "extrapolated": False,
            "sample": f"oracle D phase + G phase of ONE stage-1 iteration on the full P2..P6 pyramid of batch {batch} (the GPU step's workload; the two "
                      f"guide-network forwards excluded): {dt:.2f} s",
            "af_interpolator_out_mpix_per_s": 3400 / tg / 1e6, "af_interpolator_ms": tg * 1e3}


def file_sha256(path):
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()
    except OSError:
        return None


_T0 = time.perf_counter()


def log(msg):
    """progress on stderr (stdout carries only the one JSON line)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


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
    if args.debug_nt_ablation:
        _lib.load().afi_debug_set_nt_ablation(args.debug_nt_ablation)
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
    if dist is not None and os.environ.get("AFI_BENCH_OVERLAP_AB", "1") != "0":
        # the other setting of overlap_comm on the same engine, same inputs, same K (never the headline): the first RCCL run measures both
        other = not step.overlap_comm
        ab = {("overlapped" if step.overlap_comm else "blocking") + "_ms_per_step": round(elapsed / args.steps * 1e3, 3)}
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
        comm["overlap_ab"] = ab
        log(f"overlap_comm A/B: {ab}")
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
                                      note="two extra steps with the engine's overlap_d / overlap_g and the guide prefetch off (one stream), same HIP-event brackets; profiles/r05 holds both traces")
                                 if kernel_alone else None),
                # fp32-equivalent rate of the dominant kernel against the fp32 MFMA roof it replaces (> 1 is the point of the emulated forms)
                "achieved_over_fp32_mfma_peak": dom["tflops"] / PEAK_FP32_MFMA_TFLOPS,
                "peak_note": "fp32-equivalent TFLOP/s: 2*M*N*K once per product; peak = dense bf16 / f16 MFMA 2500 (quoted at 2.4 GHz) / MFMAs per product (6, 3 or 1), or the fp32 MFMA 157.3.  `achieved` is taken in the two-stream step, where a launch shares the chip with the other stream's kernels and its HIP-event duration stretches; `kernel_alone` is the same kernel with the chip to itself.  The library's own plain-fp16 GEMM (hipBLASLt) reaches 922 TFLOP/s on the largest shape of this step with random operands = 307 fp32-equivalent at three products (profiles/r05/hipblaslt_f16_ceiling.txt).  Dense f16 / bf16 MFMA work is power-limited on this chip: traffic.held_clock_ghz / traffic.mfma_busy_at_held_clock are the clock it holds under this kernel and the matrix-pipe duty there (SQ_BUSY_CYCLES, SQ_VALU_MFMA_BUSY_CYCLES of the one-stream counter pass): DESIGN.md 4b",
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
    roofline["winograd_multiply_reduction"] = "4x F(4x4,3x3): data / weight gradients, forwards without a backward, the discriminator's blocks 1 and 2; 2.25x F(2x2,3x3): the other forwards with a backward behind them"
    line = {
        "metric": "stage1_G+D_step_images_per_s", "value": n_img / elapsed, "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": {"fp32": "f32", "f16x3": "f32 emulated on the f16 matrix cores (operands scaled by a power of two per Winograd plane and split into two fp16 pieces, three f16 MFMAs per k-step, fp32 accumulate; tensors fp32)", "bf16x6": "f32 emulated on the bf16 matrix cores (operands split exactly into three bf16, six bf16 MFMAs per k-step, fp32 accumulate; tensors fp32)", "bf16x3": "bf16x3 (split-bf16 operands, three bf16 MFMAs per k-step, fp32 accumulate; tensors fp32)",
                                       "bf16": "bf16 (bf16 operands, fp32 accumulate; tensors fp32)"}[run_dtype], "data": "synthetic",
        "backend": (args.backend if dist is not None else None), "comm": comm,
        "config": {"workload": "configs[1]: stage-1 AFI-GAN G+D step, R-50-FPN guide random-init (eval), "
                               f"{B}x3x800x1333 synthetic images per GPU, P2..P6, G n_rdb=3",
                   "global_batch": world * B, "parallelism": f"dp{world}", "guide": "r50fpn (1x1 and 3x3 convs on this library's kernels, stem GEMM via hipBLASLt)" if guide is not None else "synthetic-pyramid",
                   "guide_prefetch": prefetch is not None, "options": args.option or None,
                   "host_enqueue_ms": {k: round(sum(v) / max(1, len(v)), 2) for k, v in host_timed.items() if v},
                   "reuse_generator_forward": True},
        "algorithmic_tflop_per_image": flop_img / 1e12,
        "step_tflops_per_gpu": flop_img * B * args.steps / elapsed / 1e12,
        # direct-convolution FLOP count of the step over the dense fp32 MFMA peak.  The big 3x3 convs run in Winograd form,
        # which EXECUTES 2.25x / 4x fewer products than this count, so the ratio can pass 1; `roofline` below is in executed products of
        # the dominant kernel (the batched Winograd GEMM) against that kernel's own roof.
        "step_algorithmic_tflops_over_fp32_mfma_peak": flop_img * B * args.steps / elapsed / 1e12 / PEAK_FP32_MFMA_TFLOPS,
        "conv_algorithm": "Winograd F(4x4,3x3) / F(3x3,4x4) (interpolation points {0, 1, -1, 1/2, -2, inf}) for data and weight gradients, forward-only passes and the discriminator's forwards of blocks 1 and 2 (winograd_f4_forward = 12: gradient deviation from fp64 equal to the exact-fp32 direct kernels'); F(2x2,3x3) for the other forwards a backward follows (the discriminator's block 0, the interpolator); for every 3x3 conv with >= 128 channels on both sides and >= 1024 pixels (planes fp32, or split into two fp16 pieces by the transforms under f16x3; their batched GEMMs in the arithmetic named by `dtype`); direct implicit GEMM on the fp32 MFMA elsewhere",
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
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
