"""The secondary legs of bench.py (the driver's contract lives in bench.py: one JSON line; this file holds what that line's extra objects
are measured with): roofs per kernel kind, the committed-traffic lookup, BASELINE metric 1 (the AF-interpolator forward + backward), the
SURVEY 8(f) rows (FPN / PAFPN merge, BiFPN inference and training, stage-2 adversarial terms) and the CPU baseline (the oracle, timed as
the checker's own port -- kind "port").  tools/interp_sweep.py, fpn_loop.py and bifpn_train_loop.py call the same functions."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
_T0 = time.perf_counter()

PEAK_FP32_MFMA_TFLOPS = 157.3           # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0          # MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA", dense
G_FWD_FLOP_PER_INPX = 19_206_144        # SURVEY.md 8(d) / BASELINE.md section 3
D_FWD_FLOP_PER_PX = 30_689_280
D_FWDBWD_DETACHED_FLOP_PER_PX = 89_708_544
KERNEL_TOKENS = ("gemm_nt_f16x3", "gemm_tn_f16x3", "gemm_nt", "gemm_tn", "pix_gemm_wk6", "pix_gemm_wk", "pix_gemm", "wgrad6", "wgrad")


def log(msg):
    """progress on stderr (stdout carries only the one JSON line)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def file_sha256(path):
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()
    except OSError:
        return None


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


def committed_traffic(fname, dom_kernel):
    """HBM traffic of a dominant kernel cannot be read live (PMC counters need their own rocprofv3 passes): report the committed
    measurement of the same command when there is one (the newest profiles/rNN/<fname>), else None.  The record names the kernel it was
    taken on -- another dominant kernel nulls it -- and is tied to the kernel SOURCE it was measured on by a sha256: a later edit of that
    file nulls it too."""
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
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
    # a kernel's own roof (kind_peak): the small-map kernels multiply six bf16 MFMAs per fp32-equivalent product, the f16x3 GEMMs three, the others
    # use the fp32 MFMA
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
