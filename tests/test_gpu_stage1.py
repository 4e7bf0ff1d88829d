"""GPU parity of one stage-1 G+D iteration (afigan_amd/stage1.py) against the golden replay of
stage1_trainer.py:336-433 over the reference modules, and against the CPU oracle on a ragged small pyramid."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# D gradients on the library's own forward: mask-flip noise (~1e-3 L2 per flipped LeakyReLU element); tests/test_gpu_d_parity.py
D_GRAD_L2_TOL = 3e-3
D_GRAD_MAXNORM_TOL = 3e-2

from oracle import afigan_oracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def amd():
    import afigan_amd
    assert torch.cuda.is_available()
    return afigan_amd


def _digest(t, nsample=64):
    f = t.detach().contiguous().reshape(-1).double().cpu()
    stride = max(1, f.numel() // nsample)
    return np.array([f.sum().item(), f.norm().item(), f.abs().max().item()]), f[::stride][:nsample].float().numpy()


@pytest.mark.parametrize("pair", [0, 600, 10 ** 9])
def test_stage1_step_vs_reference_replay(amd, golden_dir, pair):
    """`pair` > 0: the two D calls per level and phase run as ONE call with per-batch BatchNorm statistics (Stage1Step(pair_d_max_pixels=...),
    afi_discriminator_fwd_paired): the same replay, the same tolerances."""
    fx = dict(np.load(os.path.join(golden_dir, "stage1_step.npz")))
    G = amd.Generator(n_residual_dense_blocks=3).cuda()
    D = amd.Discriminator().cuda()
    G.load_state_dict(orc.closed_form_generator_params(), strict=True)
    D.load_state_dict(orc.closed_form_discriminator_params(), strict=True)
    G.train(); D.train()
    gen = torch.Generator().manual_seed(int(fx["seed"][0]))
    lr_f = [torch.randn((2, 256, 13, 21), generator=gen), torch.randn((2, 256, 7, 11), generator=gen)]
    hr_f = [torch.randn((2, 256, 25, 42), generator=gen), torch.randn((2, 256, 13, 21), generator=gen)]
    step = amd.Stage1Step(G, D, base_lr=float(fx["lr"][0]), momentum=float(fx["mom"][0]), weight_decay=float(fx["wd"][0]),
                          warmup_iters=0, pair_d_max_pixels=pair)          # the golden replay uses a constant lr
    step.run_step([t.cuda() for t in lr_f], [t.cuda() for t in hr_f])
    m = step.metrics()
    for k in ("d_loss_p2", "d_loss_p3", "adv_loss_p2", "adv_loss_p3", "content_loss_p2", "content_loss_p3", "g_loss_p2", "g_loss_p3"):
        ref = float(fx[k][0])
        assert abs(m[k] - ref) <= 1e-3 * abs(ref), (k, m[k], ref)
    # gradients left in .grad are those of the D phase / G phase
    for k, p in D.named_parameters():
        if k.endswith(".0.bias") and not k.startswith("Discriminators.0.3"):
            continue                                  # rounding-noise gradients (bias before a train-mode BN)
        d, s = _digest(p.grad)                      # LeakyReLU mask flips: see test_gpu_modules.test_discriminator_vs_reference
        rd, rs = fx["Dgd/" + k], fx["Dgs/" + k]
        assert abs(d[1] - rd[1]) <= D_GRAD_L2_TOL * rd[1], (k, d, rd)
        np.testing.assert_allclose(s, rs, rtol=0, atol=D_GRAD_MAXNORM_TOL * rd[2], err_msg=k)
    for k, p in G.named_parameters():
        d, s = _digest(p.grad)
        rd, rs = fx["Ggd/" + k], fx["Ggs/" + k]
        assert abs(d[1] - rd[1]) <= 1e-3 * rd[1], (k, d, rd)
        np.testing.assert_allclose(s, rs, rtol=0, atol=1e-3 * rd[2], err_msg=k)
    # post-SGD weights
    for k, p in list(D.named_parameters()) + list(G.named_parameters()):
        ref = fx[("Dw_after/" if k.startswith("Disc") else "Gw_after/") + k]
        d, _ = _digest(p)
        assert abs(d[1] - ref[1]) <= 1e-5 * ref[1] + 1e-9, k
    # Q2: BN buffers advanced 4x per level
    sd = D.state_dict()
    for k in sd:
        if "num_batches" in k:
            assert int(sd[k]) == int(fx["Dbuf_after/" + k]) == 8
        elif "running" in k:
            ref = torch.from_numpy(fx["Dbuf_after/" + k])
            assert ((sd[k].cpu() - ref).abs().max() / ref.abs().max()).item() < 1e-3, k


@pytest.mark.parametrize("reuse", [True, False])
def test_stage1_small_ragged_vs_oracle(amd, reuse):
    """Small channel counts, odd sizes, three levels incl. a crop case (G(7x11)=14x22 vs hr 13x21, Q4)."""
    C, g = 16, 4
    gp = orc.closed_form_generator_params(C, 3, g)
    dp = orc.closed_form_discriminator_params(C)
    G = amd.Generator(in_channels=C, n_residual_dense_blocks=3, growth_rate=g).cuda()
    D = amd.Discriminator(in_filters=C).cuda()
    G.load_state_dict(gp); D.load_state_dict(dp)
    gen = torch.Generator().manual_seed(3)
    lr_f = [torch.randn((2, C, 13, 21), generator=gen), torch.randn((2, C, 7, 11), generator=gen), torch.randn((2, C, 4, 6), generator=gen)]
    hr_f = [torch.randn((2, C, 25, 42), generator=gen), torch.randn((2, C, 13, 21), generator=gen), torch.randn((2, C, 8, 12), generator=gen)]
    lr0 = 0.01
    step = amd.Stage1Step(G, D, base_lr=lr0, warmup_iters=0, reuse_generator_forward=reuse)
    seen = []
    step.on_d_level = seen.append                     # the caller's hook: once per level of the D phase, largest level first, while the step is enqueued
    step.run_step([t.cuda() for t in lr_f], [t.cuda() for t in hr_f])
    assert seen == [0, 1, 2]
    step.on_d_level = None
    m = step.metrics()
    d_losses, d_grads, d_bufs = orc.stage1_d_phase(gp, dp, lr_f, hr_f, first_level=2)
    for k, v in d_losses.items():
        assert abs(m[k] - v) <= 1e-3 * abs(v), (k, m[k], v)
    dparams = {k: v for k, v in dp.items() if k in d_grads}
    orc.sgd_momentum_step(dparams, d_grads, {}, lr=lr0)
    dp2 = dict(dp); dp2.update(dparams); dp2.update(d_bufs)
    g_losses, g_grads, d_bufs2 = orc.stage1_g_phase(gp, dp2, lr_f, hr_f, first_level=2)
    for k, v in g_losses.items():
        assert abs(m[k] - v) <= 1e-3 * abs(v), (k, m[k], v)
    for k, p in G.named_parameters():
        ref = g_grads[k]
        assert ((p.grad.cpu() - ref).abs().max() / ref.abs().max()).item() < 1e-3, k
    for k, p in D.named_parameters():
        ref = dparams[k]
        assert ((p.detach().cpu() - ref).abs().max() / ref.abs().max()).item() < 1e-4, k
    sd = D.state_dict()
    for k, v in d_bufs2.items():
        if "num_batches" in k:
            assert int(sd[k]) == int(v) == 12
        else:
            assert ((sd[k].cpu() - v).abs().max() / v.abs().max()).item() < 1e-3, k
    gparams = dict(gp)
    orc.sgd_momentum_step(gparams, g_grads, {}, lr=lr0)
    for k, p in G.named_parameters():
        assert ((p.detach().cpu() - gparams[k]).abs().max() / gparams[k].abs().max()).item() < 1e-4, k
    # second iteration keeps working on the same buffers (momentum path) and stays finite
    step.run_step([t.cuda() for t in lr_f], [t.cuda() for t in hr_f])
    assert all(np.isfinite(v) for v in step.metrics().values())
    # an option set on one of the engine's two contexts only is refused before anything is launched (a backward must run under its forward's options)
    step.ctx.set_option("winograd_f4_forward", 0)
    with pytest.raises(amd.AfiError):
        step.run_step([t.cuda() for t in lr_f], [t.cuda() for t in hr_f])
    step.set_option("winograd_f4_forward", 0)             # ... and on both it is a legal setting
    step.run_step([t.cuda() for t in lr_f], [t.cuda() for t in hr_f])
    assert all(np.isfinite(v) for v in step.metrics().values())


@pytest.mark.parametrize("pair", [0, 600, 10 ** 9])
def test_stage1_trajectory_four_iterations_vs_oracle(amd, pair):
    """Four consecutive iterations with the LR warm-up and a decay step inside the window, momentum and weight decay
    (stage1_trainer.py:110-125,336-433): per-iteration losses and the final weights / BN running statistics follow the
    oracle's trajectory, i.e. the optimizer state, the schedule and the buffer updates carry over correctly between steps."""
    C, g = 16, 4
    gp = orc.closed_form_generator_params(C, 3, g)
    dp = orc.closed_form_discriminator_params(C)
    G = amd.Generator(in_channels=C, n_residual_dense_blocks=3, growth_rate=g).cuda()
    D = amd.Discriminator(in_filters=C).cuda()
    G.load_state_dict(gp); D.load_state_dict(dp)
    gen = torch.Generator().manual_seed(11)
    sched = dict(lr_steps=(3,), lr_gamma=0.5, warmup_factor=0.1, warmup_iters=2)
    step = amd.Stage1Step(G, D, base_lr=0.02, pair_d_max_pixels=pair, **sched)
    g_bufs, d_bufs_m = {}, {}
    for it in range(4):
        lr_f = [torch.randn((2, C, 7, 11), generator=gen), torch.randn((2, C, 4, 6), generator=gen)]
        hr_f = [torch.randn((2, C, 13, 21), generator=gen), torch.randn((2, C, 8, 12), generator=gen)]
        step.run_step([t.cuda() for t in lr_f], [t.cuda() for t in hr_f])
        m = step.metrics()
        lr = orc.warmup_multistep_lr(0.02, it, steps=(3,), gamma=0.5, warmup_factor=0.1, warmup_iters=2)
        d_losses, d_grads, d_bufs = orc.stage1_d_phase(gp, dp, lr_f, hr_f, first_level=2)
        dparams = {k: v for k, v in dp.items() if k in d_grads}
        orc.sgd_momentum_step(dparams, d_grads, d_bufs_m, lr=lr)
        dp = dict(dp); dp.update(dparams); dp.update(d_bufs)
        g_losses, g_grads, d_bufs2 = orc.stage1_g_phase(gp, dp, lr_f, hr_f, first_level=2)
        dp.update(d_bufs2)
        gparams = dict(gp)
        orc.sgd_momentum_step(gparams, g_grads, g_bufs, lr=lr)
        gp = gparams
        for k, v in list(d_losses.items()) + list(g_losses.items()):
            assert abs(m[k] - v) <= 2e-3 * abs(v), (it, k, m[k], v)
    for k, p in G.named_parameters():
        assert ((p.detach().cpu() - gp[k]).abs().max() / gp[k].abs().max()).item() < 1e-3, k
    sd = D.state_dict()
    for k, v in dp.items():
        if "num_batches" in k:
            assert int(sd[k]) == int(v) == 4 * 4 * 2, k                      # 4 D forwards per level per iteration (Q2)
        else:
            assert ((sd[k].cpu() - v).abs().max() / (v.abs().max() + 1e-30)).item() < 2e-3, k


def _dp_gpu_worker(rank, world, port, tmp, backend="gloo", overlap_comm=True):
    """backend "gloo": two ranks share the one GPU of the test box (RCCL needs one GPU per rank); backend "nccl" (= RCCL): one GPU per
    rank.  Either way the real distributed code path of Stage1Step runs on the HIP kernels: broadcast, flat-buffer all-reduces (issued
    asynchronously beside G's backward passes / the G phase's D forwards with overlap_comm), fused SGD with 1/world, reduced metrics."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(rank if backend == "nccl" else 0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    import afigan_amd as amd
    C, g = 16, 4
    torch.manual_seed(50 + rank)                               # different init per rank; the broadcast must fix it
    G = amd.Generator(in_channels=C, n_residual_dense_blocks=2, growth_rate=g).cuda()
    D = amd.Discriminator(in_filters=C).cuda()
    gen = torch.Generator().manual_seed(1000 + rank)           # this rank's shard of the global batch
    lr_f = [torch.randn((1, C, 7, 11), generator=gen).cuda(), torch.randn((1, C, 4, 6), generator=gen).cuda()]
    hr_f = [torch.randn((1, C, 13, 21), generator=gen).cuda(), torch.randn((1, C, 8, 12), generator=gen).cuda()]
    step = amd.Stage1Step(G, D, base_lr=0.01, warmup_iters=0, overlap_comm=overlap_comm)  # picks up the initialised process group
    assert step.distributed and step.world == world and step.overlap_comm == overlap_comm
    w0 = {k: v.detach().clone() for k, v in list(G.state_dict().items()) + list(D.state_dict().items())}
    # single-rank gradients of this shard from the same starting weights (independent engine, no process group)
    import copy
    G1, D1 = copy.deepcopy(G), copy.deepcopy(D)
    solo = amd.Stage1Step(G1, D1, base_lr=0.01, warmup_iters=0, distributed=False)
    solo.run_step(lr_f, hr_f)
    snap = {}
    step.after_allreduce = lambda which, opt: snap.__setitem__(which, opt.flat_grad.detach().clone())   # right after the collective
    step.measure_comm = True                                   # what the consuming stream (gloo: the host) waits for each exchange
    step.run_step(lr_f, hr_f)
    torch.cuda.synchronize()
    exposure = step.comm_exposure()
    assert not step._pending_work                              # every issued collective was waited for
    out = {"comm_exposed_ms": exposure, "backend": step.backend,
           "w0": {k: v.cpu() for k, v in w0.items()},
           "w1": {k: v.detach().cpu() for k, v in list(G.state_dict().items()) + list(D.state_dict().items())},
           "g_sum": {k: p.grad.detach().cpu().clone() for k, p in G.named_parameters()},
           "g_solo": {k: p.grad.detach().cpu().clone() for k, p in G1.named_parameters()},
           # D: the all-reduced flat gradient buffer, snapshotted before the optimizer step, and this rank's single-process D-phase
           # gradients from the same broadcast weights (the D phase runs before any weight moves)
           "d_sum_flat": snap["D"].cpu(), "g_sum_flat": snap["G"].cpu(),
           "d_solo_flat": solo.d_opt.flat_grad.detach().cpu().clone(),
           "d_sizes": [({id(q): n for n, q in D.named_parameters()}[id(p)], p.numel()) for p in step.d_order],
           "metrics": step.metrics(), "metrics_mean": step.metrics(reduce=True)}
    torch.save(out, os.path.join(tmp, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _check_dp_results(r0, r1):
    for k in r0["w0"]:
        assert torch.equal(r0["w0"][k], r1["w0"][k]), k           # broadcast from rank 0
        if "running" in k or "num_batches" in k:
            continue                                              # BN buffers are per-rank (plain BN, no sync)
        assert torch.equal(r0["w1"][k], r1["w1"][k]), k           # identical update on every rank
    # D's weights differ between the solo and the distributed run, so only G's D-independent gradient (L1 only, Q1) can be
    # compared: summed all-reduced G grads == sum of the two single-rank G grads (same G weights, own shard each)
    for k in r0["g_sum"]:
        want = r0["g_solo"][k] + r1["g_solo"][k]
        got = r0["g_sum"][k]
        assert ((got - want).abs().max() / want.abs().max()).item() < 1e-4, k
        assert torch.equal(r0["g_sum"][k], r1["g_sum"][k]), k
    # D: every one of its 14 parameter tensors -- all-reduced gradient == sum of the per-shard single-process gradients (identical
    # weights and kernels on both sides, so the LeakyReLU masks agree bit for bit and 1e-4 holds)
    assert torch.equal(r0["d_sum_flat"], r1["d_sum_flat"]) and torch.equal(r0["g_sum_flat"], r1["g_sum_flat"])
    want, got = r0["d_solo_flat"] + r1["d_solo_flat"], r0["d_sum_flat"]
    assert len(r0["d_sizes"]) == 14
    o = 0
    for k, n in r0["d_sizes"]:
        w, g = want[o:o + n], got[o:o + n]
        o += (n + 3) // 4 * 4
        scale = w.abs().max().item()
        if k.endswith(".0.bias") and not k.startswith("Discriminators.0.3"):
            assert scale == 0.0 and g.abs().max().item() == 0.0, k      # bias in front of a train-mode BN: exactly zero
            continue
        assert scale > 0 and ((g - w).abs().max() / scale).item() < 1e-4, k
    # metrics(reduce=True): every rank holds the mean of the ranks' own values (stage1_trainer.py:453-492 averages on rank 0)
    for k, v in r0["metrics_mean"].items():
        assert r1["metrics_mean"][k] == v, k
        assert abs(v - 0.5 * (r0["metrics"][k] + r1["metrics"][k])) <= 1e-6 * max(1.0, abs(v)), k


def test_stage1_data_parallel_two_ranks_one_gpu(amd, tmp_path):
    """gloo, two ranks on the one test GPU, with the gradient exchange overlapped (default) and blocking: same invariants, and the two
    schedules leave the same parameters (the collectives are only moved in time; fp32 atomics reorder sums, nothing else may differ)."""
    import torch.multiprocessing as mp
    res = {}
    for overlap in (True, False):
        d = tmp_path / f"ov{int(overlap)}"
        d.mkdir()
        port = 29700 + (os.getpid() % 1000) + (7 if overlap else 0)
        mp.spawn(_dp_gpu_worker, args=(2, port, str(d), "gloo", overlap), nprocs=2, join=True)
        r0, r1 = torch.load(d / "r0.pt"), torch.load(d / "r1.pt")
        _check_dp_results(r0, r1)
        res[overlap] = r0
        ex = r0["comm_exposed_ms"]
        assert set(ex) == {"D", "G", "total", "host_blocked_ms"} and all(ex[k] >= 0.0 for k in ("D", "G", "total")), ex
        print(f"[two ranks, one GPU, {r0['backend']}] overlap_comm={overlap}: exchange exposed to the consuming stream D {ex['D']:.3f} ms, G {ex['G']:.3f} ms; "
              f"host inside the waits {ex['host_blocked_ms']}")
    for k, v in res[True]["w1"].items():
        if "num_batches" in k:
            assert torch.equal(v, res[False]["w1"][k]), k
        else:
            assert ((v - res[False]["w1"][k]).abs().max() / (v.abs().max() + 1e-30)).item() < 1e-5, k


def _rccl_single_rank_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import copy
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import afigan_amd as amd
    C, g = 16, 4
    torch.manual_seed(50)
    G = amd.Generator(in_channels=C, n_residual_dense_blocks=2, growth_rate=g).cuda()
    D = amd.Discriminator(in_filters=C).cuda()
    gen = torch.Generator().manual_seed(1000)
    lr_f = [torch.randn((2, C, 7, 11), generator=gen).cuda(), torch.randn((2, C, 4, 6), generator=gen).cuda()]
    hr_f = [torch.randn((2, C, 13, 21), generator=gen).cuda(), torch.randn((2, C, 8, 12), generator=gen).cuda()]
    out = {}
    for tag, kw in (("solo", dict(distributed=False)), ("rccl_overlapped", dict(distributed=True, overlap_comm=True)),
                    ("rccl_blocking", dict(distributed=True, overlap_comm=False))):
        G1, D1 = copy.deepcopy(G), copy.deepcopy(D)
        st = amd.Stage1Step(G1, D1, base_lr=0.01, warmup_iters=0, **kw)
        if kw["distributed"]:
            assert st.backend == "nccl" and st.world == 1 and st.overlap_comm == kw["overlap_comm"]
        st.measure_comm = bool(kw["distributed"])
        for _ in range(3):
            st.run_step(lr_f, hr_f)
        torch.cuda.synchronize()
        assert not st._pending_work
        out[tag] = {"w": {k: v.detach().cpu() for k, v in list(G1.state_dict().items()) + list(D1.state_dict().items())},
                    "metrics": st.metrics(reduce=bool(kw["distributed"])), "exposure": st.comm_exposure() if kw["distributed"] else None}
    torch.save(out, os.path.join(tmp, "r0.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_stage1_rccl_single_rank_group_runs_both_exchange_paths(amd, tmp_path):
    """One GPU is all this box has, and RCCL wants one GPU per rank -- but a ONE-rank `nccl` group is a real RCCL communicator: the
    engine's distributed path (broadcast at construction, `all_reduce(async_op=True)` issued from the caller's stream for D and from the
    engine's second stream for G, `work.wait()` ordering the consuming stream behind RCCL's, the 1/world factor in the fused SGD kernel,
    `metrics(reduce=True)`) runs on ProcessGroupNCCL with overlap_comm on and off.  Three steps each way land where the non-distributed
    engine lands (a sum over one rank is the identity), the host never blocks in a wait (what hid the gloo figure
    cannot hide here), and no collective is left pending.  (stage1_trainer.py:80-89.)"""
    import torch.multiprocessing as mp
    port = 29300 + (os.getpid() % 500)
    mp.spawn(_rccl_single_rank_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    r = torch.load(tmp_path / "r0.pt")
    for tag in ("rccl_overlapped", "rccl_blocking"):
        for k, v in r["solo"]["w"].items():
            if "num_batches" in k:
                assert torch.equal(v, r[tag]["w"][k]), (tag, k)
            else:
                assert ((v - r[tag]["w"][k]).abs().max() / (v.abs().max() + 1e-30)).item() < 1e-5, (tag, k)
        for k, v in r["solo"]["metrics"].items():
            assert abs(r[tag]["metrics"][k] - v) <= 1e-5 * max(1.0, abs(v)), (tag, k)
        print(f"[RCCL, one-rank group] {tag}: exchange exposed to the consuming stream {r[tag]['exposure']}")
    hb = r["rccl_overlapped"]["exposure"]["host_blocked_ms"]
    assert all(v < 5.0 for v in hb.values()), hb              # nccl: wait() orders streams, the host goes on


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: runs where the box has >= 2 devices")
def test_stage1_data_parallel_rccl(amd, tmp_path):
    """The same invariants over the `nccl` backend (RCCL over xGMI), one process per device -- picked up by any box that shows two or
    more GPUs (the builder's boxes show one; SURVEY 8e, stage1_trainer.py:80-89)."""
    import torch.multiprocessing as mp
    port = 29900 + (os.getpid() % 1000)
    mp.spawn(_dp_gpu_worker, args=(2, port, str(tmp_path), "nccl", True), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    _check_dp_results(r0, torch.load(tmp_path / "r1.pt"))
    print(f"[RCCL, two GPUs] overlap_comm=True: exchange exposed to the consumer {r0['comm_exposed_ms']}")


def test_stage1_engine_state_round_trip(amd):
    """Stage1Step.state_dict / load_state_dict (momentum buffers by parameter name in logical [O,I,kh,kw] shape, iteration;
    stage1_trainer.py:129-174): 2 steps -> save -> new engine on freshly constructed networks -> load -> 2 steps lands where 4
    uninterrupted steps land (the weight-gradient kernels add partial tiles by fp32 atomics, so "where" is 1e-5, not bit for bit; the
    buffers themselves travel bit for bit), and dropping the momentum on the way does not (the test can see the state)."""
    import copy
    C, g = 16, 4
    torch.manual_seed(21)
    G0 = amd.Generator(in_channels=C, n_residual_dense_blocks=2, growth_rate=g).cuda()
    D0 = amd.Discriminator(in_filters=C).cuda()
    gen = torch.Generator().manual_seed(22)
    batches = [([torch.randn((2, C, 7, 11), generator=gen).cuda(), torch.randn((2, C, 4, 6), generator=gen).cuda()],
                [torch.randn((2, C, 13, 21), generator=gen).cuda(), torch.randn((2, C, 8, 12), generator=gen).cuda()]) for _ in range(4)]
    sched = dict(base_lr=0.05, lr_steps=(3,), lr_gamma=0.5, warmup_factor=0.1, warmup_iters=2)

    def flat(G, D):
        return torch.cat([v.detach().double().reshape(-1).cpu() for v in list(G.state_dict().values()) + list(D.state_dict().values())])

    Ga, Da = copy.deepcopy(G0), copy.deepcopy(D0)
    ref = amd.Stage1Step(Ga, Da, **sched)
    ref.set_option("deterministic", 1)                   # (every engine of this test: it is about the state that travels, not about the atomics' summation order)
    for lr_f, hr_f in batches:
        ref.run_step(lr_f, hr_f)
    want = flat(Ga, Da)

    Gb, Db = copy.deepcopy(G0), copy.deepcopy(D0)
    first = amd.Stage1Step(Gb, Db, **sched)
    first.set_option("deterministic", 1)
    for lr_f, hr_f in batches[:2]:
        first.run_step(lr_f, hr_f)
    torch.cuda.synchronize()
    ckpt = {"G": {k: v.cpu() for k, v in Gb.state_dict().items()}, "D": {k: v.cpu() for k, v in Db.state_dict().items()},
            "engine": {k: v for k, v in first.state_dict().items()}}
    assert ckpt["engine"]["iteration"] == 2 and ckpt["engine"]["scheduler"]["last_epoch"] == 2
    for net, opt in (("G_optimizer", first.g_opt), ("D_optimizer", first.d_opt)):
        mb = ckpt["engine"][net]["momentum_buffer"]
        assert list(mb) == opt.names
        for n, p in zip(opt.names, opt.params):
            assert tuple(mb[n].shape) == tuple(p.shape) and mb[n].is_contiguous(), n       # logical shape, layout-independent
        assert any(float(v.abs().max()) > 0 for v in mb.values())

    def resume(with_momentum):
        Gc = amd.Generator(in_channels=C, n_residual_dense_blocks=2, growth_rate=g).cuda()
        Dc = amd.Discriminator(in_filters=C).cuda()
        Gc.load_state_dict(ckpt["G"]); Dc.load_state_dict(ckpt["D"])
        eng = amd.Stage1Step(Gc, Dc, **sched)
        eng.set_option("deterministic", 1)
        sd = copy.deepcopy(ckpt["engine"])
        if not with_momentum:
            for net in ("G_optimizer", "D_optimizer"):
                sd[net]["momentum_buffer"] = {k: torch.zeros_like(v) for k, v in sd[net]["momentum_buffer"].items()}
        eng.load_state_dict(sd)
        assert eng.iter == 2
        if with_momentum:
            assert torch.equal(eng.g_opt.flat_mom, first.g_opt.flat_mom) and torch.equal(eng.d_opt.flat_mom, first.d_opt.flat_mom)
        for lr_f, hr_f in batches[2:]:
            eng.run_step(lr_f, hr_f)
        return flat(Gc, Dc)

    got = resume(True)
    assert float((got - want).norm() / want.norm()) < 1e-5
    assert float((resume(False) - want).norm() / want.norm()) > 1e-4
    with pytest.raises(KeyError):
        bad = copy.deepcopy(ckpt["engine"]); bad["G_optimizer"]["momentum_buffer"].pop(first.g_opt.names[0])
        first.load_state_dict(bad)


def test_stage1_phase_caches_do_not_change_the_gradients(amd, monkeypatch):
    """The per-phase weight-transform cache and the transform-domain weight-gradient accumulator (afi_ctx_set_wino_weight_cache /
    afi_ctx_set_wino_wgrad_accum + afi_ctx_wino_wgrad_flush) are pure re-orderings: one step on a mid-size two-level pyramid (both
    Winograd tilings active: 2x64x96 -> F(4x4), 2x32x48 -> F(2x2)) yields the same flat gradient buffers with and without them,
    and two steps (weights moved in between: the caches must have been invalidated) the same losses."""
    def run(wcache, wgacc):
        torch.manual_seed(0)
        G = amd.Generator(n_residual_dense_blocks=3).cuda()
        D = amd.Discriminator().cuda()
        step = amd.Stage1Step(G, D, base_lr=0.05, warmup_iters=0, weight_cache=wcache == "1", wgrad_accum=wgacc == "1")
        gen = torch.Generator(device="cuda").manual_seed(3)
        hrs = [torch.randn((2, 256, 64, 96), device="cuda", generator=gen), torch.randn((2, 256, 32, 48), device="cuda", generator=gen)]
        lrs = [torch.randn((2, 256, 33, 49), device="cuda", generator=gen), torch.randn((2, 256, 16, 24), device="cuda", generator=gen)]
        step.run_step(lrs, hrs)
        grads = (step.d_opt.flat_grad.detach().clone(), step.g_opt.flat_grad.detach().clone())
        step.run_step(lrs, hrs)
        return grads, step.metrics()
    (d0, g0), m0 = run("0", "0")
    for mode in (("1", "0"), ("1", "1")):
        (d1, g1), m1 = run(*mode)
        assert float((d1 - d0).norm() / d0.norm()) < 1e-5, mode       # fp32 atomics reorder sums; nothing else may differ
        assert float((g1 - g0).norm() / g0.norm()) < 1e-5, mode
        for k, v in m0.items():
            assert abs(m1[k] - v) <= 2e-4 * abs(v) + 1e-6, (mode, k, m1[k], v)


@pytest.mark.parametrize("overlap_d,overlap_g", [(False, False), (True, False), (True, True)])
def test_stage1_one_stream_and_two_stream_steps_agree(amd, overlap_d, overlap_g):
    """The D phase on two streams (forwards on the caller's, backwards on the engine's second one, each with its own afi_ctx_t) and G's
    backward beside the G-phase D forwards are pure schedules: two steps from the same state give the same flat gradients, parameters and
    losses as the one-stream engine (fp32 atomics reorder sums; nothing else may differ).  ADVICE r2: Stage1Step(overlap_d=, overlap_g=)."""
    import copy
    torch.manual_seed(0)
    G0 = amd.Generator(n_residual_dense_blocks=3).cuda()
    D0 = amd.Discriminator().cuda()
    gen = torch.Generator(device="cuda").manual_seed(4)
    hrs = [torch.randn((2, 256, 64, 96), device="cuda", generator=gen), torch.randn((2, 256, 32, 48), device="cuda", generator=gen)]
    lrs = [torch.randn((2, 256, 33, 49), device="cuda", generator=gen), torch.randn((2, 256, 16, 24), device="cuda", generator=gen)]

    def run(od, og):
        G, D = copy.deepcopy(G0), copy.deepcopy(D0)
        step = amd.Stage1Step(G, D, base_lr=0.05, warmup_iters=0, overlap_d=od, overlap_g=og)
        assert step.ctx.handle.value != step.bctx.handle.value       # one context per stream (include/afigan_hip.h)
        step.run_step(lrs, hrs)
        grads = (step.d_opt.flat_grad.detach().clone(), step.g_opt.flat_grad.detach().clone())
        params = torch.cat([p.detach().reshape(-1) for p in list(G.parameters()) + list(D.parameters())])     # after ONE step
        step.run_step(lrs, hrs)          # (a second step's gradients sit behind LeakyReLU masks of moved weights: compared through the losses only)
        torch.cuda.synchronize()
        return grads, step.metrics(), params
    (d0, g0), m0, p0 = run(False, False)
    (d1, g1), m1, p1 = run(overlap_d, overlap_g)
    assert float((d1 - d0).norm() / d0.norm()) < 1e-5 and float((g1 - g0).norm() / g0.norm()) < 1e-5
    assert float((p1 - p0).norm() / p0.norm()) < 1e-6
    for k, v in m0.items():
        assert abs(m1[k] - v) <= 2e-4 * abs(v) + 1e-6, (k, m1[k], v)


def test_stage1_error_in_a_phase_joins_the_second_stream(amd):
    """An exception inside the phases must not leave backward kernels of the second stream reading memory the caller's stream gets back
    (ADVICE r2): the handler joins the stream, drops the partial weight-gradient sums, and the next step works."""
    torch.manual_seed(0)
    G = amd.Generator(n_residual_dense_blocks=3).cuda()
    D = amd.Discriminator().cuda()
    step = amd.Stage1Step(G, D, base_lr=0.01, warmup_iters=0)
    gen = torch.Generator(device="cuda").manual_seed(5)
    hrs = [torch.randn((1, 256, 32, 48), device="cuda", generator=gen)]
    lrs = [torch.randn((1, 256, 16, 24), device="cuda", generator=gen)]
    joined = []
    orig_join, orig_all = step._join_bstream, step._allreduce_start
    step._join_bstream = lambda: (joined.append(1), orig_join())[1]

    def boom(opt):
        raise RuntimeError("injected")
    step._allreduce_start = boom
    with pytest.raises(RuntimeError, match="injected"):
        step.run_step(lrs, hrs)
    assert len(joined) >= 2                                            # the phase's own join and the handler's
    step._allreduce_start = orig_all
    step.run_step(lrs, hrs)                                            # contexts are clean again (no pending sums, caches unregistered)
    assert all(np.isfinite(v) for v in step.metrics().values())


def test_guide_prefetcher_hands_over_the_same_features_one_step_early(amd):
    """afigan_amd.GuidePrefetcher (the frozen guide's two forwards of stage1_trainer.py:316-327 issued for batch i + 1 while batch i trains):
    the features it hands over equal a direct call's bit for bit, over several overlapped iterations; misuse raises."""
    from afigan_amd.guide import GuideR50FPN
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    guide = GuideR50FPN().to(dev)
    gen = torch.Generator(device=dev).manual_seed(7)
    images = [torch.rand((1, 3, 128, 160), device=dev, generator=gen) * 255.0 for _ in range(3)]

    def pair(img):
        return lambda: [guide(img)[f"p{d}"] for d in range(2, 7)]
    with torch.no_grad():
        direct = [[t.clone() for t in pair(img)()] for img in images]
    pf = amd.GuidePrefetcher(dev)
    with pytest.raises(RuntimeError):
        pf.take()
    pf.submit(pair(images[0]))
    with pytest.raises(RuntimeError):
        pf.submit(pair(images[1]))
    burn = torch.randn((2048, 2048), device=dev)
    for i in range(3):
        feats = pf.take()
        if i + 1 < 3:
            pf.submit(pair(images[i + 1]))                 # runs beside the "training step" below
        for _ in range(4):                                 # something for the caller's stream to do meanwhile
            burn = torch.tanh(burn @ burn * 1e-3)
        assert not pf.pending or i + 1 < 3
        for a, b in zip(feats, direct[i]):
            assert torch.equal(a, b), i
    torch.cuda.synchronize()


def test_guide_prefetcher_keeps_its_inputs_alive(amd):
    """the images of batch i + 1 are allocated on the caller's stream, read by the prefetch stream, and dropped by the caller
    right after ``submit``.  The prefetcher must keep them from the caching allocator until the guide forward has run: the block is not
    handed to the caller's next same-size allocation, and the features equal a direct call's."""
    from afigan_amd.guide import GuideR50FPN
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    guide = GuideR50FPN().to(dev)
    gen = torch.Generator(device=dev).manual_seed(9)
    img = torch.rand((1, 3, 128, 160), device=dev, generator=gen) * 255.0
    with torch.no_grad():
        direct = [guide(img)[f"p{d}"].clone() for d in range(2, 7)]
    torch.cuda.synchronize()
    burn = torch.randn((4096, 4096), device=dev)
    pf = amd.GuidePrefetcher(dev)
    def make(batch):
        def fn(b=None):
            x = burn
            for _ in range(6):                              # keeps the prefetch stream busy before it reads the image
                x = torch.tanh(x @ x * 1e-3)
            return [guide((batch if b is None else b)["image"])[f"p{d}"] for d in range(2, 7)]
        return fn

    import functools
    for how in ("closure", "partial", "explicit"):
        batch = {"image": img.clone()}
        old_ptr = batch["image"].data_ptr()
        if how == "closure":
            pf.submit(make(batch))                          # the callable's closure holds the batch dict
        elif how == "partial":
            pf.submit(functools.partial(make({}), batch))   # reached through functools.partial arguments
        else:
            holder = [batch]
            pf.submit(lambda: make(holder[0])(), batch["image"])     # (closure -> list -> dict is walked too; passed explicitly as well)
            holder.clear()
        assert any(t.data_ptr() == old_ptr for t in pf._pending[1]), how
        batch.clear()                                       # the caller drops its references right after submit
        fresh = [torch.full((1, 3, 128, 160), 1e9, device=dev) for _ in range(4)]     # the caller's next allocations of that size
        assert all(t.data_ptr() != old_ptr for t in fresh), how
        feats = pf.take()
        for a, b in zip(feats, direct):
            assert torch.equal(a, b), how
        del fresh, feats
    torch.cuda.synchronize()


def test_guide_forward_queued_from_the_step_hook_is_isolated_from_the_engine(amd):
    """``Stage1Step.on_d_level`` is called while the engine's forward context (phase weight cache registered) is the active one,
    and bench.py uses the hook to ``GuidePrefetcher.submit`` the next batch's guide forwards onto another stream.  That work must never
    land on the engine's contexts: (1) every context-taking library call made from the hook goes to the prefetcher's own context (observed
    call by call); (2) the features of a FULL-SIZE guide forward (2 x 3 x 800 x 1333, image and image_x0.5) queued from the hook beside a
    full-size step equal a stand-alone forward's bit for bit; (3) the engine's own results do not depend on the hook's work."""
    from afigan_amd import _lib
    from afigan_amd.guide import GuideR50FPN
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    guide = GuideR50FPN().to(dev)
    G = amd.Generator(in_channels=256, n_residual_dense_blocks=3).to(dev)
    D = amd.Discriminator().to(dev)
    gen = torch.Generator(device=dev).manual_seed(11)
    img = torch.rand((2, 3, 800, 1333), device=dev, generator=gen) * 255.0
    half = torch.nn.functional.interpolate(img, size=(400, 666), mode="bilinear", align_corners=False)

    def guide_pair():
        hr, lr = guide(img), guide(half)
        return [hr[f"p{d}"] for d in range(2, 7)], [lr[f"p{d}"] for d in range(2, 7)]
    with torch.no_grad():
        hr0, lr0 = guide_pair()
        hr0, lr0 = [t.clone() for t in hr0], [t.clone() for t in lr0]
    torch.cuda.synchronize()

    step = amd.Stage1Step(G, D, base_lr=1e-3)
    pf = amd.GuidePrefetcher(dev)
    assert pf.ctx is not step.ctx and pf.ctx is not step.bctx and pf.ctx is not _lib.current_ctx()
    in_hook, foreign = [False], []

    def observer(name, cx):
        if in_hook[0] and cx is not pf.ctx:
            foreign.append((name, "engine fwd" if cx is step.ctx else "engine bwd" if cx is step.bctx else "other"))

    def hook(i):
        if i == 3:                                          # beside the small levels, where bench.py places it
            in_hook[0] = True
            try:
                pf.submit(guide_pair)
            finally:
                in_hook[0] = False
    _lib._observers.append(observer)
    try:
        step.on_d_level = hook
        step.run_step(lr0, hr0)
        hr1, lr1 = pf.take()
        torch.cuda.synchronize()
    finally:
        _lib._observers.remove(observer)
        step.on_d_level = None
    assert foreign == [], foreign[:4]
    for a, b in zip(hr1 + lr1, hr0 + lr0):
        assert torch.equal(a, b)
    with_hook = step.metrics()
    # the same step without the hook, from the same weights: the engine's numbers do not depend on what the hook queued
    torch.manual_seed(5)
    _ = GuideR50FPN()                                       # (consumes the same random numbers as above)
    G2 = amd.Generator(in_channels=256, n_residual_dense_blocks=3).to(dev)
    D2 = amd.Discriminator().to(dev)
    step2 = amd.Stage1Step(G2, D2, base_lr=1e-3)
    step2.run_step(lr0, hr0)
    without = step2.metrics()
    for k, v in without.items():
        assert abs(with_hook[k] - v) <= 1e-5 * abs(v) + 1e-7, (k, with_hook[k], v)


def test_stage1_reference_checkpoint_layout_round_trips_through_torch_sgd(amd):
    """Stage1Step.to_reference_checkpoints / load_reference_checkpoints: the layout the reference's two DetectionCheckpointers write
    (stage1_trainer.py:129-174: "optimizer" = torch.optim.SGD.state_dict() keyed by parameter index, "scheduler", "iteration" = the iteration
    that just FINISHED; resume at + 1, :167-172).  (1) a real torch.optim.SGD over the modules' parameters accepts the optimizer entry and
    holds the engine's momentum afterwards; (2) what that SGD saves loads back and two more steps land where four uninterrupted ones do;
    (3) the iteration convention; (4) a stored schedule that differs from the engine's is refused by both loaders."""
    import copy
    C, g = 16, 4
    torch.manual_seed(31)
    G0 = amd.Generator(in_channels=C, n_residual_dense_blocks=2, growth_rate=g).cuda()
    D0 = amd.Discriminator(in_filters=C).cuda()
    gen = torch.Generator().manual_seed(32)
    batches = [([torch.randn((2, C, 7, 11), generator=gen).cuda(), torch.randn((2, C, 4, 6), generator=gen).cuda()],
                [torch.randn((2, C, 13, 21), generator=gen).cuda(), torch.randn((2, C, 8, 12), generator=gen).cuda()]) for _ in range(4)]
    sched = dict(base_lr=0.05, lr_steps=(3,), lr_gamma=0.5, warmup_factor=0.1, warmup_iters=2)

    def flat(G, D):
        return torch.cat([v.detach().double().reshape(-1).cpu() for v in list(G.state_dict().values()) + list(D.state_dict().values())])

    Ga, Da = copy.deepcopy(G0), copy.deepcopy(D0)
    # (AFI_OPT_DETERMINISTIC on the three engines: this test is about the checkpoint layout; with the default atomics-summed weight gradients a
    #  single LeakyReLU mask flipped by the summation order moves "where four steps land" by 2.5e-5 -- seen once in round 6)
    ref = amd.Stage1Step(Ga, Da, **sched)
    ref.set_option("deterministic", 1)
    for lr_f, hr_f in batches:
        ref.run_step(lr_f, hr_f)
    want = flat(Ga, Da)

    Gb, Db = copy.deepcopy(G0), copy.deepcopy(D0)
    first = amd.Stage1Step(Gb, Db, **sched)
    first.set_option("deterministic", 1)
    for lr_f, hr_f in batches[:2]:
        first.run_step(lr_f, hr_f)
    torch.cuda.synchronize()
    ck = first.to_reference_checkpoints()
    assert ck["G"]["iteration"] == 1 and ck["D"]["iteration"] == 1            # two steps done: iteration index 1 just finished
    assert ck["G"]["scheduler"]["last_epoch"] == 2 and ck["G"]["scheduler"]["milestones"] == [3]
    # (1) torch's own optimizer takes it: one group per parameter (detectron2 v0.1.1 build_optimizer), norm parameters without weight decay
    saved = {}
    for tag, net, opt in (("G", Gb, first.g_opt), ("D", Db, first.d_opt)):
        params = list(net.parameters())
        names = [n for n, _ in net.named_parameters()]
        tsgd = torch.optim.SGD([{"params": [q], "weight_decay": 0.0 if ".norm." in n else 1e-4} for n, q in zip(names, params)], lr=0.05, momentum=0.9)
        tsgd.load_state_dict(ck[tag]["optimizer"])
        for n, q in zip(names, params):
            mb = tsgd.state[q]["momentum_buffer"]
            assert tuple(mb.shape) == tuple(q.shape), n
            assert torch.equal(mb.cpu(), opt.state_dict()[n].cpu()), n
        assert [gr["weight_decay"] for gr in tsgd.param_groups] == [0.0 if ".norm." in n else 1e-4 for n in names]
        assert abs(tsgd.param_groups[0]["lr"] - first.lr_at(2)) < 1e-12
        saved[tag] = {"optimizer": copy.deepcopy(tsgd.state_dict()), "scheduler": ck[tag]["scheduler"], "iteration": ck[tag]["iteration"],
                      "model": {k: v.cpu() for k, v in net.state_dict().items()}}
    # (2) resume from what torch saved
    Gc = amd.Generator(in_channels=C, n_residual_dense_blocks=2, growth_rate=g).cuda()
    Dc = amd.Discriminator(in_filters=C).cuda()
    Gc.load_state_dict(saved["G"]["model"]); Dc.load_state_dict(saved["D"]["model"])
    eng = amd.Stage1Step(Gc, Dc, **sched)
    eng.set_option("deterministic", 1)
    eng.load_reference_checkpoints(saved)
    assert eng.iter == 2                                                       # (3) resumes at the finished iteration + 1
    for lr_f, hr_f in batches[2:]:
        eng.run_step(lr_f, hr_f)
    got = flat(Gc, Dc)
    assert ((got - want).norm() / want.norm()).item() < 1e-5
    # (4) a schedule that is not this engine's is refused
    other = amd.Stage1Step(copy.deepcopy(G0), copy.deepcopy(D0), **dict(sched, lr_steps=(5,)))
    with pytest.raises(ValueError):
        other.load_reference_checkpoints(saved)
    with pytest.raises(ValueError):
        other.load_state_dict(first.state_dict())
    bad = copy.deepcopy(saved); bad["D"]["iteration"] = 7
    with pytest.raises(ValueError):
        eng.load_reference_checkpoints(bad)


def test_stage1_deterministic_option_resumes_bit_for_bit(amd):
    """AFI_OPT_DETERMINISTIC on both of the engine's contexts: no weight gradient is split over blocks that meet in atomics, so (1) two runs
    from the same state produce the same bits, and (2) a run that is saved after 2 steps and resumed in fresh modules continues bit for bit
    where 4 uninterrupted steps land (without the option: 1e-5, test_stage1_engine_state_round_trip).  Maps large enough for the Winograd
    weight gradients (the split-K TN GEMM) on the first level, small-map kernels on the second."""
    import copy
    C = 128
    torch.manual_seed(41)
    G0 = amd.Generator(in_channels=C, n_residual_dense_blocks=1, growth_rate=32).cuda()
    D0 = amd.Discriminator(in_filters=C).cuda()
    gen = torch.Generator().manual_seed(42)
    batches = [([torch.randn((1, C, 24, 44), generator=gen).cuda(), torch.randn((1, C, 7, 11), generator=gen).cuda()],
                [torch.randn((1, C, 47, 87), generator=gen).cuda(), torch.randn((1, C, 13, 21), generator=gen).cuda()]) for _ in range(4)]
    sched = dict(base_lr=0.01, lr_steps=(3,), lr_gamma=0.5, warmup_factor=0.1, warmup_iters=2)

    def engine(G, D):
        e = amd.Stage1Step(G, D, **sched)
        for cx in (e.ctx, e.bctx):
            cx.set_option("deterministic", 1)
        return e

    def flat(G, D):
        return torch.cat([v.detach().reshape(-1).float().cpu() for v in list(G.state_dict().values()) + list(D.state_dict().values())])

    runs = []
    for _ in range(2):
        Ga, Da = copy.deepcopy(G0), copy.deepcopy(D0)
        e = engine(Ga, Da)
        for lr_f, hr_f in batches:
            e.run_step(lr_f, hr_f)
        torch.cuda.synchronize()
        runs.append(flat(Ga, Da))
    assert torch.equal(runs[0], runs[1])                       # (1)
    Gb, Db = copy.deepcopy(G0), copy.deepcopy(D0)
    first = engine(Gb, Db)
    for lr_f, hr_f in batches[:2]:
        first.run_step(lr_f, hr_f)
    torch.cuda.synchronize()
    ck = {"G": {k: v.cpu() for k, v in Gb.state_dict().items()}, "D": {k: v.cpu() for k, v in Db.state_dict().items()}, "engine": first.state_dict()}
    Gc = amd.Generator(in_channels=C, n_residual_dense_blocks=1, growth_rate=32).cuda()
    Dc = amd.Discriminator(in_filters=C).cuda()
    Gc.load_state_dict(ck["G"]); Dc.load_state_dict(ck["D"])
    second = engine(Gc, Dc)
    second.load_state_dict(ck["engine"])
    for lr_f, hr_f in batches[2:]:
        second.run_step(lr_f, hr_f)
    torch.cuda.synchronize()
    assert torch.equal(flat(Gc, Dc), runs[0])                  # (2)
    assert float((runs[0] - flat(G0, D0)).abs().max()) > 0     # (the steps moved the weights)
