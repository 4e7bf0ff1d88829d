"""How bench.py becomes an N-rank job, and what it checks across the ranks: `python bench.py --gpus N` spawning its own ranks
(reference: stage1_train.py:52-59, detectron2 launch), the `comm` object's measurements (each gradient exchange alone, per-rank rates, the
cross-rank identity check) -- used by the real run on device buffers and by --rehearse-launch on CPU tensors over gloo (what
tests/test_host_logic.py runs with 2 and with 8 ranks in a container without a GPU)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")

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
        procs.append(subprocess.Popen([sys.executable, BENCH] + sys.argv[1:], env=env,
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
        overlap = bool(args.overlap_comm) if args.overlap_comm is not None else False       # (the engine's default: blocking)
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
