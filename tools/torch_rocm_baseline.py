"""What the reference's own ops (torch.nn.functional.conv2d -> MIOpen, fp32) reach on this GPU for the dominant layers and for
the whole interpolator / discriminator: the "just run the reference on an MI355X" baseline next to this package's kernels.
Uses the CPU oracle's restatement moved to the GPU (test infrastructure as a baseline, never the product path).
MIOpen compiles its kernels on first use: progress is printed per item so a long first call is visible."""
import os, sys, time
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import afigan_oracle as orc

torch.backends.cudnn.benchmark = False
# the oracle builds its bilinear index tensors on the CPU; on the GPU use the reference's own op (generator_rdb.py:125)
orc.bilinear2x = lambda x: F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
dev = "cuda"


def timed(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


print("MIOPEN_FIND_MODE =", os.environ.get("MIOPEN_FIND_MODE"), flush=True)
for name, N, H, W, Ci, Co in [("D1@P2", 2, 200, 336, 512, 1024), ("D2@P3", 2, 100, 168, 1024, 1024), ("G0@lrP2", 2, 104, 168, 256, 256)]:
    for fmt in (torch.contiguous_format, torch.channels_last):
        x = torch.randn(N, Ci, H, W, device=dev).contiguous(memory_format=fmt).requires_grad_(True)
        w = (torch.randn(Co, Ci, 3, 3, device=dev) * 0.02).contiguous(memory_format=fmt).requires_grad_(True)
        t0 = time.perf_counter()
        y = F.conv2d(x, w, None, 1, 1); torch.cuda.synchronize()
        print(f"  {name} {fmt}: first forward {time.perf_counter() - t0:.1f} s", flush=True)
        dy = torch.randn_like(y)
        t0 = time.perf_counter()
        y.backward(dy); torch.cuda.synchronize()
        print(f"  {name} {fmt}: first backward {time.perf_counter() - t0:.1f} s", flush=True)
        fl = 2.0 * N * H * W * Ci * Co * 9
        tf = timed(lambda: F.conv2d(x, w, None, 1, 1))

        def fb():
            x.grad = None; w.grad = None
            F.conv2d(x, w, None, 1, 1).backward(dy)
        tb = timed(fb)
        print(f"{name} N{N} {H}x{W} {Ci}->{Co} {str(fmt).split('.')[-1]:16s}: fwd {tf * 1e3:8.3f} ms {fl / tf / 1e12:6.1f} TF | fwd+bwd {tb * 1e3:8.3f} ms {3 * fl / tb / 1e12:6.1f} TF", flush=True)

if "nets" in sys.argv:
    gp = {k: v.to(dev).requires_grad_(True) for k, v in orc.closed_form_generator_params().items()}
    for N in (1, 16):
        x = torch.randn(N, 256, 25, 34, device=dev, requires_grad=True)

        def gfb():
            for v in gp.values():
                v.grad = None
            x.grad = None
            orc.generator_forward(x, gp).sum().backward()
        t0 = time.perf_counter(); gfb(); torch.cuda.synchronize()
        print(f"  generator N={N}: first fwd+bwd {time.perf_counter() - t0:.1f} s", flush=True)
        t = timed(gfb, iters=10, warm=3)
        print(f"generator fwd+bwd {N}x256x25x34 (torch ops on the GPU): {t * 1e3:.3f} ms = {N * 3400 / t / 1e6:.3f} out-Mpix/s", flush=True)

if "step" in sys.argv:
    # one stage-1 iteration of the oracle (D phase + G phase incl. both SGD updates) on GPU tensors: config-2 pyramid, batch 2
    gp = {k: v.to(dev) for k, v in orc.closed_form_generator_params().items()}
    dp = {k: v.to(dev) for k, v in orc.closed_form_discriminator_params().items()}
    g = torch.Generator(device=dev).manual_seed(0)
    hr = [torch.randn((2, 256, h, w), device=dev, generator=g) for h, w in [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]]
    lr = [torch.randn((2, 256, h, w), device=dev, generator=g) for h, w in [(104, 168), (52, 84), (26, 42), (13, 21), (7, 11)]]
    dm, gm = {}, {}

    def step():
        global gp, dp
        d_losses, d_grads, d_bufs = orc.stage1_d_phase(gp, dp, lr, hr)
        dparams = {k: v for k, v in dp.items() if k in d_grads}
        orc.sgd_momentum_step(dparams, d_grads, dm, lr=1e-6)
        dp = dict(dp); dp.update(dparams); dp.update(d_bufs)
        g_losses, g_grads, d_bufs2 = orc.stage1_g_phase(gp, dp, lr, hr)
        dp.update(d_bufs2)
        gparams = dict(gp)
        orc.sgd_momentum_step(gparams, g_grads, gm, lr=1e-6)
        gp = gparams
    t0 = time.perf_counter(); step(); torch.cuda.synchronize()
    print(f"  oracle stage-1 step on the GPU: first iteration {time.perf_counter() - t0:.1f} s", flush=True)
    t = timed(step, iters=3, warm=1)
    print(f"stage-1 step, reference ops through torch/MIOpen on this GPU (batch 2, features given): {t * 1e3:.1f} ms = {2 / t:.2f} images/s", flush=True)
