"""Gradient deviation of the discriminator (forward + backward, train-mode BatchNorm) from an fp64 evaluation,
per forward-conv setting of the library and for torch's own fp32 ops, on the SAME seeded inputs.

    python tools/dflip_table.py H W [seed ...] > profiles/r06/dflip_<size>.txt

One process: the fp64 reference (torch ops on the GPU, dtype float64) is evaluated once per seed and every variant is
compared with it.  Columns: relative L2 of dx and of the worst parameter gradient (conv biases ahead of a train-mode
BatchNorm have an identically zero gradient and are skipped), per seed and as the mean over the seeds.  The decision
rule the table serves: the default `winograd_f4_forward` is the largest block set whose two
columns do not exceed torch fp32's on the same inputs (feature_patch_discriminator.py:32-41 is the op sequence)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import afigan_amd as amd

_args = [a for a in sys.argv[1:] if not a.startswith("--dtype=")]
_dtype = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--dtype=")), None)      # arithmetic of the big GEMMs (default: the library's)
H, W = int(_args[0]), int(_args[1])
seeds = [int(s) for s in _args[2:]] or [0]
VARIANTS = [("torch32", None),
            ("direct", {"winograd": 0}),
            ("f2fwd (=0), plain order", {"winograd_f4_forward": 0, "f16_local_sums": 0}),
            ("f4 block 2 (=8), plain", {"winograd_f4_forward": 8, "f16_local_sums": 0}),
            ("f4 block 1 (=4), plain", {"winograd_f4_forward": 4, "f16_local_sums": 0}),
            ("f4 blocks 1,2 (=12), plain", {"winograd_f4_forward": 12, "f16_local_sums": 0}),
            ("f4 all (=1), plain", {"winograd_f4_forward": 1, "f16_local_sums": 0}),
            ("=0, local sums blk 1,2", {"winograd_f4_forward": 0, "f16_local_sums": 12}),
            ("=8, local sums blk 1,2", {"winograd_f4_forward": 8, "f16_local_sums": 12}),
            ("=12, local sums blk 1,2", {"winograd_f4_forward": 12, "f16_local_sums": 12}),
            ("=12, local sums all", {"winograd_f4_forward": 12, "f16_local_sums": 1}),
            ("=1, local sums all", {"winograd_f4_forward": 1, "f16_local_sums": 1})]
cx = amd._lib.current_ctx()
if _dtype:
    cx.set_dtype(_dtype)
defaults = {k: cx.get_option(k) for k in ("winograd", "winograd_f4_forward", "f16_local_sums")}


def torch_grads(D, x, r, dt):
    sd = {k: v.detach().to(dt).requires_grad_(True) for k, v in D.named_parameters()}
    xx = x.detach().to(dt).requires_grad_(True)
    h = xx
    for n in range(3):
        p = f"Discriminators.0.{n}.0."
        h = F.conv2d(h, sd[p + "weight"], sd[p + "bias"], padding=1)
        h = F.batch_norm(h, None, None, sd[p + "norm.weight"], sd[p + "norm.bias"], training=True, eps=1e-5)
        h = F.leaky_relu(h, 0.2)
    h = F.conv2d(h, sd["Discriminators.0.3.0.weight"], sd["Discriminators.0.3.0.bias"], padding=1)
    (h * r.to(dt)).sum().backward()
    return {"dx": xx.grad.double(), **{n: sd[n].grad.double() for n in sd}}


def lib_grads(D, x, r):
    for p in D.parameters():
        p.grad = None
    x.grad = None
    (D(x) * r).sum().backward()
    return {"dx": x.grad.double(), **{n: p.grad.double() for n, p in D.named_parameters()}}


def deviations(o, ref):
    live = [k for k in ref if ref[k].norm() > 1e-9 * ref[k].numel() ** 0.5]
    worst = max(((o[k] - ref[k]).norm() / ref[k].norm()).item() for k in live)
    return ((o["dx"] - ref["dx"]).norm() / ref["dx"].norm()).item(), worst


torch.backends.cudnn.allow_tf32 = False
rows = {name: [] for name, _ in VARIANTS}
for seed in seeds:
    torch.manual_seed(seed)
    D = amd.Discriminator().cuda()
    x = torch.randn(2, 256, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r = torch.randn(2, 1, H, W, device="cuda")
    ref = torch_grads(D, x, r, torch.float64)
    for name, opts in VARIANTS:
        if opts is None:
            g = torch_grads(D, x, r, torch.float32)
        else:
            for k, v in defaults.items():
                cx.set_option(k, v)
            for k, v in opts.items():
                cx.set_option(k, v)
            g = lib_grads(D, x, r)
        rows[name].append(deviations(g, ref))
        del g
        torch.cuda.synchronize()
        print(f"# seed {seed} {name}: dx {rows[name][-1][0]:.3e} worst {rows[name][-1][1]:.3e}", file=sys.stderr, flush=True)
    del ref
for k, v in defaults.items():
    cx.set_option(k, v)

print(f"D fwd+bwd at 2x256x{H}x{W}, dtype {cx.dtype}, seeds {seeds}: relative L2 against fp64 (dx | worst parameter gradient)")
print(f"library defaults: winograd_f4_forward = {defaults['winograd_f4_forward']}, f16_local_sums = {defaults['f16_local_sums']} (every row sets both)")
hdr = "".join(f"  seed {s}: dx / worst " for s in seeds)
print(f"{'forward convs':28s}{hdr}  mean: dx / worst")
for name, _ in VARIANTS:
    v = rows[name]
    cells = "".join(f"  {a:.3e} / {b:.3e}" for a, b in v)
    print(f"{name:28s}{cells}  {sum(a for a, _ in v) / len(v):.3e} / {sum(b for _, b in v) / len(v):.3e}")
