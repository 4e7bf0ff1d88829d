"""D forward+backward on one seeded input under the conv algorithm the tag names -- context options of the library (afi_ctx_set_option):
direct (winograd off), f2fwd (F(2x2) forwards, F(4x4) gradients), f4fwd (F(4x4) forwards too: the default since round 5), f4bK (F(4x4) forward in the
blocks K names only), f2all (F(2x2) everywhere) --
or torch's own fp32 / fp64 ops (torch32 / torch64); dumps the gradients so tools/dflip_cmp.py can compare the variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import afigan_amd as amd

tag, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
_cx = amd._lib.current_ctx()
for _k, _v in {"direct": {"winograd": 0}, "f2fwd": {"winograd_f4_forward": 0}, "f4fwd": {"winograd_f4_forward": 1}, "f2all": {"winograd_f4_forward": 0, "winograd_f4_backward": 0},
               "f4b0": {"winograd_f4_forward": 2}, "f4b1": {"winograd_f4_forward": 4}, "f4b2": {"winograd_f4_forward": 8},
               "f4b12": {"winograd_f4_forward": 12}, "f4b02": {"winograd_f4_forward": 10}, "f4b01": {"winograd_f4_forward": 6}}.get(tag, {}).items():
    _cx.set_option(_k, _v)
torch.manual_seed(0)
D = amd.Discriminator().cuda()
x = torch.randn(2, 256, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
r = torch.randn(2, 1, H, W, device="cuda")
names = [n for n, _ in D.named_parameters()]

if tag.startswith("torch"):
    dt = torch.float64 if tag == "torch64" else torch.float32
    torch.backends.cudnn.allow_tf32 = False
    sd = {k: v.detach().to(dt).requires_grad_(True) for k, v in D.named_parameters()}
    xx = x.detach().to(dt).requires_grad_(True)
    def one():
        h = xx
        for n in range(3):
            p = f"Discriminators.0.{n}.0."
            h = F.conv2d(h, sd[p + "weight"], sd[p + "bias"], padding=1)
            h = F.batch_norm(h, None, None, sd[p + "norm.weight"], sd[p + "norm.bias"], training=True, eps=1e-5)
            h = F.leaky_relu(h, 0.2)
        h = F.conv2d(h, sd["Discriminators.0.3.0.weight"], sd["Discriminators.0.3.0.bias"], padding=1)
        for v in sd.values(): v.grad = None
        xx.grad = None
        (h * r.to(dt)).sum().backward()
    grads = lambda: {"dx": xx.grad, **{n: sd[n].grad for n in names}}
else:
    def one():
        for p in D.parameters(): p.grad = None
        x.grad = None
        (D(x) * r).sum().backward()
    grads = lambda: {"dx": x.grad, **{n: p.grad for n, p in D.named_parameters()}}
one(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(3): one()
b.record(); torch.cuda.synchronize()
torch.save({k: v.double().cpu() for k, v in grads().items()}, f"gpurun_out/dflip_{tag}.pt")
print(tag, f"{a.elapsed_time(b) / 3:.3f} ms fwd+bwd", flush=True)
