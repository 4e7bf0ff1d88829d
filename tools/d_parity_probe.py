"""Where does the discriminator's gradient deviation come from?  

For each D shape: forward through the C-ABI, saved conv outputs / batch statistics / activations against an fp64 evaluation of the
oracle -- next to the same comparison for the oracle run in fp32 on the CPU (what the reference's own ops give); then the backward
twice, on the library's own forward and on the fp64 forward's activations injected into the workspace (identical LeakyReLU masks by
construction), each against fp64 autograd.   Usage: python tools/d_parity_probe.py [N H W seed] ..."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import afigan_amd as amd
from oracle import afigan_oracle as orc
from d_parity_util import DProbe, rel


def probe(N, H, W, seed):
    pr = DProbe(amd, N, H, W, seed)
    r64, r32 = pr.r64, pr.r32
    print(f"== D forward {N}x256x{H}x{W} (seed {seed}):   HIP vs fp64   |   torch-CPU fp32 vs fp64")
    print(f"logits  max-norm rel   {rel(pr.logits, r64['logits']):.2e} | {rel(r32['logits'], r64['logits']):.2e}")
    fg, fc = pr.mask_flips()
    for n in range(3):
        c, y, mean, invstd = pr.saved(n)
        var = 1.0 / invstd.double() ** 2 - orc.BN_EPS
        y64 = r64["y"][n].detach()
        print(f"layer {n}: conv {rel(c, r64['c'][n].detach()):.2e} | {rel(r32['c'][n], r64['c'][n].detach()):.2e}   mean {rel(mean, r64['mean'][n]):.2e} | {rel(r32['mean'][n], r64['mean'][n]):.2e}"
              f"   var {rel(var, r64['var'][n]):.2e} | {rel(r32['var'][n], r64['var'][n]):.2e}   act {rel(y, y64):.2e} | {rel(r32['y'][n], y64):.2e}"
              f"   mask flips {fg[n]} | {fc[n]} of {y64.numel()}")
    for tag in ("own forward", "fp64 forward's activations injected"):
        if tag != "own forward":
            pr.inject_fp64_forward()
        e = pr.errors(*pr.backward())
        print(f"backward [{tag}]: dx L2 {e['dx_l2']:.2e} max-norm {e['dx_max']:.2e};  worst tensor L2 {e['worst_l2'][0]:.2e} ({e['worst_l2'][1]}), "
              f"max-norm {e['worst_max'][0]:.2e} ({e['worst_max'][1]})")
    e = pr.cpu_fp32_backward_errors()
    print(f"backward [torch-CPU fp32]: dx L2 {e['dx_l2']:.2e} max-norm {e['dx_max']:.2e};  worst tensor L2 {e['worst_l2'][0]:.2e} ({e['worst_l2'][1]})")


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    cases = [tuple(a[i:i + 4]) for i in range(0, len(a), 4)] or [(2, 13, 21, 1), (1, 7, 11, 11), (1, 25, 42, 3)]
    for cs in cases:
        probe(*cs)
