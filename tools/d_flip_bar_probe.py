"""What tests/test_gpu_d_parity.py bounds, printed instead of asserted: per seed and per `winograd_f4_forward` setting, the LeakyReLU mask flips
of the library's forward and of torch-CPU fp32 against the fp64 oracle, and the own-forward gradient deviation of both.

    python tools/d_flip_bar_probe.py N H W seed [seed ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import afigan_amd as amd
from d_parity_util import DProbe

N, H, W = (int(v) for v in sys.argv[1:4])
for seed in [int(s) for s in sys.argv[4:]]:
    cpu = None
    for f4, ls in ((0, 0), (8, 0), (12, 0), (12, 12), (1, 12)):
        pr = DProbe(amd, N, H, W, seed, options={"winograd_f4_forward": f4, "f16_local_sums": ls})
        fg, fc = pr.mask_flips()
        e = pr.errors(*pr.backward())
        if cpu is None:
            cpu = pr.cpu_fp32_backward_errors()
            print(f"seed {seed} torch-CPU fp32: flips {fc} = {sum(fc)}; dx {cpu['dx_l2']:.3e} worst {cpu['worst_l2'][0]:.3e}", flush=True)
        print(f"seed {seed} f4_forward={f4:2d} local_sums={ls:2d}: flips {fg} = {sum(fg)}; dx {e['dx_l2']:.3e} worst {e['worst_l2'][0]:.3e} ({e['worst_l2'][1]})", flush=True)
        del pr
