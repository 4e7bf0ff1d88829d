"""AF-interpolator fwd+bwd over map sizes between the small-map and the Winograd regimes.  Threshold scans: name=value pairs set context
options first (g_winograd_min_pixels, g_smallmap_max_pixels, g_grouped_wgrad_max_pixels).  Usage: python tools/interp_sweep.py [opt=v ...] [N H W] ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import afigan_amd as amd
for kv in [v for v in sys.argv[1:] if "=" in v]:
    amd._lib.current_ctx().set_option(kv.split("=")[0], int(kv.split("=")[1]))
a = [int(v) for v in sys.argv[1:] if "=" not in v]
cases = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)] or [(1, 25, 34), (1, 25, 42), (2, 25, 34), (1, 50, 68), (1, 50, 84), (8, 25, 34), (1, 100, 168)]
for n, h, w in cases:
    r = bench.interp_bench(amd, torch, n, h, w, iters=30, warmup=5, graph=False)
    print(f"{n}x256x{h}x{w} P={n*h*w:6d}  {r['ms']:.3f} ms  {r['out_mpix_per_s']:.2f} Mpix/s  {r['tflops']:.1f} TF/s", flush=True)
