"""AF-interpolator fwd+bwd over map sizes between the small-map and the Winograd regimes (threshold scans: AFI_G_WINO_MINPIX,
AFI_RDB_BATCH_MAXP, AFI_WG_GROUP_MAXP).  Usage: python tools/interp_sweep.py [N H W] ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import afigan_amd as amd
a = [int(v) for v in sys.argv[1:]]
cases = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)] or [(1, 25, 34), (1, 25, 42), (2, 25, 34), (1, 50, 68), (1, 50, 84), (8, 25, 34), (1, 100, 168)]
for n, h, w in cases:
    r = bench.interp_bench(amd, torch, n, h, w, iters=30, warmup=5, graph=False)
    print(f"{n}x256x{h}x{w} P={n*h*w:6d}  {r['ms']:.3f} ms  {r['out_mpix_per_s']:.2f} Mpix/s  {r['tflops']:.1f} TF/s", flush=True)
