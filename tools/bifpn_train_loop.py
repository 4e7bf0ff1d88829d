"""BiFPN_AFIGAN training forward + backward in a loop, for rocprofv3 --kernel-trace --stats (the `bifpn_training` leg of bench.py without the timing
harness): python tools/bifpn_train_loop.py [iterations]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import afigan_amd as amd
import bench

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
r = bench.bifpn_train_bench(amd, torch, iters=iters, warmup=2)
print({k: v for k, v in r.items() if k != "gemm_kernel_families"}, flush=True)
for f in r["gemm_kernel_families"]:
    print(f, flush=True)
