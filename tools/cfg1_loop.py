"""config-1 AF-interpolator fwd+bwd through the C-ABI in a loop (for rocprofv3 --kernel-trace --stats); prints the eager time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import afigan_amd as amd
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
r = bench.interp_bench(amd, torch, 1, 25, 34, iters=iters, warmup=10, graph=len(sys.argv) > 2)
print({k: r[k] for k in ("ms", "ms_eager", "ms_graph", "ms_host_enqueue", "tflops")}, flush=True)
