"""config-1 AF-interpolator fwd+bwd through the C-ABI in a loop (for rocprofv3 --kernel-trace --stats); prints the eager time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import afigan_amd as amd
opts = [a for a in sys.argv[1:] if "=" in a]             # name=value: context options set first (e.g. g_rdb_chain=1)
args = [a for a in sys.argv[1:] if "=" not in a]
for kv in opts:
    amd._lib.current_ctx().set_option(kv.split("=")[0], int(kv.split("=")[1]))
iters = int(args[0]) if args else 100
r = bench.interp_bench(amd, torch, 1, 25, 34, iters=iters, warmup=10, graph=len(args) > 1)
print({k: r[k] for k in ("ms", "ms_eager", "ms_graph", "ms_host_enqueue", "tflops")}, flush=True)
