import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import afigan_amd as amd
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
print(json.dumps(bench.interp_bench(amd, torch, N, 25, 34, iters=30, warmup=5)))
