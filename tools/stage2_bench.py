import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import afigan_amd as amd
import bench
print(json.dumps(bench.stage2_bench(amd, torch)))
