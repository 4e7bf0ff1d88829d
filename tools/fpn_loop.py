"""FPN_AFIGAN / PAFPN_AFIGAN top-down merge forward + backward in a loop, for rocprofv3 --kernel-trace --stats (the `fpn_topdown` / `pafpn` legs of
bench.py): python tools/fpn_loop.py [fpn|pafpn] [iterations]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import afigan_amd as amd
import bench

which = sys.argv[1] if len(sys.argv) > 1 else "fpn"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
print(bench.fpn_bench(amd, torch, iters=iters, warmup=3, pafpn=(which == "pafpn")), flush=True)
