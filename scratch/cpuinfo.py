import os, time, sys
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "ERR", e)
os.system("lscpu | head -20; free -g | head -2")
import torch
sys.path.insert(0, os.getcwd())
from oracle import afigan_oracle as orc
for nt in (16, 32):
    torch.set_num_threads(nt)
    gen = torch.Generator().manual_seed(0)
    gp = orc.reference_init_generator_params(generator=gen)
    x = torch.randn((1,256,25,34), generator=gen).requires_grad_(True)
    gq = {k: v.clone().requires_grad_(True) for k, v in gp.items()}
    for i in range(3):
        t=time.perf_counter(); orc.generator_forward(x, gq).sum().backward(); print(nt, "G fwd+bwd", time.perf_counter()-t, flush=True)
