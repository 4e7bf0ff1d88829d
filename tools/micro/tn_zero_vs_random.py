import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, afigan_amd as amd
from afigan_amd import _lib
lib=_lib.load()
def kms(fn, tag="gemm_tn", iters=10):
    fn(); torch.cuda.synchronize(); lib.afi_profile_enable(1)
    for _ in range(iters): fn()
    torch.cuda.synchronize(); lib.afi_profile_enable(0)
    for k in range(lib.afi_profile_num_kinds()):
        o=(C.c_double*3)(); lib.afi_profile_get(k,o)
        if o[0]>0 and tag in lib.afi_profile_kind_name(k).decode(): return o[1]/o[0]
for planes,rows,M,N in ((36,8448,1024,1024),(36,8448,1024,512)):
    g=torch.Generator(device="cuda").manual_seed(1)
    Q=torch.randn((planes,rows,M),device="cuda",generator=g); V=torch.randn((planes,rows,N),device="cuda",generator=g)
    out=torch.zeros((planes,M,N),device="cuda")
    fl=2.0*planes*rows*M*N
    for name,(q,v) in (("random",(Q,V)),("zeros",(torch.zeros_like(Q),torch.zeros_like(V)))):
        for dt in ("f16x3","bf16x6"):
            ms=kms(lambda: amd.ops.gemm_tn(q,v,dt,out=out))
            print(f"TN {planes}x{rows}->{M}x{N} {dt:7s} {name:7s} {ms*1e3:8.1f} us {fl/ms/1e9:6.1f} TFLOP/s",flush=True)
