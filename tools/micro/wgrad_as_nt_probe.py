"""Probe for DESIGN section 8 "next (a)": the Winograd weight-gradient product dU[a] = Q[a]^T V[a] evaluated by the NT kernel on TRANSPOSED operands
(K = the tiles of a plane), GEMM only -- the transposes are made by torch and the split of B by the library's split kernel, neither is part of the
comparison (run under rocprofv3 --kernel-trace --stats and compare afi_gemm_nt_bf16_dma_kernel with afi_gemm_tn_bf16_kernel)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from afigan_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for planes, T, M, N in [(36, 8448, 1024, 1024), (36, 8448, 1024, 512), (36, 2176, 1024, 1024)]:
    Q = torch.randn(planes, T, M, device=dev, generator=g)
    V = torch.randn(planes, T, N, device=dev, generator=g)
    dU = ops.gemm_tn(Q, V, "bf16x6")
    Qt, Vt = Q.transpose(1, 2).contiguous(), V.transpose(1, 2).contiguous()
    C = ops.gemm_nt(Qt, Vt, "bf16x6")
    print(planes, T, M, N, "NT-form vs TN result rel diff", ((C - dU).norm() / dU.norm()).item(), flush=True)
    for _ in range(5):
        ops.gemm_tn(Q, V, "bf16x6", out=dU)
        ops.gemm_nt(Qt, Vt, "bf16x6", out=C)
    torch.cuda.synchronize()
    del Q, V, Qt, Vt, dU, C
