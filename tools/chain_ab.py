"""The interpolator's forward / backward with the dense blocks' chain kernel (option g_rdb_chain) on and off: every output, the input
gradient and each parameter gradient of the two schedules against each other (same products in another order: fp32 rounding only).
Usage: python tools/chain_ab.py [N H W]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import afigan_amd as amd
from afigan_amd import _lib, ops
lib = _lib.load()
N, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (1, 25, 34)
torch.manual_seed(0)
G = amd.Generator(n_residual_dense_blocks=3).cuda()
x = ops.pixel_major(torch.randn(N, 256, H, W).cuda())
params = G._ordered_params(); prm, keep = G._param_struct(params)
names = {id(p): n for n, p in G.named_parameters()}
nf = lib.afi_generator_fwd_ws_floats(256, 32, 3, N, H, W); nb = lib.afi_generator_bwd_ws_floats(256, 32, 3, N, H, W)
dout = ops.new_pixel_major(N, 256, 2 * H, 2 * W, "cuda"); dout.normal_()
res = {}
runs = [int(v) for v in os.environ.get("CHAIN_RUNS", "0,1").split(",")]      # CHAIN_RUNS=0,0 : the same schedule twice (determinism check)
for ri, chain in enumerate(runs):
    _lib.current_ctx().set_option("g_rdb_chain", chain)
    ws = torch.zeros(nf, device="cuda"); sc = torch.zeros(nb, device="cuda")
    grads = [torch.zeros_like(p) for p in params]; gst, _ = G._param_struct(grads, already_packed=True)
    out = ops.new_pixel_major(N, 256, 2 * H, 2 * W, "cuda"); dx = ops.new_pixel_major(N, 256, H, W, "cuda")
    st = ops.stream_ptr()
    _lib.call("afi_generator_fwd", C.byref(prm), ops.view_of(x), N, H, W, ops.view_of(out), C.c_void_p(ws.data_ptr()), nf, st)
    _lib.call("afi_generator_bwd", C.byref(prm), C.byref(gst), ops.view_of(x), N, H, W, C.c_void_p(ws.data_ptr()), C.c_void_p(dout.data_ptr()), C.c_void_p(dx.data_ptr()), C.c_void_p(sc.data_ptr()), nb, st)
    torch.cuda.synchronize()
    res[ri] = {"out": out.clone(), "dx": dx.clone(), **{names[id(p)]: g.clone() for p, g in zip(params, grads)}}
rel = lambda a, b: ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()
if os.environ.get("CHAIN_REF"):                            # both schedules against the oracle's op sequence in fp64 on the host (test infrastructure)
    from oracle import afigan_oracle as orc
    pr = {k: v.detach().double().cpu().clone().requires_grad_(True) for k, v in G.state_dict().items()}
    xr = x.detach().double().cpu().clone().requires_grad_(True)
    ref = orc.generator_forward(xr, pr, 3)
    (ref * dout.double().cpu()).sum().backward()
    want = {"out": ref.detach(), "dx": xr.grad, **{k: v.grad for k, v in pr.items()}}
    for ri in res:
        errs = {k: rel(res[ri][k].double().cpu(), want[k]) for k in res[ri]}
        wk = max(errs, key=errs.get)
        print(f"run {ri} (chain {runs[ri]}) vs fp64: worst {errs[wk]:.3e} at {wk}; head weight {errs['Generators.0.0.0.weight']:.3e}; out {errs['out']:.3e} dx {errs['dx']:.3e}")
        if errs[wk] > 1e-4:
            for k in res[ri]:
                d = (res[ri][k].double().cpu() - want[k]).abs(); sc_ = want[k].abs().max()
                bad = (d > 1e-5 * sc_)
                print(f"    {k:44s} err {errs[k]:.2e}  elements off {int(bad.sum())} of {bad.numel()}  mean err {float(d.mean() / sc_):.2e}")
worst = 0.0
for k in res[0]:
    e = rel(res[1][k].double(), res[0][k].double())
    worst = max(worst, e)
    flag = "  <<<" if e > 1e-4 else ""
    print(f"{k:44s} {e:.3e}{flag}")
    if e > 1e-4 and res[0][k].dim() == 4:                  # where: per output row (co) / input column (ci) of a weight gradient
        d = (res[1][k] - res[0][k]).abs().double(); sc_ = res[0][k].abs().max().item()
        print("    rows off:", (d.amax(dim=(1, 2, 3)) > 1e-4 * sc_).nonzero().flatten().tolist()[:16], " cols off:", (d.amax(dim=(0, 2, 3)) > 1e-4 * sc_).nonzero().flatten().tolist()[:24])
print("worst", worst)
