"""Per-launch GEMM timings of one AF-interpolator fwd+bwd (HIP-event brackets of the library): kind, shape, us.
Usage: python tools/cfg1_launches.py [N H W]   (default: config 1, 1 25 34)"""
import ctypes as C, os, sys, csv, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import afigan_amd as amd
from afigan_amd import _lib, ops
lib = _lib.load()
N, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (1, 25, 34)
G = amd.Generator(n_residual_dense_blocks=3).cuda()
x = ops.pixel_major(torch.randn(N, 256, H, W).cuda())
params = G._ordered_params(); prm, keep = G._param_struct(params)
grads = [torch.zeros_like(p) for p in params]; gst, _ = G._param_struct(grads, already_packed=True)
nf = lib.afi_generator_fwd_ws_floats(256, 32, 3, N, H, W); nb = lib.afi_generator_bwd_ws_floats(256, 32, 3, N, H, W)
ws = torch.empty(nf, device="cuda"); sc = torch.empty(nb, device="cuda")
out = ops.new_pixel_major(N, 256, 2 * H, 2 * W, "cuda"); dout = ops.new_pixel_major(N, 256, 2 * H, 2 * W, "cuda"); dout.fill_(1.0)
dx = ops.new_pixel_major(N, 256, H, W, "cuda")
def one():
    st = ops.stream_ptr()
    _lib.call("afi_generator_fwd", C.byref(prm), ops.view_of(x), N, H, W, ops.view_of(out), C.c_void_p(ws.data_ptr()), nf, st)
    _lib.call("afi_generator_bwd", C.byref(prm), C.byref(gst), ops.view_of(x), N, H, W, C.c_void_p(ws.data_ptr()), C.c_void_p(dout.data_ptr()), C.c_void_p(dx.data_ptr()), C.c_void_p(sc.data_ptr()), nb, st)
for _ in range(10): one()
torch.cuda.synchronize()
iters = 10
lib.afi_profile_enable(1)
for _ in range(iters): one()
torch.cuda.synchronize()
lib.afi_profile_enable(0)
path = "gpurun_out/cfg1_launches.csv"
os.makedirs("gpurun_out", exist_ok=True)
lib.afi_profile_dump(path.encode())
rows = list(csv.DictReader(open(path)))
per = len(rows) // iters
tot = 0.0
for i in range(per):
    rs = [rows[k * per + i] for k in range(iters)]
    ms = sorted(float(r["ms"]) for r in rs)[iters // 2]
    tot += ms
    r = rs[0]
    print(f"{i:3d} {r['kind'][:48]:48s} rows {r['rows']:>6s} cols {r['cols']:>5s} k {r['k']:>5s} split {r['split']:>3s}  {ms * 1e3:7.1f} us  {r['tflops']:>7s} TF/s")
print("launches", per, "sum of bracketed GEMM launches", round(tot * 1e3, 1), "us")
