"""Tensor-level wrappers over the per-op C-ABI entry points (used by the modules, the stage-1 engine and the tests).

All tensors are logically NCHW (what detectron2 hands around) but must be *pixel-major* in memory:
``stride(1) == 1`` (torch.channels_last, or any crop / channel slice of such a tensor).  ``pixel_major`` converts an
NCHW-contiguous tensor with the library's own transpose kernel.
"""
import contextlib
import ctypes as C
import os

import torch

from . import _lib
from ._lib import View, call

_NULL_VIEW = View(None, 0, 0, 0)


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check_cuda(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.AfiError("the AFI-GAN hot path runs on the GPU only (got a CPU tensor); there is no CPU fallback")
        if t.dtype != torch.float32:
            raise _lib.AfiError(f"fp32 only on this path, got {t.dtype}")


def is_pixel_major(t: torch.Tensor) -> bool:
    return t.dim() == 4 and (t.stride(1) == 1 or t.size(1) == 1) and all(s % 4 == 0 for s in (t.stride(0), t.stride(2), t.stride(3))) \
        and t.data_ptr() % 16 == 0


_POISON = os.environ.get("AFI_POISON_WS", "0") != "0"     # debugging aid: workspaces start as NaN, so a read of memory the library did not
                                                            # write first shows up in the results instead of depending on what the allocator recycled


def new_workspace(floats: int, device) -> torch.Tensor:
    """Uninitialised fp32 scratch for a library call (NaN-filled under AFI_POISON_WS=1)."""
    if _POISON:
        return torch.full((int(floats),), float("nan"), device=device, dtype=torch.float32)
    return torch.empty(int(floats), device=device, dtype=torch.float32)


def new_pixel_major(N, C_, H, W, device, zero=False) -> torch.Tensor:
    """Fresh [N,C,H,W] tensor whose memory is [N][H][W][C]."""
    if _POISON and not zero:
        return torch.full((N, H, W, C_), float("nan"), device=device, dtype=torch.float32).permute(0, 3, 1, 2)
    f = torch.zeros if zero else torch.empty
    return f((N, H, W, C_), device=device, dtype=torch.float32).permute(0, 3, 1, 2)


def pixel_major(t: torch.Tensor) -> torch.Tensor:
    """Return `t` itself when it is already pixel-major, else an NHWC copy made by afi_nchw_to_nhwc."""
    _check_cuda(t)
    if is_pixel_major(t):
        return t
    src = t if t.is_contiguous() else t.contiguous()
    N, C_, H, W = src.shape
    out = new_pixel_major(N, C_, H, W, t.device)
    call("afi_nchw_to_nhwc", C.c_void_p(src.data_ptr()), C.c_void_p(out.data_ptr()), N, C_, H * W, stream_ptr())
    return out


def to_nchw_contiguous(t: torch.Tensor) -> torch.Tensor:
    """Dense pixel-major tensor -> NCHW-contiguous copy (afi_nhwc_to_nchw)."""
    _check_cuda(t)
    N, C_, H, W = t.shape
    assert t.permute(0, 2, 3, 1).is_contiguous()
    out = torch.empty((N, C_, H, W), device=t.device, dtype=torch.float32)
    call("afi_nhwc_to_nchw", C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()), N, C_, H * W, stream_ptr())
    return out


def view_of(t: torch.Tensor, c0: int = 0) -> View:
    """afi_view_t of a pixel-major tensor (optionally starting at channel c0)."""
    if not is_pixel_major(t):
        raise _lib.AfiError(f"tensor is not pixel-major: shape {tuple(t.shape)} strides {t.stride()}")
    return View(t.data_ptr() + 4 * c0, t.stride(0), t.stride(2), t.stride(3))


def is_dense_pm(t: torch.Tensor) -> bool:
    return t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous()


def ohwi(w: torch.Tensor) -> torch.Tensor:
    """[O,I,kh,kw] weight whose memory is [O][kh][kw][I] (what the kernels read); copies only if it is not already."""
    if w.permute(0, 2, 3, 1).is_contiguous():
        return w
    return keep_alive(w.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2))


def new_ohwi(O, I, kh, kw, device, zero=True) -> torch.Tensor:
    f = torch.zeros if zero else torch.empty
    return f((O, kh, kw, I), device=device, dtype=torch.float32).permute(0, 3, 1, 2)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


def _dense(t):
    """sizes/strides describe a permutation of a contiguous block (what zeros_like can reproduce exactly)"""
    expect = 1
    for size, stride in sorted(((sz, st) for sz, st in zip(t.size(), t.stride()) if sz != 1), key=lambda p: p[1]):
        if stride != expect:
            return False
        expect *= size
    return True


@contextlib.contextmanager
def weight_transform_cache(device, floats=32 * 1024 * 1024):
    """Register a buffer for transformed / packed conv weights (afi_ctx_set_wino_weight_cache) with the active context for the duration
    of a block in which weight VALUES do not change -- e.g. one backbone forward, where the interpolator runs 3 (FPN) to 28 (BiFPN) times
    on one set of weights.  The cache is keyed by weight ADDRESS, so every temporary weight copy made inside the block (``ohwi`` of a
    parameter that is not stored in the kernels' layout, ``.contiguous()`` of a bias) is kept alive until the block exits: a freed
    temporary's address could otherwise be handed to another same-shaped weight and hit the first one's transform.  Not re-entrant."""
    cx = _lib.current_ctx()
    buf = cx.bufs.get("wcache")
    if buf is None or buf.numel() < floats:
        buf = cx.bufs["wcache"] = new_workspace(floats, device)
    if cx.keep is not None:
        raise _lib.AfiError("weight_transform_cache is not re-entrant")
    cx.keep = []
    call("afi_ctx_set_wino_weight_cache", cx.handle, _p(buf), floats)
    try:
        yield
    finally:
        call("afi_ctx_set_wino_weight_cache", cx.handle, C.c_void_p(None), 0)
        cx.keep = None


def keep_alive(t: torch.Tensor) -> torch.Tensor:
    """Hold a temporary until the open weight_transform_cache block (if any) exits; returns it."""
    cx = _lib.current_ctx()
    if cx.keep is not None:
        cx.keep.append(t)
    return t


def zeros_like_many(tensors, need):
    """``[torch.zeros_like(t) if n else None ...]`` out of ONE zero-filled allocation (one fill kernel instead of one per
    tensor: 23 launches per interpolator backward otherwise).  Each result keeps its tensor's sizes and strides (the
    [O][kh][kw][I] weight layout) and starts 16-byte aligned."""
    offs, total = [], 0
    for t, n in zip(tensors, need):
        if n:
            assert t.dtype == torch.float32 and _dense(t), (t.dtype, t.size(), t.stride())
            offs.append(total)
            total += (t.numel() + 3) & ~3
        else:
            offs.append(None)
    if total == 0:
        return [None] * len(offs)
    ref = next(t for t, n in zip(tensors, need) if n)
    flat = torch.zeros(total, device=ref.device, dtype=torch.float32)
    return [flat.as_strided(t.size(), t.stride(), o) if o is not None else None for t, o in zip(tensors, offs)]


def gemm_nt(A, B, dtype, out=None):
    """afi_gemm_nt for tests and micro-benchmarks: C[g] = A[g] @ B[g]^T over the planes of dense [planes, rows, K] / [planes, N, K] tensors,
    in the arithmetic `dtype` ("fp32", "f16x3", "bf16x6", "bf16x3", "bf16"); allocates the split-operand scratch the emulated settings need."""
    _check_cuda(A, B)
    planes, rows, K = A.shape
    N = B.shape[1]
    lib = _lib.load()
    dt = _lib.DTYPES[dtype]
    nb = lib.afi_gemm_nt_scratch_bytes(planes, N, K, dt)
    scratch = torch.empty(max(int(nb), 16), device=A.device, dtype=torch.uint8)
    if out is None:
        out = torch.empty((planes, rows, N), device=A.device, dtype=torch.float32)
    _lib.check(lib.afi_gemm_nt(_p(A), _p(B), _p(out), planes, rows, N, K, dt, _p(scratch), nb, stream_ptr()), "afi_gemm_nt")
    return out


def gemm_tn(Q, V, dtype, out=None):
    """afi_gemm_tn for tests and micro-benchmarks: dU[g] += Q[g]^T @ V[g] over the planes of dense [planes, rows, M] / [planes, rows, N]
    tensors in the arithmetic `dtype`; `out` ([planes, M, N], zero-filled when omitted) is accumulated into, as the weight gradient is."""
    _check_cuda(Q, V)
    planes, rows, M = Q.shape
    N = V.shape[2]
    if out is None:
        out = torch.zeros((planes, M, N), device=Q.device, dtype=torch.float32)
    lib, dt = _lib.load(), _lib.DTYPES[dtype]
    nb = lib.afi_gemm_tn_scratch_bytes(planes, dt)
    scratch = torch.empty(max(int(nb), 16), device=Q.device, dtype=torch.uint8)
    _lib.check(lib.afi_gemm_tn(_p(Q), _p(V), _p(out), planes, rows, M, N, dt, _p(scratch), nb, stream_ptr()), "afi_gemm_tn")
    return out


# ------------------------------------------------------------------------------------------------ convs
OP_SCRATCH_FLOATS = 100 * 1024 * 1024          # 400 MB: 4 slabs of the largest map that is split (1536 tiles of 128x128)


def _ensure_op_scratch(device):
    """Split-K scratch for the per-op conv calls (include/afigan_hip.h: afi_ctx_set_op_scratch): one buffer per context, registered
    once; all per-op calls of a context are issued on one stream, one after the other."""
    cx = _lib.current_ctx()
    buf = cx.bufs.get("op_scratch")
    if buf is None:
        buf = cx.bufs["op_scratch"] = new_workspace(OP_SCRATCH_FLOATS, device)
        call("afi_ctx_set_op_scratch", cx.handle, C.c_void_p(buf.data_ptr()), buf.numel())
    return buf


def conv3x3_fwd(x, w, bias=None, lrelu=False, out=None, alpha=1.0, beta=0.0):
    _check_cuda(x, w, bias, out)
    _ensure_op_scratch(x.device)
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    w = ohwi(w)
    if out is None:
        out = new_pixel_major(N, Cout, H, W, x.device)
    call("afi_conv3x3_fwd", view_of(x), N, H, W, Cin, _p(w), _p(bias), Cout, view_of(out), float(alpha), float(beta), int(lrelu),
         stream_ptr())
    return out


def conv3x3_dgrad(dy, w, dx=None, alpha=1.0, beta=0.0, z=None):
    _check_cuda(dy, w, dx, z)
    _ensure_op_scratch(dy.device)
    N, Cout, H, W = dy.shape
    Cin = w.shape[1]
    w = ohwi(w)
    if dx is None:
        dx = new_pixel_major(N, Cin, H, W, dy.device)
    call("afi_conv3x3_dgrad", view_of(dy), N, H, W, Cout, _p(w), Cin, view_of(dx), float(alpha), float(beta),
         view_of(z) if z is not None else _NULL_VIEW, stream_ptr())
    return dx


def conv3x3_wgrad(dy, x, dw=None, alpha=1.0):
    _check_cuda(dy, x, dw)
    N, Cout, H, W = dy.shape
    Cin = x.shape[1]
    if dw is None:
        dw = new_ohwi(Cout, Cin, 3, 3, dy.device)
    assert dw.permute(0, 2, 3, 1).is_contiguous()
    call("afi_conv3x3_wgrad", view_of(dy), view_of(x), N, H, W, Cout, Cin, _p(dw), float(alpha), stream_ptr())
    return dw


def conv3x3_wino_fwd(x, w, bias=None):
    """3x3 conv in Winograd F(2x2,3x3) form (large maps, many channels): out = conv(x, w) + bias."""
    _check_cuda(x, w, bias)
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    w = ohwi(w)
    n = _lib.load().afi_conv3x3_wino_ws_floats(N, H, W, Cin, Cout)
    ws = new_workspace(n, x.device)
    out = new_pixel_major(N, Cout, H, W, x.device)
    call("afi_conv3x3_wino_fwd", view_of(x), N, H, W, Cin, _p(w), _p(bias), Cout, view_of(out), _p(ws), n, stream_ptr())
    return out


def conv3x3_wino_infer(x, w, bias=None, act=0):
    """act(conv3x3(x, w) + bias) for inference (no backward): Winograd, F(4x4) tiles on maps of >= 8192 pixels.  act: 0 / 1 LeakyReLU / 2 ReLU."""
    _check_cuda(x, w, bias)
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    w = ohwi(w)
    n = _lib.load().afi_conv3x3_wino_ws_floats(N, H, W, Cin, Cout)
    ws = new_workspace(n, x.device)
    out = new_pixel_major(N, Cout, H, W, x.device)
    call("afi_conv3x3_wino_infer", view_of(x), N, H, W, Cin, _p(w), _p(bias), Cout, view_of(out), int(act), _p(ws), n, stream_ptr())
    return out


def conv3x3_wino_dgrad(dy, w, z=None):
    """Data gradient of the 3x3 conv in Winograd form: dx = conv^T(dy) [* lrelu'(z)]."""
    _check_cuda(dy, w, z)
    N, Cout, H, W = dy.shape
    Cin = w.shape[1]
    w = ohwi(w)
    n = _lib.load().afi_conv3x3_wino_ws_floats(N, H, W, Cin, Cout)
    ws = new_workspace(n, dy.device)
    dx = new_pixel_major(N, Cin, H, W, dy.device)
    call("afi_conv3x3_wino_dgrad", view_of(dy), N, H, W, Cout, _p(w), Cin, view_of(dx), view_of(z) if z is not None else _NULL_VIEW,
         _p(ws), n, stream_ptr())
    return dx


def conv3x3_wino_wgrad(dy, x, dw=None, alpha=1.0):
    """Weight gradient of the 3x3 conv in Winograd F(3x3,2x2) form; accumulates into dw ([Cout,Cin,3,3] with OHWI memory)."""
    _check_cuda(dy, x, dw)
    N, Cout, H, W = dy.shape
    Cin = x.shape[1]
    if dw is None:
        dw = new_ohwi(Cout, Cin, 3, 3, dy.device)
    assert dw.permute(0, 2, 3, 1).is_contiguous()
    n = _lib.load().afi_conv3x3_wino_ws_floats(N, H, W, Cin, Cout)
    ws = new_workspace(n, dy.device)
    call("afi_conv3x3_wino_wgrad", view_of(dy), view_of(x), N, H, W, Cout, Cin, _p(dw), float(alpha), _p(ws), n, stream_ptr())
    return dw


def conv3x3s2_fwd(x, w, bias=None, act=0, add=None, add_scale=1.0, post_scale=1.0, keep_act=False):
    """Conv2d(k3, s2, p1) with the fused PAFPN merge:  a = act(conv(x, w) + bias);  out = post_scale*a + add_scale*add.
    act: 0 none, 1 LeakyReLU(0.2), 2 ReLU.  Returns out, or (out, a) with keep_act (a is what the ReLU backward needs)."""
    _check_cuda(x, w, bias, add)
    _ensure_op_scratch(x.device)
    N, Cin, Hi, Wi = x.shape
    Cout = w.shape[0]
    w = ohwi(w)
    Ho, Wo = (Hi + 1) // 2, (Wi + 1) // 2
    if add is not None and tuple(add.shape) != (N, Cout, Ho, Wo):
        raise _lib.AfiError(f"conv3x3s2_fwd: residual shape {tuple(add.shape)} != {(N, Cout, Ho, Wo)}")
    out = new_pixel_major(N, Cout, Ho, Wo, x.device)
    a = new_pixel_major(N, Cout, Ho, Wo, x.device) if keep_act else None
    call("afi_conv3x3s2_fwd", view_of(x), N, Hi, Wi, Cin, _p(w), _p(bias), Cout, view_of(out), int(act),
         view_of(a) if a is not None else _NULL_VIEW, float(post_scale), view_of(add) if add is not None else _NULL_VIEW,
         float(add_scale), stream_ptr())
    return (out, a) if keep_act else out


def conv3x3s2_dgrad(dy, w, in_hw, dx=None, alpha=1.0, beta=0.0):
    """dx [N,Cin,Hi,Wi] = alpha * (data gradient of Conv2d(k3,s2,p1)) + beta*dx;  in_hw = (Hi, Wi) of the conv's input."""
    _check_cuda(dy, w, dx)
    _ensure_op_scratch(dy.device)
    N, Cout, Ho, Wo = dy.shape
    Hi, Wi = in_hw
    if ((Hi + 1) // 2, (Wi + 1) // 2) != (Ho, Wo):
        raise _lib.AfiError(f"conv3x3s2_dgrad: dy {Ho}x{Wo} is not the stride-2 output of {Hi}x{Wi}")
    Cin = w.shape[1]
    w = ohwi(w)
    if dx is None:
        dx = new_pixel_major(N, Cin, Hi, Wi, dy.device)
    call("afi_conv3x3s2_dgrad", view_of(dy), N, Hi, Wi, Cout, _p(w), Cin, view_of(dx), float(alpha), float(beta), stream_ptr())
    return dx


def conv3x3s2_wgrad(dy, x, dw=None, alpha=1.0):
    _check_cuda(dy, x, dw)
    N, Cout, Ho, Wo = dy.shape
    _, Cin, Hi, Wi = x.shape
    if ((Hi + 1) // 2, (Wi + 1) // 2) != (Ho, Wo):
        raise _lib.AfiError(f"conv3x3s2_wgrad: dy {Ho}x{Wo} is not the stride-2 output of {Hi}x{Wi}")
    if dw is None:
        dw = new_ohwi(Cout, Cin, 3, 3, dy.device)
    assert dw.permute(0, 2, 3, 1).is_contiguous()
    call("afi_conv3x3s2_wgrad", view_of(dy), view_of(x), N, Hi, Wi, Cout, Cin, _p(dw), float(alpha), stream_ptr())
    return dw


def relu_bwd(g, act, scale=1.0):
    """scale * g * (act > 0) for dense pixel-major g / act of the same shape."""
    _check_cuda(g, act)
    assert is_dense_pm(g) and is_dense_pm(act) and g.shape == act.shape
    out = new_pixel_major(*g.shape, g.device)
    call("afi_relu_bwd", _p(g), _p(act), _p(out), g.numel(), float(scale), stream_ptr())
    return out


def conv1x1_fwd(x, w, bias=None, add=None, add_scale=1.0, alpha=1.0, out=None, act=0):
    """out = act(alpha*conv1x1(x, w) + bias + add_scale*add);  w: [Cout, Cin] or [Cout, Cin, 1, 1];  act: 0 none, 1 LeakyReLU(0.2), 2 ReLU.
    ``x`` may be a strided view of a pixel-major tensor (a crop, or every second pixel: a stride-2 1x1 conv reads it in place)."""
    _check_cuda(x, w, bias, add, out)
    _ensure_op_scratch(x.device)
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    w2 = w.reshape(Cout, Cin).contiguous()
    if out is None:
        out = new_pixel_major(N, Cout, H, W, x.device)
    call("afi_conv1x1_fwd", view_of(x), N, H, W, Cin, _p(w2), _p(bias), Cout, view_of(out), float(alpha), 0.0,
         view_of(add) if add is not None else _NULL_VIEW, float(add_scale), int(act), stream_ptr())
    return out


def conv1x1_dgrad(dy, w, dx=None, alpha=1.0, beta=0.0):
    _check_cuda(dy, w, dx)
    _ensure_op_scratch(dy.device)
    N, Cout, H, W = dy.shape
    Cin = w.shape[1]
    w2 = w.reshape(Cout, Cin).contiguous()
    if dx is None:
        dx = new_pixel_major(N, Cin, H, W, dy.device)
    call("afi_conv1x1_dgrad", view_of(dy), N, H, W, Cout, _p(w2), Cin, view_of(dx), float(alpha), float(beta), stream_ptr())
    return dx


def conv1x1_wgrad(dy, x, alpha=1.0):
    _check_cuda(dy, x)
    N, Cout, H, W = dy.shape
    Cin = x.shape[1]
    dw = torch.zeros((Cout, Cin), device=dy.device, dtype=torch.float32)
    call("afi_conv1x1_wgrad", view_of(dy), view_of(x), N, H, W, Cout, Cin, _p(dw), float(alpha), stream_ptr())
    return dw


def bias_grad(dy):
    """Column sums of a dense pixel-major gradient: d/d bias of a conv."""
    assert is_dense_pm(dy)
    N, C_, H, W = dy.shape
    db = torch.zeros(C_, device=dy.device, dtype=torch.float32)
    colsum_accum(dy.permute(0, 2, 3, 1).reshape(-1, C_), db)
    return db


def convT_pack(w_iohw):
    Cin, Cout = w_iohw.shape[:2]
    w_iohw = w_iohw.contiguous()
    wp = torch.empty((4 * Cout, 3, 3, Cin), device=w_iohw.device, dtype=torch.float32)
    call("afi_convT6s2_pack_weight", _p(w_iohw), _p(wp), Cin, Cout, stream_ptr())
    return wp


def convT_fwd(x, wp, bias, Cout, lrelu=False):
    _ensure_op_scratch(x.device)
    N, Cin, H, W = x.shape
    out = new_pixel_major(N, Cout, 2 * H, 2 * W, x.device)
    call("afi_convT6s2_fwd", view_of(x), N, H, W, Cin, _p(wp), _p(bias), Cout, view_of(out), int(lrelu), stream_ptr())
    return out


def convT_dgrad(dy, wp, Cin, z=None):
    _ensure_op_scratch(dy.device)
    N, Cout, H2, W2 = dy.shape
    H, W = H2 // 2, W2 // 2
    dx = new_pixel_major(N, Cin, H, W, dy.device)
    call("afi_convT6s2_dgrad", view_of(dy), N, H, W, Cout, _p(wp), Cin, view_of(dx), view_of(z) if z is not None else _NULL_VIEW,
         stream_ptr())
    return dx


def convT_wgrad(dy, x):
    """Returns the gradient in torch's [Cin][Cout][6][6] layout."""
    N, Cout, H2, W2 = dy.shape
    _, Cin, H, W = x.shape
    dwp = torch.zeros((4 * Cout, 3, 3, Cin), device=dy.device, dtype=torch.float32)
    call("afi_convT6s2_wgrad", view_of(dy), view_of(x), N, H, W, Cout, Cin, _p(dwp), 1.0, stream_ptr())
    dw = torch.zeros((Cin, Cout, 6, 6), device=dy.device, dtype=torch.float32)
    call("afi_convT6s2_unpack_wgrad", _p(dwp), _p(dw), Cin, Cout, stream_ptr())
    return dw


# ------------------------------------------------------------------------------------------------ BiFPN inference pieces
def dwconv3x3(x, w9c):
    """Depthwise 3x3 (stride 1, zero pad 1, no bias); w9c: [9, C] tap-major weights (see bifpn_sr.pack_depthwise)."""
    _check_cuda(x, w9c)
    N, C_, H, W = x.shape
    assert tuple(w9c.shape) == (9, C_) and w9c.is_contiguous()
    out = new_pixel_major(N, C_, H, W, x.device)
    call("afi_dwconv3x3_fwd", view_of(x), N, H, W, C_, _p(w9c), _p(out), stream_ptr())
    return out


def maxpool3s2_same(x):
    """MaxPool2d(3, 2, padding_mode="static_same") of bifpn_layers/wrappers.py (zero pad right/bottom)."""
    _check_cuda(x)
    N, C_, H, W = x.shape
    out = new_pixel_major(N, C_, (H - 2) // 2 + 1, (W - 2) // 2 + 1, x.device)
    call("afi_maxpool3s2_same_fwd", view_of(x), N, H, W, C_, _p(out), stream_ptr())
    return out


def fuse_swish(w, a, b, c=None):
    """swish(w[0]*a + w[1]*b (+ w[2]*c)) for dense pixel-major tensors of one shape; w: device tensor with 2 or 3 weights."""
    _check_cuda(w, a, b, c)
    assert is_dense_pm(a) and is_dense_pm(b) and a.shape == b.shape and (c is None or (is_dense_pm(c) and c.shape == a.shape))
    assert w.numel() == (2 if c is None else 3) and w.is_contiguous()
    out = new_pixel_major(*a.shape, a.device)
    call("afi_fuse_swish_fwd", _p(a), _p(b), _p(c), _p(w), _p(out), a.numel(), stream_ptr())
    return out


# ------------------------------------------------------------------------------------------------ BiFPN node backward (training mode)
def fuse_swish_bwd(w, a, b, c, dout, need=(True, True, True, True)):
    """Backward of fuse_swish: returns (dw, da, db, dc); entries not in `need` (w, a, b, c) come back None."""
    _check_cuda(w, a, b, c, dout)
    assert is_dense_pm(dout) and dout.shape == a.shape
    dev = a.device
    da = new_pixel_major(*a.shape, dev) if need[1] else None
    db = new_pixel_major(*a.shape, dev) if need[2] else None
    dc = new_pixel_major(*a.shape, dev) if (c is not None and need[3]) else None
    dw = torch.empty_like(w) if need[0] else None
    scratch = new_workspace(_lib.load().afi_fuse_swish_bwd_scratch_floats(), dev)
    call("afi_fuse_swish_bwd", _p(a), _p(b), _p(c), _p(w), _p(dout), _p(da), _p(db), _p(dc), _p(dw), a.numel(), _p(scratch), stream_ptr())
    return dw, da, db, dc


def dwconv3x3_wgrad(dy, x):
    """[9, C] tap-major weight gradient of dwconv3x3 from dense pixel-major dy and x."""
    _check_cuda(dy, x)
    assert is_dense_pm(dy) and is_dense_pm(x) and dy.shape == x.shape
    N, C_, H, W = x.shape
    dw = torch.empty((9, C_), device=x.device, dtype=torch.float32)
    scratch = new_workspace(_lib.load().afi_dwconv3x3_wgrad_scratch_floats(C_), x.device)
    call("afi_dwconv3x3_wgrad", _p(dy), _p(x), N, H, W, C_, _p(dw), _p(scratch), stream_ptr())
    return dw


def maxpool3s2_same_idx(x):
    """maxpool3s2_same keeping the argmax taps (uint8, [N, Ho, Wo, C]) for maxpool3s2_same_bwd."""
    _check_cuda(x)
    N, C_, H, W = x.shape
    Ho, Wo = (H - 2) // 2 + 1, (W - 2) // 2 + 1
    out = new_pixel_major(N, C_, Ho, Wo, x.device)
    idx = torch.empty((N, Ho, Wo, C_), device=x.device, dtype=torch.uint8)
    call("afi_maxpool3s2_same_fwd_idx", view_of(x), N, H, W, C_, _p(out), _p(idx), stream_ptr())
    return out, idx


def maxpool3s2_same_bwd(dout, idx, in_hw):
    _check_cuda(dout)
    if not idx.is_cuda or idx.dtype != torch.uint8 or not idx.is_contiguous():
        raise _lib.AfiError("idx must be the contiguous uint8 CUDA tensor maxpool3s2_same_idx returned")
    N, C_ = dout.shape[:2]
    H, W = in_hw
    assert is_dense_pm(dout) and tuple(idx.shape) == (N, (H - 2) // 2 + 1, (W - 2) // 2 + 1, C_) and tuple(dout.shape[2:]) == tuple(idx.shape[1:3])
    dx = new_pixel_major(N, C_, H, W, dout.device)
    call("afi_maxpool3s2_same_bwd", _p(dout), _p(idx), N, H, W, C_, _p(dx), stream_ptr())
    return dx


def bn_stats_ex(x2d, eps, momentum, running_mean=None, running_var=None, num_batches_tracked=None, want_var=False):
    """Train-mode statistics of a dense [P, C] matrix with the norm's own eps / momentum; updates the running buffers in place.
    want_var: also the biased batch variance (for a caller that combines the statistics of several ranks)."""
    P, C_ = x2d.shape
    mean, invstd = torch.empty(C_, device=x2d.device), torch.empty(C_, device=x2d.device)
    var = torch.empty(C_, device=x2d.device) if want_var else None
    call("afi_bn_stats_ex", _p(x2d), P, C_, float(eps), float(momentum), _p(mean), _p(invstd), _p(var), _p(running_mean), _p(running_var),
         _p(num_batches_tracked), _p(reduce_scratch(C_, x2d.device)), stream_ptr())
    return (mean, invstd, var) if want_var else (mean, invstd)


def bn_apply(x2d, mean, invstd, gamma, beta, slope=1.0):
    P, C_ = x2d.shape
    y = torch.empty_like(x2d)
    call("afi_bn_apply_fwd", _p(x2d), _p(y), _p(mean), _p(invstd), _p(gamma), _p(beta), P, C_, float(slope), stream_ptr())
    return y


# ------------------------------------------------------------------------------------------------ dual-scale data path
def _check_u8(t):
    if not t.is_cuda:
        raise _lib.AfiError("the dual-scale data path runs on the GPU only (got a CPU tensor); there is no CPU fallback")
    if t.dtype != torch.uint8 or not t.is_contiguous():
        raise _lib.AfiError(f"expected a contiguous uint8 tensor, got {t.dtype}, contiguous={t.is_contiguous()}")


def resize_bilinear_u8(img_hwc, new_h, new_w, hflip=False, chw=True):
    """ResizeTransform.apply_image (Pillow BILINEAR, bit-exact) + the shared HFlipTransform for a uint8 [H,W,C] or [H,W]
    device tensor; returns uint8 [C,new_h,new_w] (chw, the DatasetMapper tensor layout) or [new_h,new_w,C]."""
    _check_u8(img_hwc)
    squeeze = img_hwc.dim() == 2
    H0, W0 = img_hwc.shape[:2]
    Cn = 1 if squeeze else img_hwc.shape[2]
    new_h, new_w = int(new_h), int(new_w)
    lib = _lib.load()
    ws_bytes = lib.afi_resize_bilinear_u8_ws_bytes(H0, W0, Cn, new_h, new_w)
    if ws_bytes < 0:
        raise _lib.AfiError(f"afi_resize_bilinear_u8: unsupported shape {tuple(img_hwc.shape)} -> {(new_h, new_w)}")
    ws = torch.empty(ws_bytes, device=img_hwc.device, dtype=torch.uint8)
    out = torch.empty((Cn, new_h, new_w) if chw else (new_h, new_w, Cn), device=img_hwc.device, dtype=torch.uint8)
    call("afi_resize_bilinear_u8", _p(img_hwc), H0, W0, Cn, _p(out), new_h, new_w, int(bool(hflip)), int(bool(chw)), _p(ws), ws_bytes,
         stream_ptr())
    if squeeze:
        out = out[0] if chw else out[..., 0]
    return out


def dual_scale_u8(img_hwc, size, size_r, hflip=False, hflip_r=False, chw=True):
    """Both images of one DatasetMapper sample in two launches: `image` = resize(img, size) and `image_x0.5` =
    resize(img, size_r), each from the ORIGINAL uint8 [H,W,3] device tensor (dataset_mapper.py:103-105), each with its flip flag."""
    _check_u8(img_hwc)
    if img_hwc.dim() != 3:
        raise _lib.AfiError(f"expected a uint8 HWC image, got shape {tuple(img_hwc.shape)}")
    H0, W0, Cn = img_hwc.shape
    (H1, W1), (H2, W2) = (int(v) for v in size), (int(v) for v in size_r)
    lib = _lib.load()
    ws_bytes = lib.afi_dual_scale_u8_ws_bytes(H0, W0, Cn, H1, W1, H2, W2)
    if ws_bytes < 0:
        raise _lib.AfiError(f"afi_dual_scale_u8: unsupported shape {tuple(img_hwc.shape)} -> {(H1, W1)}, {(H2, W2)}")
    ws = torch.empty(ws_bytes, device=img_hwc.device, dtype=torch.uint8)
    out = torch.empty((Cn, H1, W1) if chw else (H1, W1, Cn), device=img_hwc.device, dtype=torch.uint8)
    out_r = torch.empty((Cn, H2, W2) if chw else (H2, W2, Cn), device=img_hwc.device, dtype=torch.uint8)
    call("afi_dual_scale_u8", _p(img_hwc), H0, W0, Cn, _p(out), H1, W1, int(bool(hflip)), _p(out_r), H2, W2, int(bool(hflip_r)),
         int(bool(chw)), _p(ws), ws_bytes, stream_ptr())
    return out, out_r


def normalize_pad(images_chw, pixel_mean, pixel_std, size_divisibility=0):
    """RCNN_FPN_only.forward's `(x - mean) / std` per image + ImageList.from_tensors (rcnn_only.py:36-39): a list of uint8
    [C,H,W] device tensors -> fp32 [N,C,Hp,Wp], zero-padded bottom/right to the batch maximum rounded up to size_divisibility."""
    assert len(images_chw) > 0
    for t in images_chw:
        _check_u8(t)
    Cn = images_chw[0].shape[0]
    hm = max(t.shape[1] for t in images_chw)
    wm = max(t.shape[2] for t in images_chw)
    if size_divisibility > 0:
        hm = -(-hm // size_divisibility) * size_divisibility
        wm = -(-wm // size_divisibility) * size_divisibility
    mean = (C.c_float * Cn)(*[float(v) for v in pixel_mean])
    std = (C.c_float * Cn)(*[float(v) for v in pixel_std])
    out = torch.empty((len(images_chw), Cn, hm, wm), device=images_chw[0].device, dtype=torch.float32)
    for n, t in enumerate(images_chw):
        assert t.shape[0] == Cn
        call("afi_normalize_pad_u8", _p(t), Cn, t.shape[1], t.shape[2], mean, std, _p(out[n]), hm, wm, stream_ptr())
    return out


# ------------------------------------------------------------------------------------------------ bandwidth ops
def bilinear2x(x, out=None, beta=0.0):
    N, C_, H, W = x.shape
    if out is None:
        out = new_pixel_major(N, C_, 2 * H, 2 * W, x.device)
    assert is_dense_pm(out)
    call("afi_bilinear2x_add_fwd", view_of(x), N, H, W, C_, float(beta), _p(out), stream_ptr())
    return out


def bilinear2x_bwd(dout, dx=None, beta=0.0):
    N, C_, H2, W2 = dout.shape
    assert is_dense_pm(dout)
    if dx is None:
        dx = new_pixel_major(N, C_, H2 // 2, W2 // 2, dout.device)
    call("afi_bilinear2x_add_bwd", _p(dout), N, H2 // 2, W2 // 2, C_, float(beta), _p(dx), stream_ptr())
    return dx


def reduce_scratch(C_, device):
    return new_workspace(_lib.load().afi_reduce_scratch_floats(C_), device)


def bn_stats(x2d, running_mean=None, running_var=None):
    """x2d: dense [P, C].  Returns (mean, invstd, var_biased)."""
    P, C_ = x2d.shape
    mean, invstd, var = (torch.empty(C_, device=x2d.device) for _ in range(3))
    call("afi_bn_stats", _p(x2d), P, C_, _p(mean), _p(invstd), _p(var), _p(running_mean), _p(running_var),
         _p(reduce_scratch(C_, x2d.device)), stream_ptr())
    return mean, invstd, var


def bn_apply_lrelu(x2d, mean, invstd, gamma, beta):
    P, C_ = x2d.shape
    y = torch.empty_like(x2d)
    call("afi_bn_apply_lrelu_fwd", _p(x2d), _p(y), _p(mean), _p(invstd), _p(gamma), _p(beta), P, C_, stream_ptr())
    return y


def bn_bwd(g2d, x2d, mean, invstd, gamma, dgamma, dbeta):
    P, C_ = x2d.shape
    dx = torch.empty_like(x2d)
    call("afi_bn_bwd", _p(g2d), _p(x2d), _p(dx), _p(mean), _p(invstd), _p(gamma), _p(dgamma), _p(dbeta), P, C_,
         _p(reduce_scratch(C_, x2d.device)), stream_ptr())
    return dx


def bn_bwd_sums(g2d, x2d, mean, invstd, dgamma, dbeta):
    """First half of bn_bwd (afi_bn_bwd_sums): [2, C] = (sum g, sum g * xhat) over this tensor's rows; dbeta / dgamma += them."""
    P, C_ = x2d.shape
    sums = torch.empty((2, C_), device=x2d.device)
    call("afi_bn_bwd_sums", _p(g2d), _p(x2d), _p(mean), _p(invstd), _p(dgamma), _p(dbeta), _p(sums), P, C_, _p(reduce_scratch(C_, x2d.device)), stream_ptr())
    return sums


def bn_bwd_apply(g2d, x2d, mean, invstd, gamma, sums, P_total):
    """Second half (afi_bn_bwd_apply): dx from sums taken over P_total rows in all (the caller's all-reduce of bn_bwd_sums)."""
    P, C_ = x2d.shape
    dx = torch.empty_like(x2d)
    call("afi_bn_bwd_apply", _p(g2d), _p(x2d), _p(dx), _p(mean), _p(invstd), _p(gamma), _p(sums), P, int(P_total), C_, stream_ptr())
    return dx


def colsum_accum(g2d, db, alpha=1.0):
    P, C_ = g2d.shape
    call("afi_colsum_accum", _p(g2d), P, C_, g2d.stride(0), float(alpha), _p(db), _p(reduce_scratch(C_, g2d.device)), stream_ptr())
    return db


def bce_logits(z, target, loss, lscale=1.0, gscale=1.0, want_grad=True):
    """*loss += lscale * BCEWithLogits(z, target).mean(); returns dz (or None)."""
    z = z.contiguous()
    dz = torch.empty_like(z) if want_grad else None
    call("afi_bce_logits_fwd_bwd", _p(z), z.numel(), float(target), float(lscale), _p(loss), float(gscale), _p(dz), stream_ptr())
    return dz


def l1_crop(a, b, loss, lscale=1.0, gscale=1.0, want_grad=True):
    """*loss += lscale * l1(a[:, :, :h, :w], b[:, :, :h, :w]) with h, w the common extent; returns da (dense, full extent of a)."""
    N, C_, Ha, Wa = a.shape
    h, w = min(Ha, b.shape[2]), min(Wa, b.shape[3])
    da = new_pixel_major(N, C_, Ha, Wa, a.device) if want_grad else None
    call("afi_l1_fwd_bwd", view_of(a), view_of(b), N, h, w, C_, Ha, Wa, float(lscale), _p(loss), float(gscale), _p(da), stream_ptr())
    return da
