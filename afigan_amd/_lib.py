"""ctypes binding of libafigan_hip.so (the C-ABI declared in include/afigan_hip.h).

This is the reference-side binding a maintainer would add for the hot path.  There is NO
fallback: if the HIP library is missing or a call returns a non-zero status, an exception is raised.
"""
import ctypes as C
import os

# torch FIRST: the ROCm wheel ships its own libamdhip64.so; importing torch before dlopen()ing libafigan_hip.so makes the
# library's HIP dependency resolve to the runtime torch already loaded.  The other order maps a second HIP runtime into the
# process, and every launch on one of torch's streams then fails ("HIP kernel launch failed" on the first call).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AFI_LIB_PATH") or os.path.join(_HERE, "csrc", "libafigan_hip.so")   # override: A/B kernel builds

AFI_MAX_RDB = 8
ABI_VERSION = 8


class AfiError(RuntimeError):
    pass


class View(C.Structure):
    _fields_ = [("p", C.c_void_p), ("sN", C.c_longlong), ("sH", C.c_longlong), ("sW", C.c_longlong)]


class GenParams(C.Structure):
    _fields_ = [("C", C.c_int), ("G", C.c_int), ("n_rdb", C.c_int), ("residual_scale", C.c_float),
                ("w0", C.c_void_p), ("b0", C.c_void_p),
                ("rdb_w", (C.c_void_p * 5) * AFI_MAX_RDB),
                ("w7", C.c_void_p), ("b7", C.c_void_p),
                ("wT", C.c_void_p), ("bT", C.c_void_p),
                ("w9", C.c_void_p), ("b9", C.c_void_p)]


class DiscParams(C.Structure):
    _fields_ = [("F", C.c_int * 4),
                ("w", C.c_void_p * 3), ("b", C.c_void_p * 3),
                ("gamma", C.c_void_p * 3), ("beta", C.c_void_p * 3),
                ("running_mean", C.c_void_p * 3), ("running_var", C.c_void_p * 3),
                ("num_batches_tracked", C.c_void_p * 3),
                ("w3", C.c_void_p), ("b3", C.c_void_p)]


class SgdDesc(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("n", C.c_longlong),
                ("wd", C.c_float), ("pad_", C.c_float)]


_vp, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong
_GP, _DP = C.POINTER(GenParams), C.POINTER(DiscParams)

# name -> (restype, argtypes); every symbol declared in include/afigan_hip.h
SIGNATURES = {
    "afi_abi_version": (_i, []),
    "afi_build_id": (C.c_char_p, []),
    "afi_status_string": (C.c_char_p, [_i]),
    "afi_ctx_create": (_i, [C.POINTER(C.c_void_p)]),
    "afi_ctx_destroy": (_i, [_vp]),
    "afi_ctx_set_op_scratch": (_i, [_vp, _vp, _ll]),
    "afi_ctx_set_compute_dtype": (_i, [_vp, _i]),
    "afi_ctx_get_compute_dtype": (_i, [_vp]),
    "afi_ctx_set_option": (_i, [_vp, _i, _ll]),
    "afi_ctx_get_option": (_ll, [_vp, _i]),
    "afi_gemm_nt_scratch_bytes": (_ll, [_i, _i, _i, _i]),
    "afi_gemm_nt": (_i, [_vp, _vp, _vp, _i, _ll, _i, _i, _i, _vp, _ll, _vp]),
    "afi_gemm_tn_scratch_bytes": (_ll, [_i, _i]),
    "afi_gemm_tn": (_i, [_vp, _vp, _vp, _i, _ll, _i, _i, _i, _vp, _ll, _vp]),
    "afi_ctx_set_wino_weight_cache": (_i, [_vp, _vp, _ll]),
    "afi_ctx_wino_weight_cache_invalidate": (_i, [_vp]),
    "afi_ctx_set_wino_wgrad_accum": (_i, [_vp, _vp, _ll]),
    "afi_ctx_wino_wgrad_flush": (_i, [_vp, _vp]),
    "afi_ctx_wino_wgrad_discard": (_i, [_vp]),
    "afi_generator_fwd_ws_floats": (_ll, [_i] * 6),
    "afi_generator_bwd_ws_floats": (_ll, [_i] * 6),
    "afi_generator_fwd": (_i, [_vp, _GP, View, _i, _i, _i, View, _vp, _ll, _vp]),
    "afi_generator_bwd": (_i, [_vp, _GP, _GP, View, _i, _i, _i, _vp, _vp, _vp, _vp, _ll, _vp]),
    "afi_discriminator_fwd_ws_floats": (_ll, [C.POINTER(C.c_int), _i, _i, _i]),
    "afi_discriminator_fwd_ws_floats_ex": (_ll, [_vp, C.POINTER(C.c_int), _i, _i, _i, _i]),
    "afi_discriminator_bwd_ws_floats": (_ll, [C.POINTER(C.c_int), _i, _i, _i]),
    "afi_discriminator_ws_layout": (_i, [C.POINTER(C.c_int), _i, _i, _i, C.POINTER(C.c_longlong)]),
    "afi_discriminator_saved_activations": (_i, [_vp, C.POINTER(C.c_int), _i, _i, _i]),
    "afi_discriminator_fwd": (_i, [_vp, _DP, View, _i, _i, _i, _vp, _i, _vp, _ll, _vp]),
    "afi_discriminator_bwd": (_i, [_vp, _DP, _DP, View, _i, _i, _i, _vp, _vp, _vp, _vp, _ll, _vp]),
    "afi_discriminator_fwd_paired": (_i, [_vp, _DP, View, _i, _i, _i, _vp, _i, _vp, _ll, _vp]),
    "afi_discriminator_bwd_paired": (_i, [_vp, _DP, _DP, View, _i, _i, _i, _vp, _vp, _vp, _vp, _ll, _vp]),
    "afi_conv3x3_fwd": (_i, [_vp, View, _i, _i, _i, _i, _vp, _vp, _i, View, _f, _f, _i, _vp]),
    "afi_conv3x3_dgrad": (_i, [_vp, View, _i, _i, _i, _i, _vp, _i, View, _f, _f, View, _vp]),
    "afi_conv3x3_wgrad": (_i, [_vp, View, View, _i, _i, _i, _i, _i, _vp, _f, _vp]),
    "afi_conv1x1_fwd": (_i, [_vp, View, _i, _i, _i, _i, _vp, _vp, _i, View, _f, _f, View, _f, _i, _vp]),
    "afi_conv1x1_dgrad": (_i, [_vp, View, _i, _i, _i, _i, _vp, _i, View, _f, _f, _vp]),
    "afi_conv1x1_wgrad": (_i, [_vp, View, View, _i, _i, _i, _i, _i, _vp, _f, _vp]),
    "afi_conv3x3_wino_ws_floats": (_ll, [_i] * 5),
    "afi_conv3x3_wino_fwd": (_i, [_vp, View, _i, _i, _i, _i, _vp, _vp, _i, View, _vp, _ll, _vp]),
    "afi_conv3x3_wino_infer": (_i, [_vp, View, _i, _i, _i, _i, _vp, _vp, _i, View, _i, _vp, _ll, _vp]),
    "afi_conv3x3_wino_dgrad": (_i, [_vp, View, _i, _i, _i, _i, _vp, _i, View, View, _vp, _ll, _vp]),
    "afi_conv3x3_wino_wgrad": (_i, [_vp, View, View, _i, _i, _i, _i, _i, _vp, _f, _vp, _ll, _vp]),
    "afi_conv3x3s2_fwd": (_i, [_vp, View, _i, _i, _i, _i, _vp, _vp, _i, View, _i, View, _f, View, _f, _vp]),
    "afi_conv3x3s2_dgrad": (_i, [_vp, View, _i, _i, _i, _i, _vp, _i, View, _f, _f, _vp]),
    "afi_conv3x3s2_wgrad": (_i, [_vp, View, View, _i, _i, _i, _i, _i, _vp, _f, _vp]),
    "afi_relu_bwd": (_i, [_vp, _vp, _vp, _ll, _f, _vp]),
    "afi_dwconv3x3_fwd": (_i, [View, _i, _i, _i, _i, _vp, _vp, _vp]),
    "afi_maxpool3s2_same_fwd": (_i, [View, _i, _i, _i, _i, _vp, _vp]),
    "afi_fuse_swish_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _vp]),
    "afi_fuse_swish_bwd_scratch_floats": (_ll, []),
    "afi_fuse_swish_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _vp]),
    "afi_dwconv3x3_wgrad_scratch_floats": (_ll, [_i]),
    "afi_dwconv3x3_wgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "afi_maxpool3s2_same_fwd_idx": (_i, [View, _i, _i, _i, _i, _vp, _vp, _vp]),
    "afi_maxpool3s2_same_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "afi_bn_stats_ex": (_i, [_vp, _ll, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "afi_bn_apply_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _f, _vp]),
    "afi_resize_bilinear_u8_ws_bytes": (_ll, [_i, _i, _i, _i, _i]),
    "afi_resize_bilinear_u8": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _ll, _vp]),
    "afi_dual_scale_u8_ws_bytes": (_ll, [_i, _i, _i, _i, _i, _i, _i]),
    "afi_dual_scale_u8": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _ll, _vp]),
    "afi_normalize_pad_u8": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "afi_convT6s2_pack_weight": (_i, [_vp, _vp, _i, _i, _vp]),
    "afi_convT6s2_unpack_wgrad": (_i, [_vp, _vp, _i, _i, _vp]),
    "afi_convT6s2_fwd": (_i, [_vp, View, _i, _i, _i, _i, _vp, _vp, _i, View, _i, _vp]),
    "afi_convT6s2_dgrad": (_i, [_vp, View, _i, _i, _i, _i, _vp, _i, View, View, _vp]),
    "afi_convT6s2_wgrad": (_i, [_vp, View, View, _i, _i, _i, _i, _i, _vp, _f, _vp]),
    "afi_bilinear2x_add_fwd": (_i, [View, _i, _i, _i, _i, _f, _vp, _vp]),
    "afi_bilinear2x_add_bwd": (_i, [_vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "afi_reduce_scratch_floats": (_ll, [_i]),
    "afi_bn_stats": (_i, [_vp, _ll, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "afi_bn_apply_lrelu_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _vp]),
    "afi_bn_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _vp, _vp]),
    "afi_bn_bwd_sums": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _vp, _vp]),
    "afi_bn_bwd_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _ll, _i, _vp]),
    "afi_colsum_accum": (_i, [_vp, _ll, _i, _ll, _f, _vp, _vp, _vp]),
    "afi_bce_logits_fwd_bwd": (_i, [_vp, _ll, _f, _f, _vp, _f, _vp, _vp]),
    "afi_l1_fwd_bwd": (_i, [View, View, _i, _i, _i, _i, _i, _i, _f, _vp, _f, _vp, _vp]),
    "afi_sgd_momentum_step": (_i, [_vp, _i, _ll, _f, _f, _f, _vp]),
    "afi_scale_inplace": (_i, [_vp, _ll, _f, _vp]),
    "afi_nchw_to_nhwc": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "afi_nhwc_to_nchw": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "afi_profile_enable": (_i, [_i]),
    "afi_profile_num_kinds": (_i, []),
    "afi_profile_kind_name": (C.c_char_p, [_i]),
    "afi_profile_get": (_i, [_i, C.POINTER(C.c_double)]),
    "afi_profile_dump": (_i, [C.c_char_p]),
    "afi_debug_wk6_convT_images": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, C.POINTER(C.c_longlong), _vp]),
    "afi_debug_wgrad_sk_plan": (_i, [C.POINTER(C.c_longlong), C.POINTER(C.c_int), _i, _i, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
}

_lib = None


def load():
    """Load libafigan_hip.so once; raise AfiError (never fall back) when it is missing or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AfiError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
                       "There is no CPU fallback for the AFI-GAN hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.afi_abi_version() != ABI_VERSION:
        raise AfiError(f"ABI mismatch: library {lib.afi_abi_version()} != binding {ABI_VERSION}")
    # which binary is this?  The library carries the digest of the sources it was compiled from; the tree next to it says what it should be.
    # (An A/B library given by AFI_LIB_PATH is somebody's deliberate other build: reported by build_id(), not refused.)
    if not os.environ.get("AFI_LIB_PATH"):
        want, have = tree_digest(), lib.afi_build_id().decode()
        if want is not None and have != want:
            raise AfiError(f"{LIB_PATH} was built from other sources than this tree (library {have[:16]}, tree {want[:16]}): "
                           "rebuild it with `python -c 'import __graft_entry__ as g; g.build()'`")
    _lib = lib
    return lib


def tree_digest():
    """The digest afi_build_id() must return for the sources of this tree (__graft_entry__.source_digest), None when they are not there."""
    import glob
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    hdr = os.path.join(os.path.dirname(_HERE), "include", "afigan_hip.h")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + [f for f in glob.glob(os.path.join(csrc, "*.h")) if os.path.basename(f) != "afi_build_id.h"])
    if not files or not os.path.exists(hdr):
        return None
    h = hashlib.sha256()
    for f in files + [hdr]:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def build_id():
    return load().afi_build_id().decode()


def check(status: int, what: str = ""):
    if status != 0:
        msg = load().afi_status_string(status).decode()
        raise AfiError(f"{what or 'afigan_hip call'} failed: status {status} ({msg})")


# Entry points whose first argument is the caller-owned afi_ctx_t (include/afigan_hip.h): `call` passes the ACTIVE context of the calling
# thread -- the one made current by `use_ctx`, else the default context of the current GPU -- so call sites read like the header minus
# its first argument.  Contexts hold every piece of library state (op scratch, weight-transform cache, weight-gradient accumulator,
# side stream); nothing in the library is process-global.
CTX_FIRST = frozenset(n for n, (_, a) in SIGNATURES.items() if n.startswith(("afi_conv", "afi_generator_fwd", "afi_generator_bwd",
                                                                                 "afi_discriminator_fwd", "afi_discriminator_bwd"))
                      and not n.endswith(("_ws_floats", "_ws_layout", "pack_weight", "unpack_wgrad")))


DTYPES = {"fp32": 0, "bf16": 1, "f16x3": 2, "bf16x3": 3, "bf16x6": 6}             # AFI_DTYPE_* of include/afigan_hip.h
OPTIONS = {"winograd": 0, "winograd_f4_backward": 1, "winograd_f4_forward": 2, "bn_stats_fp64": 3, "d_winograd_min_pixels": 4,
           "g_winograd_min_pixels": 5, "g_smallmap_max_pixels": 6, "g_grouped_wgrad_max_pixels": 7, "g_batch_growth_grads": 8,
           "g_smallmap6_max_pixels": 9, "g_rdb_chain": 10, "d_fold_bn_apply": 11, "deterministic": 12, "f16_presplit": 13, "f16_nt256_min_tiles": 14, "f16_local_sums": 15, "d_fuse_tail": 16, "d_fuse_bwd_sums": 17}      # AFI_OPT_*


class Ctx:
    """afi_ctx_t owned from Python: created on the current GPU, destroyed with the object.  Buffers registered with it are kept alive here.
    `dtype`: arithmetic of the Winograd-domain GEMMs run under this context ("fp32", "bf16x6", "bf16x3", "bf16"; None = the library's
    default; see afi_ctx_set_compute_dtype)."""

    def __init__(self, dtype=None):
        h = C.c_void_p()
        check(load().afi_ctx_create(C.byref(h)), "afi_ctx_create")
        self.handle = h
        self.device = torch.cuda.current_device()
        self.bufs = {}                # name -> tensor registered with the context
        self.keep = None              # list collecting temporaries that must outlive an open weight-transform-cache block
        self.dtype = {v: k for k, v in DTYPES.items()}[load().afi_ctx_get_compute_dtype(h)]      # the library's default
        if dtype is not None and dtype != self.dtype:
            self.set_dtype(dtype)

    def set_dtype(self, dtype):
        if dtype not in DTYPES:
            raise AfiError(f"compute dtype must be one of {sorted(DTYPES)}, got {dtype!r}")
        with _dtype_lock:                                  # the default context is shared by the threads of a device (autograd worker included)
            check(load().afi_ctx_set_compute_dtype(self.handle, DTYPES[dtype]), "afi_ctx_set_compute_dtype")
            self.dtype = dtype

    def set_option(self, name, value):
        """Algorithm option of this context (OPTIONS / AFI_OPT_* of include/afigan_hip.h); the library reads no environment variable."""
        check(load().afi_ctx_set_option(self.handle, OPTIONS[name], int(value)), f"afi_ctx_set_option({name})")

    def get_option(self, name):
        return int(load().afi_ctx_get_option(self.handle, OPTIONS[name]))

    def __del__(self):
        try:
            if self.handle:
                _lib.afi_ctx_wino_wgrad_discard(self.handle)
                _lib.afi_ctx_destroy(self.handle)
                self.handle = None
        except Exception:             # interpreter shutdown
            pass


import functools  # noqa: E402
import threading  # noqa: E402
_tls = threading.local()
_default_ctx = {}
_default_lock = threading.Lock()
_dtype_lock = threading.RLock()       # serialises arithmetic switches of a (possibly shared) context: library call + the mirror attribute together
_observers = []                       # test hooks: fn(name, Ctx) called before every context-taking entry point (empty in production)


def current_ctx() -> "Ctx":
    """The context the calls of this thread go to: the innermost ``use_ctx`` block, else the default context of the current GPU.
    The default is per DEVICE (not per thread): PyTorch runs the backward of a custom Function on its autograd worker thread, and a
    per-thread default would hand that thread a second context -- another arithmetic setting, another 400 MB op scratch."""
    stack = getattr(_tls, "stack", None)
    if stack:
        return stack[-1]
    dev = torch.cuda.current_device()
    cx = _default_ctx.get(dev)
    if cx is None:
        with _default_lock:
            cx = _default_ctx.get(dev)
            if cx is None:
                cx = _default_ctx[dev] = Ctx()
    return cx


class compute_dtype:
    """``with compute_dtype("bf16x3"):`` -- the module-level ops (Generator / FPN / PAFPN / BiFPN autograd, afigan_amd.ops) run
    their matrix products in that arithmetic inside the block; restored on exit.  A forward run inside the block records the setting, and
    its backward runs in the same arithmetic wherever and on whichever thread ``.backward()`` is called (see ``ctx_forward``).
    Engines with their own context (Stage1Step, Stage2Adversarial) take ``dtype=`` instead."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        self.cx = current_ctx()
        self.prev = self.cx.dtype
        self.cx.set_dtype(self.dtype)
        return self.cx

    def __exit__(self, *exc):
        self.cx.set_dtype(self.prev)
        return False


class use_ctx:
    """``with use_ctx(cx):`` -- the calls of this thread inside the block run on context `cx` (an engine's own state).  With `dtype`,
    the context's arithmetic is switched for the block too (what a backward does to reproduce its forward's setting)."""

    def __init__(self, cx: "Ctx", dtype=None):
        self.cx = cx
        self.dtype = dtype
        self.prev = None

    def __enter__(self):
        if not hasattr(_tls, "stack"):
            _tls.stack = []
        _tls.stack.append(self.cx)
        if self.dtype is not None and self.dtype != self.cx.dtype:
            self.prev = self.cx.dtype
            self.cx.set_dtype(self.dtype)
        return self.cx

    def __exit__(self, *exc):
        # the stack first: afi_ctx_set_compute_dtype is refused while weight-gradient sums are pending, and if restoring the arithmetic
        # raises, every later call of this thread must still go to the context it had before the block, not to a stale one
        _tls.stack.pop()
        if self.prev is not None:
            self.cx.set_dtype(self.prev)
        return False


def ctx_forward(fn):
    """Decorator for ``autograd.Function.forward``: remembers the active context and its arithmetic on the autograd ctx."""
    @functools.wraps(fn)
    def wrapper(ctx, *args, **kw):
        ctx.afi_cx = current_ctx()
        ctx.afi_dtype = ctx.afi_cx.dtype
        return fn(ctx, *args, **kw)
    return wrapper


def ctx_backward(fn):
    """Decorator for ``autograd.Function.backward``: runs it on the context (and in the arithmetic) its forward ran on.  PyTorch calls
    backward on its autograd worker thread, where no ``use_ctx`` / ``compute_dtype`` block of the calling thread is visible."""
    @functools.wraps(fn)
    def wrapper(ctx, *grads):
        with use_ctx(ctx.afi_cx, ctx.afi_dtype):
            return fn(ctx, *grads)
    return wrapper


def call(name: str, *args):
    """Call an int-status entry point and raise on failure (context-taking entry points get the active context prepended)."""
    if name in CTX_FIRST:
        cx = current_ctx()
        for ob in _observers:
            ob(name, cx)
        args = (cx.handle,) + args
    check(getattr(load(), name)(*args), name)
