"""CPU-side checks of the C-ABI boundary: the library loads without a GPU and exports every symbol include/afigan_hip.h
declares; the binding table covers the header; workspace queries (host-only functions) answer."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    import __graft_entry__ as ge
    ge.build(verbose=False)


def test_library_exports_every_declared_symbol():
    _build()
    from afigan_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "afigan_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(afi_[A-Za-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in afigan_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))


def test_host_only_queries():
    _build()
    from afigan_amd import _lib
    lib = _lib.load()
    assert lib.afi_abi_version() == _lib.ABI_VERSION
    assert lib.afi_status_string(0) == b"ok" and b"workspace" in lib.afi_status_string(4)
    P, C, G, R = 850, 256, 32, 3
    # saved activations of one generator call: packed conv-transpose weight, the RDB dense buffers, t, a7 and the 2x map
    saved = 36 * C * C + R * P * (C + 4 * G) + 2 * P * C + 4 * P * C
    got = lib.afi_generator_fwd_ws_floats(C, G, R, 1, 25, 34)
    assert saved < got <= 8 * saved                             # + split-K slabs and the transient Winograd buffers
    assert lib.afi_generator_fwd_ws_floats(C, G, R, 2, 25, 34) > got          # grows with the batch
    small = lib.afi_generator_fwd_ws_floats(16, 4, 3, 1, 5, 7)               # no Winograd / big slabs for tiny shapes
    assert small < 64 * 1024
    assert lib.afi_conv3x3_wino_ws_floats(1, 50, 68, 256, 256) > 16 * 896 * 512
    F = (ctypes.c_int * 4)(256, 512, 1024, 1024)
    assert lib.afi_discriminator_fwd_ws_floats(F, 1, 50, 68) > 3400 * (2 * 512 + 4 * 1024)


def test_context_api_rejects_bad_arguments_without_a_gpu():
    """afi_ctx_* are host functions: NULL handling is checked here; creation needs a current device and is covered by the GPU tests."""
    _build()
    from afigan_amd import _lib
    lib = _lib.load()
    assert lib.afi_ctx_create(None) == 1                                   # AFI_ERR_BAD_ARG
    assert lib.afi_ctx_destroy(None) == 0                                  # destroying nothing is fine
    for fn in (lib.afi_ctx_wino_weight_cache_invalidate, lib.afi_ctx_wino_wgrad_discard):
        assert fn(None) == 1
    assert lib.afi_ctx_set_op_scratch(None, None, 0) == 1 and lib.afi_ctx_set_wino_weight_cache(None, None, 0) == 1
    assert lib.afi_ctx_set_wino_wgrad_accum(None, None, 0) == 1 and lib.afi_ctx_wino_wgrad_flush(None, None) == 1


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from afigan_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.load()
    except _lib.AfiError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("missing library must raise")
