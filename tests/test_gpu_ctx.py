"""The caller-owned context (include/afigan_hip.h: afi_ctx_t) follows a computation onto PyTorch's autograd thread.

PyTorch runs the backward of a custom autograd.Function on its device worker thread, where no ``use_ctx`` / ``compute_dtype`` block of the
calling thread is visible.  Every Function of the package records the active context (and its arithmetic) in forward and runs its
backward under it (afigan_amd/_lib.py: ctx_forward / ctx_backward); these tests observe the context handle and the library-side
arithmetic setting at the moment of each C-ABI call.  ("What's weak" 2 / ADVICE r2 item 1.)"""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import afigan_oracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def amd():
    import afigan_amd
    assert torch.cuda.is_available()
    return afigan_amd


class _Observe:
    """Records (entry point, thread id, context handle, arithmetic the LIBRARY reports for that handle) for every context-taking call."""

    def __init__(self, lib_mod, names):
        self.lib_mod, self.names, self.seen = lib_mod, names, []

    def __call__(self, name, cx):
        if name in self.names:
            self.seen.append((name, threading.get_ident(), cx.handle.value, self.lib_mod.load().afi_ctx_get_compute_dtype(cx.handle)))

    def __enter__(self):
        self.lib_mod._observers.append(self)
        return self

    def __exit__(self, *exc):
        self.lib_mod._observers.remove(self)
        return False


def _gen(amd, Cc=128):
    G = amd.Generator(in_channels=Cc, n_residual_dense_blocks=3).cuda()
    G.load_state_dict(orc.closed_form_generator_params(Cc, 3, 32))
    return G


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3", "bf16"])
@pytest.mark.parametrize("backward_inside_block", [True, False])
def test_backward_runs_on_the_forward_context_and_arithmetic(amd, dtype, backward_inside_block):
    from afigan_amd import _lib
    G = _gen(amd)
    x = torch.randn((1, 128, 32, 64), generator=torch.Generator().manual_seed(1)).cuda().requires_grad_(True)   # 2048 pixels: Winograd path
    default = _lib.current_ctx().dtype
    with _Observe(_lib, ("afi_generator_fwd", "afi_generator_bwd")) as ob:
        with amd.compute_dtype(dtype):
            out = G(x)
            if backward_inside_block:
                out.sum().backward()
        if not backward_inside_block:
            assert _lib.current_ctx().dtype == default
            out.sum().backward()
    assert [s[0] for s in ob.seen] == ["afi_generator_fwd", "afi_generator_bwd"]
    (_, t_f, h_f, d_f), (_, t_b, h_b, d_b) = ob.seen
    assert h_f == h_b, "backward ran on another context than its forward"
    assert d_f == d_b == _lib.DTYPES[dtype], (d_f, d_b)
    assert _lib.current_ctx().dtype == default                       # and the setting is restored afterwards
    # informational: PyTorch's CUDA backward does run on a worker thread (the reason the context has to be carried explicitly)
    print("forward thread", t_f, "backward thread", t_b, "main", threading.get_ident())


def test_bf16_backward_differs_from_fp32(amd):
    """The backward really changes arithmetic with the setting: after ONE fp32 forward, the weight gradients of a backward run under bf16
    differ from an fp32 backward's by far more than fp32's own rounding, and those of bf16x6 (the exact split) do not.  (The INPUT
    gradient is dominated by the bilinear skip, which no setting touches; the weight gradients come out of the GEMMs alone.)"""
    G = _gen(amd)
    x0 = torch.randn((1, 128, 32, 64), generator=torch.Generator().manual_seed(2))
    R = torch.randn((1, 128, 64, 128), generator=torch.Generator().manual_seed(3)).cuda()
    grads = {}
    for dt in ("fp32", "bf16x6", "bf16"):
        for q in G.parameters():
            q.grad = None
        x = x0.cuda().requires_grad_(True)
        with amd.compute_dtype("fp32"):                              # the same forward for all three: only the backward differs
            out = G(x)
        out.grad_fn.afi_dtype = dt                                   # what ctx_forward recorded; the backward must obey it
        (out * R).sum().backward()
        grads[dt] = {k: q.grad.detach().double().cpu().contiguous() for k, q in G.named_parameters() if q.dim() == 4}

    def worst(a, b):
        return max(((a[k] - b[k]).norm() / b[k].norm()).item() for k in b)
    assert worst(grads["bf16x6"], grads["fp32"]) < 2e-5
    assert worst(grads["bf16"], grads["fp32"]) > 1e-4


def test_use_ctx_covers_a_backward_inside_it(amd):
    """A caller-made context (an engine's) made current around forward AND backward serves both; the default context sees neither."""
    from afigan_amd import _lib
    cx = _lib.Ctx("bf16x3")
    G = _gen(amd)
    x = torch.randn((1, 128, 32, 64), generator=torch.Generator().manual_seed(4)).cuda().requires_grad_(True)
    with _Observe(_lib, ("afi_generator_fwd", "afi_generator_bwd")) as ob, _lib.use_ctx(cx):
        G(x).sum().backward()
    assert {s[2] for s in ob.seen} == {cx.handle.value} and {s[3] for s in ob.seen} == {3}
    assert _lib.current_ctx() is not cx


def test_one_default_context_per_device(amd):
    """The default context is per device, not per thread: a worker thread's module-level call lands on the same context (no second op scratch)."""
    from afigan_amd import _lib
    main_cx = _lib.current_ctx()
    got = []
    th = threading.Thread(target=lambda: got.append(_lib.current_ctx()))
    th.start()
    th.join()
    assert got[0] is main_cx


def test_context_options_replace_the_environment(amd):
    """afi_ctx_set_option: the algorithm switches are per-context state, changeable between calls (the library reads no environment
    variable).  Winograd off / on for one discriminator call on one context: same logits to fp32 rounding, and the defaults come back."""
    from afigan_amd import _lib
    cx = _lib.Ctx()
    assert cx.get_option("winograd") == 1 and cx.get_option("winograd_f4_forward") == 12 and cx.get_option("d_winograd_min_pixels") == 1024
    assert cx.get_option("g_winograd_min_pixels") == 2048 and cx.get_option("g_grouped_wgrad_max_pixels") == 3000 and cx.get_option("bn_stats_fp64") == 1
    lib = _lib.load()
    assert lib.afi_ctx_set_option(cx.handle, 99, 1) == 1 and lib.afi_ctx_set_option(cx.handle, 0, -1) == 1 and lib.afi_ctx_get_option(None, 0) == 1
    D = amd.Discriminator().cuda().train()
    D.load_state_dict(orc.closed_form_discriminator_params())
    x = torch.randn((1, 256, 40, 48), generator=torch.Generator().manual_seed(3)).cuda()
    outs = {}
    with torch.no_grad(), _lib.use_ctx(cx):
        for wino in (1, 0, 1):
            cx.set_option("winograd", wino)
            with _Observe(_lib, ("afi_discriminator_fwd",)) as ob:
                outs.setdefault(wino, []).append(D(x).clone())
            assert ob.seen[0][2] == cx.handle.value
    assert torch.equal(outs[1][0], outs[1][1])                                  # deterministic, and the option really toggles back
    d = ((outs[0][0] - outs[1][0]).abs().max() / outs[1][0].abs().max()).item()
    assert 0 < d < 1e-4, d                                                       # two algorithms: different rounding, same function
