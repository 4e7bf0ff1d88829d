"""Discriminator gradient parity, split into what is continuous and what is not ("What's weak" #1).

A gradient that passes a LeakyReLU is DISCONTINUOUS in the sign of the pre-activation: an element whose BatchNorm output lies within
fp32 rounding of zero gets slope 1 in one fp32 evaluation and 0.2 in another.  Measured with tools/d_parity_probe.py on an MI355X:
  * with the LeakyReLU masks of the reference forward (its saved activations injected into the workspace) the whole HIP backward --
    BN backward, Winograd / direct data gradients, weight gradients -- agrees with fp64 autograd to 1-2e-6 max-norm (2e-5 at 8400 px);
  * ONE flipped mask element out of 10^6 moves dx by 8e-4 relative L2 / 4e-3 max-norm and a weight gradient by 1.3e-2 max-norm;
  * torch's own fp32 CPU evaluation of the reference flips as well (4 elements at 2x256x50x84: dx 4.6e-4 L2 / 4.1e-3 max-norm vs fp64),
    so no fp32 implementation can hold 1e-3 against another one on the full gradient; the library's flip count is held to the
    reference's own (statistics in fp64, as torch's CPU BatchNorm accumulates; the remaining difference is the conv rounding, 1.0-1.5e-6
    against 0.3-0.5e-6).
The bars below are named after what they bound."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from d_parity_util import DProbe, l2, rel  # noqa: E402

TOL_GIVEN_MASKS_MAXNORM = 1e-4     # backward with the reference's masks: measured 1-2e-6 (fixture sizes), 2e-5 at 8400 px (Winograd F(4x4) gradients)
TOL_FORWARD_MAXNORM = 1e-5         # saved conv outputs / activations vs fp64: measured <= 1.7e-6
TOL_STATS = 2e-6                   # batch mean / variance vs fp64: measured <= 4e-7 (fp64 accumulation)
# Mask flips and the gradients behind them are Poisson-small counts: one input says little (0..7 flips of 10^7 elements, one flip moves dx by
# ~8e-4 at these sizes).  The bars are therefore on SUMS over seeds, against the same sums of torch's own fp32 evaluation of the same inputs:
#   * small maps (direct kernels / below the Winograd thresholds): flips <= torch-CPU's + FLIP_SLACK, per case, as rounds 1-5 documented;
#   * the 8,400-pixel map (2x50x84, the smallest size where every Winograd path incl. the F(4x4) forwards runs), three seeds: measured
#     (profiles/r06/flip_counts_2x256x50x84.txt) torch-CPU 9 flips in all; winograd_f4_forward = 0: 17; = 8: 19; = 12 in the plain summation
#     order (round 5's default): 42; = 12 with f16_local_sums = 12 (the default): 21; = 1 with them: 36.
#     Bar: <= 2 x torch-CPU's + FLIP_SLACK (22 there) -- holds for the default, 0 and 8, fails for plain 12 and for 1.  (torch's CPU convs sum
#     in a blocked order that rounds less than any MFMA chain: "not more than torch-CPU's" is not reachable by an fp32 matrix-core kernel.)
FLIP_SLACK = 4
FLIP_FACTOR_WINOGRAD = 2
# own-forward gradient: at most this factor over the deviation of torch's fp32 ops ON THE GPU (MIOpen) from fp64 on the same inputs, dx and worst
# parameter gradient, means over the seeds (profiles/r06/dflip_p3_*_6seeds_local_sums.txt at 2x256x100x168: default 0.76x / 0.82x of torch's; = 12
# without the local sums 1.10x / 1.13x; = 1 with them 1.15x / 1.16x).
OWN_FORWARD_OVER_TORCH_FP32 = 1.0
OWN_FORWARD_OVER_TORCH_CPU = 3.0  # 2x50x84 through the C-ABI against torch-CPU fp32 (see the test)
TOL_OWN_FORWARD_L2_SMALL = 3e-3    # small maps (one flip ~ 1e-3 of a tensor there; no statistics possible): flat bar, as before

F4_SETTINGS = [None, 0]            # None = the library's default (afi_ctx_get_option(NULL, AFI_OPT_WINOGRAD_F4_FORWARD)); 0 = F(2x2) everywhere


@pytest.fixture(scope="module")
def amd():
    import afigan_amd
    assert torch.cuda.is_available()
    return afigan_amd


CASES = [(2, 13, 21, 1), (1, 7, 11, 11), (1, 25, 42, 3)]        # the two reference fixtures' shapes (d_a, d_b) and a P4-sized map
WINO_CASE, WINO_SEEDS = (2, 50, 84), (5, 6, 7)                  # every Winograd path of the discriminator runs from 8192 pixels on


def _options(f4):
    return None if f4 is None else {"winograd_f4_forward": f4}


def _check_forward(pr):
    assert rel(pr.logits, pr.r64["logits"]) < TOL_FORWARD_MAXNORM
    for n in range(3):
        c, y, mean, invstd = pr.saved(n)
        assert rel(c, pr.r64["c"][n].detach()) < TOL_FORWARD_MAXNORM, n
        assert rel(y, pr.r64["y"][n].detach()) < TOL_FORWARD_MAXNORM, n
        assert rel(mean, pr.r64["mean"][n]) < TOL_STATS, n
        var = 1.0 / invstd.double() ** 2 - 1e-5
        assert rel(var, pr.r64["var"][n]) < TOL_STATS, n


@pytest.mark.parametrize("N,H,W,seed", CASES)
def test_forward_statistics_and_mask_flips(amd, N, H, W, seed):
    pr = DProbe(amd, N, H, W, seed)
    _check_forward(pr)
    fg, fc = pr.mask_flips()
    assert sum(fg) <= sum(fc) + FLIP_SLACK, (fg, fc)


@pytest.mark.parametrize("f4", F4_SETTINGS)
def test_forward_statistics_and_mask_flips_on_the_winograd_forwards(amd, f4):
    """The flip-visible test at the size where the F(4x4) forward runs: the library's OWN forward, masks compared with
    the fp64 oracle's, summed over three seeds and held against torch-CPU fp32's own flips on the same inputs."""
    lib_flips, cpu_flips = [], []
    for seed in WINO_SEEDS:
        pr = DProbe(amd, *WINO_CASE, seed, options=_options(f4))
        _check_forward(pr)
        fg, fc = pr.mask_flips()
        lib_flips.append(fg)
        cpu_flips.append(fc)
        del pr
    tot_g, tot_c = sum(map(sum, lib_flips)), sum(map(sum, cpu_flips))
    print(f"winograd_f4_forward={f4}: flips {lib_flips} = {tot_g}; torch-CPU fp32 {cpu_flips} = {tot_c}")
    assert tot_g <= FLIP_FACTOR_WINOGRAD * tot_c + FLIP_SLACK, (lib_flips, cpu_flips)


def test_local_sums_round_the_conv_outputs_less(amd):
    """AFI_OPT_F16_LOCAL_SUMS (csrc/afi_gemm_f16.h: the three products of a k-step are summed in a fresh fragment and added to the accumulator by
    one fp32 addition, instead of three accumulating MFMAs): the same products in an order that rounds the large accumulator once per k-step.
    At 2x256x50x84 with F(4x4) forwards in blocks 1 and 2, the saved conv outputs of those blocks against the fp64 oracle: the relative-L2
    rounding error with the local sums is below the plain order's (measured ~0.6x), both inside the forward bar."""
    err = {}
    for ls in (0, 12):
        pr = DProbe(amd, 2, 50, 84, 5, options={"winograd_f4_forward": 12, "f16_local_sums": ls})
        _check_forward(pr)
        err[ls] = [l2(pr.saved(n)[0], pr.r64["c"][n].detach()) for n in range(3)]
        del pr
    print(f"conv-output rounding (relative L2 vs fp64) per block: plain order {err[0]}, local sums {err[12]}")
    assert err[12][1] < 0.85 * err[0][1] and err[12][2] < 0.85 * err[0][2], err


def test_default_forward_gradient_deviation_not_above_torch_fp32(amd):
    """the bar is RELATIVE to what torch's own fp32 ops do on the same inputs.  D forward + backward at P3 size
    (2x256x100x168) on the library's default options against an fp64 evaluation (torch ops, float64, on the GPU), three seeds: the mean
    relative-L2 deviation of dx and of the worst parameter gradient must not exceed torch fp32's (MIOpen) own.  A wider default fails this:
    every block on F(4x4) (1.15x / 1.16x) or the default's block set in the plain summation order (1.10x / 1.13x:
    profiles/r06/dflip_p3_2x256x100x168_6seeds_local_sums.txt)."""
    import torch.nn.functional as F

    def torch_grads(D, x, r, dt):
        sd = {k: v.detach().to(dt).requires_grad_(True) for k, v in D.named_parameters()}
        xx = x.detach().to(dt).requires_grad_(True)
        h = xx
        for n in range(3):
            p = f"Discriminators.0.{n}.0."
            h = F.conv2d(h, sd[p + "weight"], sd[p + "bias"], padding=1)
            h = F.batch_norm(h, None, None, sd[p + "norm.weight"], sd[p + "norm.bias"], training=True, eps=1e-5)
            h = F.leaky_relu(h, 0.2)
        h = F.conv2d(h, sd["Discriminators.0.3.0.weight"], sd["Discriminators.0.3.0.bias"], padding=1)
        (h * r.to(dt)).sum().backward()
        return {"dx": xx.grad.double(), **{n: sd[n].grad.double() for n in sd}}

    def dev(o, ref):
        live = [k for k in ref if ref[k].norm() > 1e-9 * ref[k].numel() ** 0.5]      # (conv biases ahead of a train-mode BatchNorm: identically zero gradient)
        return ((o["dx"] - ref["dx"]).norm() / ref["dx"].norm()).item(), max(((o[k] - ref[k]).norm() / ref[k].norm()).item() for k in live)

    torch.backends.cudnn.allow_tf32 = False
    lib, t32 = [], []
    for seed in (0, 1, 2):
        torch.manual_seed(seed)
        D = amd.Discriminator().cuda()
        x = torch.randn(2, 256, 100, 168, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
        r = torch.randn(2, 1, 100, 168, device="cuda")
        ref = torch_grads(D, x, r, torch.float64)
        t32.append(dev(torch_grads(D, x, r, torch.float32), ref))
        (D(x) * r).sum().backward()
        lib.append(dev({"dx": x.grad.double(), **{n: p.grad.double() for n, p in D.named_parameters()}}, ref))
        del ref, D, x
    mean = lambda v, i: sum(t[i] for t in v) / len(v)
    print(f"library default: dx {mean(lib, 0):.3e} worst {mean(lib, 1):.3e}; torch fp32: dx {mean(t32, 0):.3e} worst {mean(t32, 1):.3e}")
    assert mean(lib, 0) <= OWN_FORWARD_OVER_TORCH_FP32 * mean(t32, 0), (lib, t32)
    assert mean(lib, 1) <= OWN_FORWARD_OVER_TORCH_FP32 * mean(t32, 1), (lib, t32)


@pytest.mark.parametrize("N,H,W,seed", CASES + [(2, 50, 84, 5)])      # the last one runs the Winograd forward / F(4x4) backward
def test_backward_with_reference_masks_full_discriminator(amd, N, H, W, seed):
    """dx and every weight / BN gradient of the FULL discriminator to 1e-4 max-norm once the masks are the reference's."""
    pr = DProbe(amd, N, H, W, seed)
    pr.inject_fp64_forward()
    dx, grads = pr.backward()
    e = pr.errors(dx, grads)
    assert e["dx_max"] < TOL_GIVEN_MASKS_MAXNORM, e
    assert e["worst_max"][0] < TOL_GIVEN_MASKS_MAXNORM, e


@pytest.mark.parametrize("N,H,W,seed", CASES)
def test_backward_on_own_forward_within_flip_noise(amd, N, H, W, seed):
    pr = DProbe(amd, N, H, W, seed)
    e = pr.errors(*pr.backward())
    assert e["dx_l2"] < TOL_OWN_FORWARD_L2_SMALL and e["worst_l2"][0] < TOL_OWN_FORWARD_L2_SMALL, e


@pytest.mark.parametrize("f4", F4_SETTINGS)
def test_backward_on_own_winograd_forward_against_torch_cpu_fp32(amd, f4):
    """The same at 2x50x84 through the C-ABI (DProbe), three seeds: the library's own-forward gradients against the fp64 oracle, beside torch-CPU
    fp32's on the same inputs.  torch's CPU convs round less than any matrix-core chain (9 flips against 17-19 here), so the bar against THEM
    is a factor: mean dx / worst-tensor deviation <= 3x torch-CPU's (profiles/r06/flip_counts_2x256x50x84.txt: the default 2.1x / 2.7x, = 0
    1.5x / 2.6x, = 8 1.8x / 2.5x; = 12 in the plain summation order 3.9x / 4.8x fails).  The bar against torch's GPU ops is test_default_forward_gradient_deviation_not_above_torch_fp32."""
    lib, cpu = [], []
    for seed in WINO_SEEDS:
        pr = DProbe(amd, *WINO_CASE, seed, options=_options(f4))
        e = pr.errors(*pr.backward())
        c = pr.cpu_fp32_backward_errors()
        lib.append((e["dx_l2"], e["worst_l2"][0]))
        cpu.append((c["dx_l2"], c["worst_l2"][0]))
        del pr
    mean = lambda v, i: sum(t[i] for t in v) / len(v)
    print(f"winograd_f4_forward={f4}: dx {mean(lib, 0):.3e} worst {mean(lib, 1):.3e}; torch-CPU fp32: dx {mean(cpu, 0):.3e} worst {mean(cpu, 1):.3e}")
    assert mean(lib, 0) <= OWN_FORWARD_OVER_TORCH_CPU * mean(cpu, 0), (lib, cpu)
    assert mean(lib, 1) <= OWN_FORWARD_OVER_TORCH_CPU * mean(cpu, 1), (lib, cpu)


@pytest.mark.parametrize("N,H,W,seed", CASES + [(2, 50, 84, 5)])
def test_masks_recomputed_in_backward_match_the_forward(amd, N, H, W, seed):
    """afi_discriminator_bwd does not read the saved activations: it recomputes each LeakyReLU' decision from the saved conv output as
    sign(((c - mean) * invstd) * gamma + beta), every operation rounded to fp32 (include/afigan_hip.h: the precondition on gamma / beta).
    On the library's OWN, un-nudged forward the recomputed decisions must agree with the activations the forward stored -- the count of
    disagreeing masks is reported and must be zero -- and with the fp64 forward injected WITHOUT the ulp nudges of
    DProbe.inject_fp64_forward the number of elements that would land on the other side is what that helper repairs."""
    pr = DProbe(amd, N, H, W, seed)
    bc = lambda v: v.view(1, -1, 1, 1)
    disagree = []
    for n in range(3):
        pre = f"Discriminators.0.{n}.0.norm"
        c, y, mean, invstd = pr.saved(n)
        ga, be = pr.dp[pre + ".weight"].float().cuda(), pr.dp[pre + ".bias"].float().cuda()
        z = ((c - bc(mean)) * bc(invstd)) * bc(ga) + bc(be)
        disagree.append(int(((z > 0) != (y > 0)).sum()))
    assert disagree == [0, 0, 0], disagree
    nudged = pr.inject_fp64_forward()                      # (count of elements whose recomputed fp32 decision differed from fp64's before nudging)
    total = sum(pr.net.F[n + 1] for n in range(3)) * pr.P
    assert sum(nudged) <= max(8, total // 10**5), (nudged, total)
