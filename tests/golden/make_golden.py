#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference modules.

Run in the build container only (needs /root/reference; it does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference's two hot-path files (afigan/modeling/feat_interpol/generator_rdb.py and
feature_patch_discriminator.py) are loaded BY PATH with importlib.  They import detectron2 / fvcore
names at module top; detectron2 v0.1.1 and fvcore are not installed here and there is no network,
so exactly those names are provided as stand-in modules in sys.modules:
  detectron2.layers.Conv2d          -> nn.Conv2d subclass applying optional `norm` then `activation`
  detectron2.layers.ConvTranspose2d -> nn.ConvTranspose2d
  detectron2.layers.get_norm("BN")  -> nn.BatchNorm2d
  detectron2.layers.ShapeSpec, detectron2.utils.registry.Registry -> inert placeholders
  fvcore.nn.weight_init.c2_msra_fill -> kaiming_normal_(fan_out, relu) + zero bias
For the FPN / PAFPN fixtures afigan/modeling/backbone/{fpn_sr,pafpn_sr}.py are loaded the same way; their extra module-top
imports get inert stand-ins too (detectron2 Backbone = nn.Module, BACKBONE_REGISTRY.register = identity,
build_resnet_backbone / build_resnest_backbone = unused stubs, c2_xavier_fill = kaiming_uniform_(a=1) + zero bias,
get_norm("") = None) and the already-loaded reference generator is registered under its package name.
Only arrays (inputs / expected outputs) are written; no reference source or bytecode is copied.
The stage-1 trainer (stage1_trainer.py) is not importable (deep detectron2 imports), so its
run_step lines 336-433 are replayed here against the imported Generator / Discriminator.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import afigan_oracle as orc  # noqa: E402  (closed-form weight generators shared by both sides)


def install_shims():
    d2 = types.ModuleType("detectron2")
    layers = types.ModuleType("detectron2.layers")
    utils = types.ModuleType("detectron2.utils")
    registry = types.ModuleType("detectron2.utils.registry")
    fv = types.ModuleType("fvcore")
    fvnn = types.ModuleType("fvcore.nn")
    wi = types.ModuleType("fvcore.nn.weight_init")

    class Conv2d(nn.Conv2d):
        def __init__(self, *a, **kw):
            norm = kw.pop("norm", None)
            act = kw.pop("activation", None)
            super().__init__(*a, **kw)
            self.norm = norm
            self.activation = act

        def forward(self, x):
            x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
            if self.norm is not None:
                x = self.norm(x)
            if self.activation is not None:
                x = self.activation(x)
            return x

    def get_norm(norm, ch):
        assert norm == "BN"
        return nn.BatchNorm2d(ch)

    class ShapeSpec:  # placeholder
        pass

    class Registry:  # placeholder
        def __init__(self, name):
            self.name = name

        def register(self, obj=None):
            return obj

    def c2_msra_fill(m):
        nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)

    def get_norm_any(norm, ch):
        if norm == "":
            return None
        assert norm == "BN"
        return nn.BatchNorm2d(ch)

    def c2_xavier_fill(m):
        nn.init.kaiming_uniform_(m.weight, a=1)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)

    # extra stand-ins needed only by afigan/modeling/backbone/fpn_sr.py (its module-top imports)
    modeling = types.ModuleType("detectron2.modeling")
    backbone = types.ModuleType("detectron2.modeling.backbone")
    bbuild = types.ModuleType("detectron2.modeling.backbone.build")
    bresnet = types.ModuleType("detectron2.modeling.backbone.resnet")

    class Backbone(nn.Module):
        pass

    class _Reg:
        def register(self, obj=None):
            return obj if obj is not None else (lambda f: f)

    backbone.Backbone = Backbone
    bbuild.BACKBONE_REGISTRY = _Reg()
    bresnet.build_resnet_backbone = lambda *a, **k: None
    d2.modeling, modeling.backbone, backbone.build, backbone.resnet = modeling, backbone, bbuild, bresnet
    wi.c2_xavier_fill = c2_xavier_fill
    layers.Conv2d = Conv2d
    layers.ConvTranspose2d = nn.ConvTranspose2d
    layers.get_norm = get_norm_any
    layers.ShapeSpec = ShapeSpec
    registry.Registry = Registry
    wi.c2_msra_fill = c2_msra_fill
    d2.layers, d2.utils, utils.registry = layers, utils, registry
    fv.nn, fvnn.weight_init = fvnn, wi
    for name, mod in [("detectron2", d2), ("detectron2.layers", layers), ("detectron2.utils", utils),
                      ("detectron2.utils.registry", registry), ("fvcore", fv), ("fvcore.nn", fvnn),
                      ("fvcore.nn.weight_init", wi), ("detectron2.modeling", modeling), ("detectron2.modeling.backbone", backbone),
                      ("detectron2.modeling.backbone.build", bbuild), ("detectron2.modeling.backbone.resnet", bresnet)]:
        sys.modules[name] = mod


def checkpoint_remap_cases():
    """Key sets (name -> shape) for the stage hand-off fixture; shared with tests/test_host_logic.py.
    case -> (model keys, checkpoint keys, reference method name)."""
    g = {k: (2,) + tuple(min(d, 3) for d in shp[1:]) for k, shp in orc.generator_param_shapes(256, 3, 32).items()}   # tiny stand-in shapes
    det = {"backbone.srf_module." + k: v for k, v in g.items()}
    det.update({"backbone.fpn_lateral2.weight": (2, 3), "backbone.fpn_output2.weight": (2, 3), "backbone.bottom_up.stem.conv1.weight": (2, 3),
                "roi_heads.box_head.fc1.weight": (4, 3), "proposal_generator.rpn_head.conv.weight": (2, 2)})
    stage1 = dict(g)
    stage1["Generators.0.4.0.bias"] = (5,)                          # shape mismatch: must be skipped
    stage1["iteration_marker"] = (1,)                               # matches nothing
    stage2 = {k: v for k, v in det.items()}
    stage2["backbone.bottom_up.stem.conv1.weight"] = (2, 3)         # same name and shape, but not an srf_module tensor: dropped
    stage2["backbone.srf_module.Generators.0.2.0.weight"] = (7, 7)  # shape mismatch inside srf_module
    return {"af_extractor": (det, stage1, "align_and_update_state_dicts_AFExtractor"),
            "target_detector": (det, stage2, "align_and_update_state_dicts_TargetDetector")}


def bifpn_closed_form(name, like):
    """Closed-form value of one BiFPN_AFIGAN state-dict entry (shared with tests/test_oracle_golden.py).  Scales are chosen for
    a per-node gain near 1 so that seven stacked BiFPN layers keep O(1) activations: conv weights with std 1/sqrt(fan_in),
    small biases, BatchNorm affine 1 +- 0.17 / shift +- 0.17, running_var in [0.5, 1.5], raw fusion weights in [0.3, 0.6]."""
    shape = tuple(like.shape)
    if name.endswith("num_batches_tracked"):
        return torch.tensor(3, dtype=torch.long)
    if name.endswith("running_var"):
        return 1.0 + orc.closed_form_tensor(name, shape, 0.5 / 3 ** 0.5)
    if name.endswith("running_mean"):
        return orc.closed_form_tensor(name, shape, 0.1)
    if "_w1" in name or "_w2" in name:                                      # the raw fusion weights (bifpn_sr.py:535-563)
        return 0.45 + orc.closed_form_tensor(name, shape, 0.15 / 3 ** 0.5)
    if len(shape) == 1:                                                     # BatchNorm affine and conv biases
        is_norm_w = name.endswith("weight")
        return (1.0 if is_norm_w else 0.0) + orc.closed_form_tensor(name, shape, 0.1)
    fan_in = shape[1] * shape[2] * shape[3]
    return orc.closed_form_tensor(name, shape, 1.0 / fan_in ** 0.5)


def load_ref(relpath, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sd_np(module):
    return {k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def tensor_digest(t: torch.Tensor, nsample=64):
    """sum, l2 norm, abs-max and a strided sample (fixed stride from numel) of a tensor."""
    f = t.detach().reshape(-1).double()
    n = f.numel()
    stride = max(1, n // nsample)
    return np.array([f.sum().item(), f.norm().item(), f.abs().max().item()], dtype=np.float64), \
        f[::stride][:nsample].float().numpy()


def grads_digest(named_grads):
    out = {}
    for k, g in named_grads.items():
        d, s = tensor_digest(g)
        out["gd/" + k] = d
        out["gs/" + k] = s
    return out


def main():
    torch.set_num_threads(8)
    install_shims()
    G_rdb = load_ref("afigan/modeling/feat_interpol/generator_rdb.py", "ref_generator_rdb")
    D_mod = load_ref("afigan/modeling/feat_interpol/feature_patch_discriminator.py", "ref_discriminator")

    # ---------------- FPN_AFIGAN (fpn_sr.py): the top-down merge around G, SURVEY 8f row 1 ----------------
    # fpn_sr.py also imports two modules of the reference's own package; register the already-loaded generator under its
    # package name and an inert stand-in for the ResNeSt builder (not on this path) so the file loads by path.
    for name in ("afigan", "afigan.modeling", "afigan.modeling.backbone", "afigan.modeling.feat_interpol"):
        sys.modules.setdefault(name, types.ModuleType(name))
    resnest = types.ModuleType("afigan.modeling.backbone.resnest")
    resnest.build_resnest_backbone = lambda *a, **k: None
    sys.modules["afigan.modeling.backbone.resnest"] = resnest
    sys.modules["afigan.modeling.feat_interpol.generator_rdb"] = G_rdb
    sys.modules["afigan.modeling.feat_interpol"].generator_rdb = G_rdb
    fpn_mod = load_ref("afigan/modeling/backbone/fpn_sr.py", "ref_fpn_sr")
    Backbone = sys.modules["detectron2.modeling.backbone"].Backbone

    class ShapeSpec_:
        def __init__(self, channels, stride):
            self.channels, self.stride = channels, stride

    class BottomUp(Backbone):
        def output_shape(self):
            return {f"res{i + 2}": ShapeSpec_(c, s) for i, (c, s) in enumerate(zip([8, 12, 16, 20], [4, 8, 16, 32]))}

        def forward(self, feats):
            return feats

    class Cfg:
        class MODEL:
            AFI_FREEZE = False

    for fuse in ("sum", "avg"):
        fpn = fpn_mod.FPN_AFIGAN(BottomUp(), ["res2", "res3", "res4", "res5"], 256, norm="", top_block=fpn_mod.LastLevelMaxPool(),
                                 fuse_type=fuse, cfg=Cfg)
        sd = {}
        for k, v in fpn.state_dict().items():
            if k.startswith("srf_module."):
                continue
            sd[k] = orc.closed_form_tensor(k, v.shape, 0.05 if k.endswith("bias") else (6.0 / (v.shape[1] * v.shape[2] * v.shape[3])) ** 0.5 / 3 ** 0.5)
        sd.update({"srf_module." + k: v for k, v in orc.closed_form_generator_params().items()})
        fpn.load_state_dict(sd, strict=True)
        gen = torch.Generator().manual_seed(31)
        feats = {f"res{i + 2}": torch.randn((1, c, 2 * 2 ** (3 - i), 3 * 2 ** (3 - i)), generator=gen).requires_grad_(True)
                 for i, c in enumerate([8, 12, 16, 20])}
        out = fpn(feats)
        sum((o * o).mean() for o in out.values()).backward()
        fx = {"seed": np.array([31])}
        for k, o in out.items():
            fx["out/" + k] = o.detach().numpy() if k != "p2" else o.detach()[:, ::4].numpy()
        for k, f in feats.items():
            fx["dfeat/" + k] = f.grad.numpy()
        fx.update(grads_digest({k: p.grad for k, p in fpn.named_parameters()}))
        np.savez_compressed(os.path.join(HERE, f"fpn_{fuse}.npz"), **fx)
        print("fpn", fuse, {k: tuple(v.shape) for k, v in out.items()})

    # ---------------- PAFPN_AFIGAN (pafpn_sr.py): top-down AFI merge + bottom-up path aggregation, SURVEY 8f row 1 ----------------
    pafpn_mod = load_ref("afigan/modeling/backbone/pafpn_sr.py", "ref_pafpn_sr")
    for fuse in ("sum", "avg"):
        pafpn = pafpn_mod.PAFPN_AFIGAN(BottomUp(), ["res2", "res3", "res4", "res5"], 256, norm="",
                                       top_block=pafpn_mod.LastLevelMaxPool(), fuse_type=fuse, cfg=Cfg)
        sd = {}
        for k, v in pafpn.state_dict().items():
            if k.startswith("srf_module."):
                continue
            sd[k] = orc.closed_form_tensor(k, v.shape, 0.05 if k.endswith("bias") else (6.0 / (v.shape[1] * v.shape[2] * v.shape[3])) ** 0.5 / 3 ** 0.5)
        sd.update({"srf_module." + k: v for k, v in orc.closed_form_generator_params().items()})
        pafpn.load_state_dict(sd, strict=True)
        gen = torch.Generator().manual_seed(33)
        feats = {f"res{i + 2}": torch.randn((1, c, 2 * 2 ** (3 - i), 3 * 2 ** (3 - i)), generator=gen).requires_grad_(True)
                 for i, c in enumerate([8, 12, 16, 20])}
        out = pafpn(feats)
        sum((o * o).mean() for o in out.values()).backward()
        fx = {"seed": np.array([33])}
        for k, o in out.items():
            fx["out/" + k] = o.detach().numpy() if k != "p2" else o.detach()[:, ::4].numpy()
        for k, f in feats.items():
            fx["dfeat/" + k] = f.grad.numpy()
        fx.update(grads_digest({k: p.grad for k, p in pafpn.named_parameters()}))
        np.savez_compressed(os.path.join(HERE, f"pafpn_{fuse}.npz"), **fx)
        print("pafpn", fuse, {k: tuple(v.shape) for k, v in out.items()}, sorted(k for k in sd if not k.startswith("srf_module."))[:6])

    # ---------------- BiFPN_AFIGAN (bifpn_sr.py), inference: 7 hard-wired BiFPN layers, 28 interpolator calls, SURVEY 8f row 4 ----------------
    # The reference ships BiFPN only in an inference config (configs/inference/...BiFPN_ST.yaml), so the fixture is an eval-mode
    # forward.  Its own helper package afigan/modeling/bifpn_layers/{wrappers,activations}.py is loaded by path too (it needs
    # detectron2.layers.batch_norm.get_norm -> the BN stand-in; eval-mode SyncBN == BN), the Swin builder gets an inert stub.
    bn_mod = types.ModuleType("detectron2.layers.batch_norm")
    bn_mod.get_norm = sys.modules["detectron2.layers"].get_norm
    sys.modules["detectron2.layers.batch_norm"] = bn_mod
    swin = types.ModuleType("afigan.modeling.backbone.swin_transformer")
    swin.build_swint_backbone = lambda *a, **k: None
    sys.modules["afigan.modeling.backbone.swin_transformer"] = swin
    wr = load_ref("afigan/modeling/bifpn_layers/wrappers.py", "afigan.modeling.bifpn_layers.wrappers")
    ac = load_ref("afigan/modeling/bifpn_layers/activations.py", "afigan.modeling.bifpn_layers.activations")
    bl = types.ModuleType("afigan.modeling.bifpn_layers")
    for k in ("Conv2d", "SeparableConv2d", "MaxPool2d"):
        setattr(bl, k, getattr(wr, k))
    bl.MemoryEfficientSwish, bl.Swish = ac.MemoryEfficientSwish, ac.Swish
    sys.modules["afigan.modeling.bifpn_layers"] = bl
    bifpn_mod = load_ref("afigan/modeling/backbone/bifpn_sr.py", "ref_bifpn_sr")

    class BottomUp3(Backbone):
        _out_feature_strides = {"stage3": 8, "stage4": 16, "stage5": 32}
        _out_feature_channels = {"stage3": 8, "stage4": 12, "stage5": 16}

        def forward(self, feats):
            return feats

    bifpn = bifpn_mod.BiFPN_AFIGAN(BottomUp3(), ["stage3", "stage4", "stage5"], 256, 7, norm="BN",
                                   top_block=bifpn_mod.LastLevelP6P7(16, 256, "BN"), fuse_type="sum", cfg=Cfg)
    sd = {}
    for k, v in bifpn.state_dict().items():
        if k.startswith("srf_module."):
            continue
        sd[k] = bifpn_closed_form(k, v)
    sd.update({"srf_module." + k: v for k, v in orc.closed_form_generator_params().items()})
    bifpn.load_state_dict(sd, strict=True)
    bifpn.eval()
    gen = torch.Generator().manual_seed(35)
    feats = {f"stage{i + 3}": torch.randn((1, c, 16 // 2 ** i, 32 // 2 ** i), generator=gen) for i, c in enumerate([8, 12, 16])}
    with torch.no_grad():
        out = bifpn(feats)
    fx = {"seed": np.array([35])}
    for k, o in out.items():
        fx["out/" + k] = o.numpy()
    fx["n_params"] = np.array([len(sd)])
    fx["state_dict_contract"] = np.array([f"{k}:{list(v.shape)}" for k, v in bifpn.state_dict().items() if not k.startswith("srf_module.")])
    np.savez_compressed(os.path.join(HERE, "bifpn_eval.npz"), **fx)
    print("bifpn", {k: tuple(v.shape) for k, v in out.items()}, len(sd), "tensors")

    # BiFPN_AFIGAN in TRAINING mode (batch-statistics norms, autograd through all seven layers): outputs, input gradients, parameter
    # gradient digests and the running buffers after the step.  norm="BN": nn.SyncBatchNorm has no CPU forward, and within one
    # process the two compute the same function.
    bifpn.load_state_dict(sd, strict=True)
    bifpn.train()
    gen = torch.Generator().manual_seed(36)
    feats = {f"stage{i + 3}": torch.randn((2, c, 32 // 2 ** i, 48 // 2 ** i), generator=gen).requires_grad_(True) for i, c in enumerate([8, 12, 16])}
    out = bifpn(feats)
    R = {k: torch.randn(o.shape, generator=torch.Generator().manual_seed(200 + i)) for i, (k, o) in enumerate(out.items())}
    sum((o * R[k]).sum() for k, o in out.items()).backward()
    fx = {"seed": np.array([36]), "state_dict_contract": fx["state_dict_contract"]}
    for k, o in out.items():
        fx["out/" + k] = o.detach().numpy() if k != "p3" else o.detach()[:, ::4].numpy()
    for k, f in feats.items():
        fx["dfeat/" + k] = f.grad.numpy()
    fx.update(grads_digest({k: p.grad for k, p in bifpn.named_parameters() if p.grad is not None}))
    for k, v in bifpn.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            d, smp = tensor_digest(v)
            fx["bd/" + k], fx["bs/" + k] = d, smp
        elif k.endswith("num_batches_tracked"):
            assert int(v) == 4, (k, int(v))
    np.savez_compressed(os.path.join(HERE, "bifpn_train.npz"), **fx)
    print("bifpn train", {k: tuple(v.shape) for k, v in out.items()}, sum(p.grad is not None for p in bifpn.parameters()), "gradients")

    # ---------------- G-small: full tensors, reference default init ----------------
    for tag, shape in (("a", (2, 16, 5, 7)), ("b", (1, 16, 7, 11))):
        torch.manual_seed(1234)
        G = G_rdb.Generator(in_channels=16, n_residual_dense_blocks=3, growth_rate=4)
        with torch.no_grad():   # reference zero-inits biases; give them values so bias paths are pinned
            for k, v in G.state_dict().items():
                if k.endswith("bias"):
                    v.copy_(orc.closed_form_tensor(k, v.shape, 0.05))
        x = torch.randn(shape, generator=torch.Generator().manual_seed(7)).requires_grad_(True)
        R = torch.randn((shape[0], 16, 2 * shape[2], 2 * shape[3]), generator=torch.Generator().manual_seed(8))
        out = G(x)
        (out * R).sum().backward()
        fx = {"x": x.detach().numpy(), "R": R.numpy(), "out": out.detach().numpy(), "dx": x.grad.numpy()}
        for k, v in sd_np(G).items():
            fx["w/" + k] = v
        for k, p in G.named_parameters():
            fx["g/" + k] = p.grad.numpy()
        np.savez_compressed(os.path.join(HERE, f"g_small_{tag}.npz"), **fx)
        print("g_small", tag, out.shape)

    # ---------------- G-full (config 1 pin), closed-form weights ----------------
    G = G_rdb.Generator(n_residual_dense_blocks=3)
    gp = orc.closed_form_generator_params()
    G.load_state_dict(gp, strict=True)
    x = torch.randn((1, 256, 25, 34), generator=torch.Generator().manual_seed(0)).requires_grad_(True)
    out = G(x)
    out.sum().backward()
    fx = {
        "x_seed": np.array([0]), "x_shape": np.array([1, 256, 25, 34]),
        "out_slice": out.detach()[0, ::16, ::5, ::7].numpy(),          # [16,10,10]
        "out_chan_sum": out.detach().double().sum(dim=(0, 2, 3)).numpy(),
        "out_absmax": np.array([out.detach().abs().max().item()]),
        "out_row": out.detach()[0, :, 17, :].numpy(),                  # [256,68] one full row
        "dx_slice": x.grad[0, ::16, ::5, ::7].numpy(),
    }
    fx.update(grads_digest({k: p.grad for k, p in G.named_parameters()}))
    dd, ds = tensor_digest(x.grad)
    fx["gd/x"], fx["gs/x"] = dd, ds
    np.savez_compressed(os.path.join(HERE, "g_full_cfg1.npz"), **fx)
    print("g_full", out.shape)

    # ---------------- bilinear index map / ramps (bit-exact) ----------------
    fx = {}
    for L in (1, 2, 5, 7, 25):
        ramp = torch.arange(L, dtype=torch.float32).view(1, 1, L, 1).expand(1, 1, L, 3).contiguous()
        up = F.interpolate(ramp, scale_factor=2, mode="bilinear")
        fx[f"ramp_h_{L}"] = up[0, 0, :, 0].numpy()
        ramp = torch.arange(L, dtype=torch.float32).view(1, 1, 1, L).expand(1, 1, 3, L).contiguous()
        up = F.interpolate(ramp, scale_factor=2, mode="bilinear")
        fx[f"ramp_w_{L}"] = up[0, 0, 0, :].numpy()
    xr = torch.randn((2, 3, 5, 7), generator=torch.Generator().manual_seed(3))
    fx["rand_in"] = xr.numpy()
    fx["rand_out"] = F.interpolate(xr, scale_factor=2, mode="bilinear").numpy()
    np.savez_compressed(os.path.join(HERE, "bilinear.npz"), **fx)

    # ---------------- D, closed-form weights ----------------
    dp0 = orc.closed_form_discriminator_params()
    for tag, shape in (("a", (2, 256, 13, 21)), ("b", (1, 256, 7, 11))):
        D = D_mod.Discriminator()
        D.load_state_dict(dp0, strict=True)
        D.train()
        x = torch.randn(shape, generator=torch.Generator().manual_seed(11)).requires_grad_(True)
        acts = {}
        hooks = []
        for n in range(3):
            conv = D.Discriminators[0][n][0]
            # conv.norm input = conv output: capture batch statistics
            hooks.append(conv.norm.register_forward_hook(
                lambda m, i, o, n=n: acts.__setitem__(n, i[0].detach())))
        logits = D.Discriminators[0](x)
        for h in hooks:
            h.remove()
        Rl = torch.randn(logits.shape, generator=torch.Generator().manual_seed(12))
        (logits * Rl).sum().backward()
        fx = {"x_seed": np.array([11]), "x_shape": np.array(shape), "R": Rl.numpy(),
              "logits": logits.detach().numpy(), "dx_slice": x.grad[0, ::16].numpy()}
        for n in range(3):
            c = acts[n]
            fx[f"bn{n}_mean"] = c.mean(dim=(0, 2, 3)).numpy()
            fx[f"bn{n}_var"] = c.var(dim=(0, 2, 3), unbiased=False).numpy()
        for k, v in D.state_dict().items():
            if "running" in k or "num_batches" in k:
                fx["buf/" + k] = v.numpy()
        fx.update(grads_digest({k: p.grad for k, p in D.named_parameters()}))
        dd, ds = tensor_digest(x.grad)
        fx["gd/x"], fx["gs/x"] = dd, ds
        np.savez_compressed(os.path.join(HERE, f"d_{tag}.npz"), **fx)
        print("d", tag, logits.shape)

    # ---------------- stage-1 step replay: stage1_trainer.py:336-433 ----------------
    G = G_rdb.Generator(n_residual_dense_blocks=3)
    G.load_state_dict(orc.closed_form_generator_params(), strict=True)
    D = D_mod.Discriminator()
    D.load_state_dict(orc.closed_form_discriminator_params(), strict=True)
    G.train()
    D.train()
    gen = torch.Generator().manual_seed(21)
    lr_features = [torch.randn((2, 256, 13, 21), generator=gen), torch.randn((2, 256, 7, 11), generator=gen)]
    hr_features = [torch.randn((2, 256, 25, 42), generator=gen), torch.randn((2, 256, 13, 21), generator=gen)]
    crit = nn.BCEWithLogitsLoss()
    base_lr, wd, mom = 1e-3, 1e-4, 0.9

    def reshape_stage1(t, size):  # replay of _reshape_stage1 (:437-443)
        if size[2] != t.size()[2] or size[3] != t.size()[3]:
            _W = size[2] if size[2] < t.size()[2] else t.size()[2]
            _H = size[3] if size[3] < t.size()[3] else t.size()[3]
            return t[:, :, 0:_W, 0:_H]
        return t

    def make_opt(mod):  # detectron2 v0.1.1 build_optimizer: per-param groups, wd 0 for norm params
        groups = []
        for m in mod.modules():
            for name, p in m.named_parameters(recurse=False):
                w = 0.0 if isinstance(m, nn.BatchNorm2d) else wd
                groups.append({"params": [p], "lr": base_lr, "weight_decay": w})
        return torch.optim.SGD(groups, base_lr, momentum=mom)

    G_opt, D_opt = make_opt(G.Generators[0]), make_opt(D.Discriminators[0])
    fx = {"seed": np.array([21]), "lr": np.array([base_lr]), "wd": np.array([wd]), "mom": np.array([mom])}
    d_loss = {}
    for p_lv, (lr_f, hr_f) in enumerate(zip(lr_features, hr_features), 2):
        tr = G(lr_f).detach()
        tr = reshape_stage1(tr, hr_f.size())
        hr = reshape_stage1(hr_f, tr.size())
        logit_real = D.Discriminators[0](hr)
        logit_fake = D.Discriminators[0](tr)
        real, fake = torch.ones(logit_real.size()), torch.zeros(logit_fake.size())
        d_loss[f"d_loss_p{p_lv}"] = crit(logit_real, real) + crit(logit_fake, fake)
        fx[f"crop_p{p_lv}"] = np.array(list(tr.shape))
    d_losses = sum(d_loss.values())
    D_opt.zero_grad()
    d_losses.backward()
    fx.update({k: np.array([v.item()]) for k, v in d_loss.items()})
    dg = grads_digest({k: p.grad for k, p in D.named_parameters()})
    fx.update({"D" + k: v for k, v in dg.items()})
    D_opt.step()
    fx.update({"Dw_after/" + k: tensor_digest(p)[0] for k, p in D.named_parameters()})

    g_loss = {}
    for p_lv, (lr_f, hr_f) in enumerate(zip(lr_features, hr_features), 2):
        tr = G(lr_f)
        tr = reshape_stage1(tr, hr_f.size())
        hr = reshape_stage1(hr_f, tr.size())
        logit_fake = D.Discriminators[0](tr).detach()
        logit_real = D.Discriminators[0](hr)
        real = torch.ones(logit_real.size())
        adv = crit(logit_fake, real)
        content = F.l1_loss(tr, hr)
        assert not adv.requires_grad  # Q1
        g_loss[f"g_loss_p{p_lv}"] = adv * 1e-3 + content
        fx[f"adv_loss_p{p_lv}"] = np.array([adv.item()])
        fx[f"content_loss_p{p_lv}"] = np.array([content.item()])
    g_losses = sum(g_loss.values())
    G_opt.zero_grad()
    g_losses.backward()
    fx.update({k: np.array([v.item()]) for k, v in g_loss.items()})
    gg = grads_digest({k: p.grad for k, p in G.named_parameters()})
    fx.update({"G" + k: v for k, v in gg.items()})
    G_opt.step()
    fx.update({"Gw_after/" + k: tensor_digest(p)[0] for k, p in G.named_parameters()})
    for k, v in D.state_dict().items():
        if "running" in k or "num_batches" in k:
            fx["Dbuf_after/" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "stage1_step.npz"), **fx)
    print("stage1 step", {k: v.item() for k, v in d_loss.items()}, {k: v.item() for k, v in g_loss.items()})

    # ---------------- stage-2 adversarial terms replay: stage2_trainer.py:299-364 ----------------
    D = D_mod.Discriminator()
    D.load_state_dict(orc.closed_form_discriminator_params(), strict=True)
    D.train()
    gen = torch.Generator().manual_seed(41)
    guide = [torch.randn((2, 256, 26, 42), generator=gen), torch.randn((2, 256, 13, 21), generator=gen)]     # hr_feature_ p-levels
    fpn = [torch.randn((2, 256, 13, 21), generator=gen).requires_grad_(True),                                # upsampled_feature_ p-levels
           torch.randn((2, 256, 7, 11), generator=gen).requires_grad_(True)]
    D_opt = make_opt(D.Discriminators[0])

    def reshape_feature(t, size):      # stage2_trainer.py _reshape_feature == stage-1's crop
        return reshape_stage1(t, size)

    hr_features = [F.interpolate(g, scale_factor=0.5) for g in guide]                                        # :302
    fx = {"seed": np.array([41]), "lr": np.array([base_lr]), "wd": np.array([wd]), "mom": np.array([mom])}
    d_loss = {}
    for p_lv, (hr_f, up_f) in enumerate(zip(hr_features, fpn), 2):
        hr = reshape_feature(hr_f, up_f.size())
        up = reshape_feature(up_f, hr.size())
        logit_real = D.Discriminators[0](hr)
        logit_fake = D.Discriminators[0](up.detach())
        d_loss[f"d_loss_p{p_lv}"] = crit(logit_real, torch.ones(logit_real.size())) + crit(logit_fake, torch.zeros(logit_fake.size()))
        fx[f"crop_p{p_lv}"] = np.array(list(up.shape))
    D_opt.zero_grad()
    sum(d_loss.values()).backward()
    fx.update({k: np.array([v.item()]) for k, v in d_loss.items()})
    fx.update({"D" + k: v for k, v in grads_digest({k: p.grad for k, p in D.named_parameters()}).items()})
    D_opt.step()
    fx.update({"Dw_after/" + k: tensor_digest(p)[0] for k, p in D.named_parameters()})
    g_loss = {}
    for p_lv, (hr_f, up_f) in enumerate(zip(hr_features, fpn), 2):
        hr = reshape_feature(hr_f, up_f.size())
        up = reshape_feature(up_f, hr.size())
        logit_fake = D.Discriminators[0](up).detach()
        logit_real = D.Discriminators[0](hr)
        adv = crit(logit_fake, torch.ones(logit_real.size()))
        content = F.l1_loss(up, hr)
        g_loss[f"g_loss_p{p_lv}"] = adv * 1e-3 + content
        fx[f"adv_loss_p{p_lv}"] = np.array([adv.item()])
        fx[f"content_loss_p{p_lv}"] = np.array([content.item()])
    sum(g_loss.values()).backward()
    fx.update({k: np.array([v.item()]) for k, v in g_loss.items()})
    for i, f in enumerate(fpn):
        fx[f"dfpn_{i}"] = f.grad.numpy()[:, ::8]
    for k, v in D.state_dict().items():
        if "running" in k or "num_batches" in k:
            fx["Dbuf_after/" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "stage2_adv.npz"), **fx)
    print("stage2 adv", {k: v.item() for k, v in d_loss.items()}, {k: v.item() for k, v in g_loss.items()})

    # ---------------- stage hand-off key remapping: afigan/engine/checkpoint.py:64-258 ----------------
    # The module's top imports (fvcore Checkpointer / PathManager, detectron2 comm / c2_model_loading) get inert stand-ins;
    # the two align_and_update_state_dicts_* methods only use `self` to reach convert_AFI_names / remain_only_AFI_names.
    import json
    for name, attrs in (("fvcore.common", {}), ("fvcore.common.checkpoint", {"Checkpointer": type("Checkpointer", (), {})}),
                        ("fvcore.common.file_io", {"PathManager": object}), ("detectron2.utils.comm", {"is_main_process": lambda: True}),
                        ("detectron2.checkpoint", {}), ("detectron2.checkpoint.c2_model_loading", {"align_and_update_state_dicts": None})):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
    ck_mod = load_ref("afigan/engine/checkpoint.py", "ref_checkpoint")
    ck = object.__new__(ck_mod.AF_DetectionCheckpointer)
    cases = {}
    for case, (model_shapes, ckpt_shapes, method) in checkpoint_remap_cases().items():
        model_sd = {k: torch.full(shape, -1.0) for k, shape in model_shapes.items()}
        ckpt_sd = {k: torch.full(shape, float(i)) for i, (k, shape) in enumerate(sorted(ckpt_shapes.items()))}
        getattr(ck, method)(model_sd, ckpt_sd)
        order = sorted(ckpt_shapes)
        cases[case] = {k: (order[int(v.reshape(-1)[0].item())] if v.reshape(-1)[0].item() >= 0 else None) for k, v in model_sd.items()}
    with open(os.path.join(HERE, "checkpoint_remap.json"), "w") as f:
        json.dump(cases, f, indent=1, sort_keys=True)
    print("checkpoint remap", {k: sum(v is not None for v in c.values()) for k, c in cases.items()})


if __name__ == "__main__":
    main()
