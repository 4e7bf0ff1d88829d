"""Frozen guide feature extractor for the stage-1 harness: a random-init ResNet-50-FPN in eval mode that turns an image
batch into the P2..P6 pyramid, the role ``RCNN_FPN_only`` plays in the reference (afigan/modeling/meta_arch/rcnn_only.py:
34-44).  It is OUT OF SCOPE as a kernel target (stock R-50-FPN); it exists so that bench.py's step
contains the same work as stage1_trainer.py:320-321 (two guide forwards per iteration).

To keep a fresh GPU box from spending minutes in MIOpen's just-in-time kernel builds, the guide avoids MIOpen entirely:
1x1 convs run on this package's pixel GEMM (afi_conv1x1_fwd: bias, the residual / top-down addend and the ReLU in its epilogue; a stride-2
conv reads every second pixel in place), the 7x7 stem is unfold + GEMM, every 3x3
conv (stride 1 under detectron2's STRIDE_IN_1X1=True) runs on this package's own fp32-MFMA conv kernels with a fused
bias + ReLU epilogue (the >= 128-channel ones in the inference Winograd form, afi_conv3x3_wino_infer), and the frozen BatchNorms are folded into weights / biases at construction.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def _kaiming(shape, fan_in, gain=1.0):
    """He init on fan_in (variance-preserving through ReLU): with the residual branches damped (gain 0.25 on their last conv) the
    random-init pyramid stays O(1), like a trained guide's features -- the stage-1 step then trains stably for as many iterations
    as a bench run asks for (an unnormalised harness with features ~1e5 drives the interpolator to NaN within ~25 iterations)."""
    return torch.randn(shape) * (gain * math.sqrt(2.0 / fan_in))


class _Conv1x1(nn.Module):
    def __init__(self, cin, cout, stride=1, relu=False, gain=1.0):
        super().__init__()
        self.stride, self.relu = stride, relu
        self.register_buffer("w", _kaiming((cin, cout), cin, gain).t().contiguous())     # [Cout, Cin] (frozen-BN scale folded in)
        self.register_buffer("b", torch.zeros(cout))

    def forward(self, x, add=None, relu=None):                          # x: [N,C,H,W] pixel-major; out = act(conv(x) + b + add)
        if self.stride != 1:
            x = x[:, :, ::self.stride, ::self.stride]                   # (a view: the kernel walks the strides)
        return ops.conv1x1_fwd(x, self.w, self.b, add=add, act=2 if (self.relu if relu is None else relu) else 0)


class _Conv3x3(nn.Module):
    def __init__(self, cin, cout, relu=False):
        super().__init__()
        self.relu = relu
        self.register_buffer("w", _kaiming((cout, 3, 3, cin), cin * 9).permute(0, 3, 1, 2))    # memory [Cout][3][3][Cin]
        self.register_buffer("b", torch.zeros(cout))

    def forward(self, x):
        N, C, H, W = x.shape
        if C >= 128 and self.w.shape[0] >= 128 and N * H * W >= 1024:      # frozen inference net: the Winograd form, F(4x4) on big maps
            return ops.conv3x3_wino_infer(x, self.w, self.b, act=2 if self.relu else 0)
        return ops.conv3x3_fwd(x, self.w, self.b, lrelu=2 if self.relu else 0)


class _Bottleneck(nn.Module):
    def __init__(self, cin, mid, cout, stride):
        super().__init__()
        self.c1 = _Conv1x1(cin, mid, stride, relu=True)                 # detectron2 STRIDE_IN_1X1 = True
        self.c2 = _Conv3x3(mid, mid, relu=True)
        self.c3 = _Conv1x1(mid, cout, gain=0.25)
        self.short = _Conv1x1(cin, cout, stride) if (cin != cout or stride != 1) else None

    def forward(self, x):
        s = x if self.short is None else self.short(x)
        return self.c3(self.c2(self.c1(x)), add=s, relu=True)           # relu(conv + shortcut) in the GEMM's epilogue


class GuideR50FPN(nn.Module):
    """images [N,3,H,W] (0..255) -> {"p2".."p6"} of 256 channels, strides 4..64, input padded to a multiple of 32."""

    def __init__(self, out_channels=256):
        super().__init__()
        self.register_buffer("stem_w", _kaiming((3 * 49, 64), 3 * 49))
        self.register_buffer("stem_b", torch.zeros(64))
        cfg = [(64, 64, 256, 3, 1), (256, 128, 512, 4, 2), (512, 256, 1024, 6, 2), (1024, 512, 2048, 3, 2)]
        self.stages = nn.ModuleList()
        for cin, mid, cout, n, stride in cfg:
            blocks = [_Bottleneck(cin, mid, cout, stride)] + [_Bottleneck(cout, mid, cout, 1) for _ in range(n - 1)]
            self.stages.append(nn.Sequential(*blocks))
        self.lateral = nn.ModuleList([_Conv1x1(c, out_channels, gain=0.1) for c in (256, 512, 1024, 2048)])   # pyramid std ~1
        self.output = nn.ModuleList([_Conv3x3(out_channels, out_channels) for _ in range(4)])
        self.register_buffer("pixel_mean", torch.tensor([103.53, 116.28, 123.675]).view(1, 3, 1, 1))
        self.eval()

    @torch.no_grad()
    def forward(self, images):
        x = (images - self.pixel_mean) * (1.0 / 58.0)          # harness only: unit-scale input (a trained guide absorbs the 0..255 range)
        H, W = x.shape[-2:]
        ph, pw = (32 - H % 32) % 32, (32 - W % 32) % 32
        if ph or pw:
            x = F.pad(x, (0, pw, 0, ph))                         # ImageList.from_tensors(size_divisibility=32)
        N, _, H, W = x.shape
        cols = F.unfold(x, kernel_size=7, padding=3, stride=2)   # [N, 147, H/2*W/2]: the 7x7/2 stem as a GEMM
        y = F.relu_(torch.baddbmm(self.stem_b.view(1, 1, -1), cols.transpose(1, 2), self.stem_w.unsqueeze(0).expand(N, -1, -1)))
        x = y.view(N, H // 2, W // 2, 64).permute(0, 3, 1, 2)
        x = F.max_pool2d(x, 3, 2, 1).contiguous(memory_format=torch.channels_last)
        feats = []
        for st in self.stages:
            x = st(x)
            feats.append(x)
        prev = self.lateral[3](feats[3])
        outs = [self.output[3](prev)]
        for i in (2, 1, 0):
            up = F.interpolate(prev, scale_factor=2, mode="nearest")
            prev = self.lateral[i](feats[i], add=up)            # lateral + top-down in the GEMM's epilogue
            outs.insert(0, self.output[i](prev))
        outs.append(outs[-1][:, :, ::2, ::2].contiguous(memory_format=torch.channels_last))   # LastLevelMaxPool(k=1,s=2) -> p6
        return {f"p{i + 2}": o for i, o in enumerate(outs)}
