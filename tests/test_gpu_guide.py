"""The bench harness's frozen guide network (afigan_amd/guide.py: the role RCNN_FPN_only plays in stage1_trainer.py:320-321) against the same
network written with torch.nn.functional in fp64 on the CPU: 1x1 convs on afi_conv1x1_fwd (stride-2 views, residual / top-down addend and
ReLU in the epilogue), 3x3 convs on the library's conv kernels."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref_forward(g, images):
    d = torch.float64
    c1 = lambda m, x: F.conv2d(x, m.w.detach().cpu().to(d)[:, :, None, None], m.b.detach().cpu().to(d), stride=m.stride)
    c3 = lambda m, x: F.conv2d(x, m.w.detach().cpu().to(d), m.b.detach().cpu().to(d), padding=1)
    x = (images.to(d) - g.pixel_mean.cpu().to(d)) * (1.0 / 58.0)
    H, W = x.shape[-2:]
    ph, pw = (32 - H % 32) % 32, (32 - W % 32) % 32
    x = F.pad(x, (0, pw, 0, ph))
    w7 = g.stem_w.detach().cpu().to(d).t().reshape(64, 3, 7, 7)
    x = F.relu(F.conv2d(x, w7, g.stem_b.cpu().to(d), stride=2, padding=3))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for st in g.stages:
        for b in st:
            y = F.relu(c1(b.c1, x))
            y = F.relu(c3(b.c2, y))
            s = x if b.short is None else c1(b.short, x)
            x = F.relu(c1(b.c3, y) + s)
        feats.append(x)
    prev = c1(g.lateral[3], feats[3])
    outs = [c3(g.output[3], prev)]
    for i in (2, 1, 0):
        prev = c1(g.lateral[i], feats[i]) + F.interpolate(prev, scale_factor=2, mode="nearest")
        outs.insert(0, c3(g.output[i], prev))
    outs.append(outs[-1][:, :, ::2, ::2])
    return outs


@pytest.mark.parametrize("shape", [(2, 3, 64, 96), (1, 3, 100, 70)])
def test_guide_network_vs_torch_functional(shape):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from afigan_amd.guide import GuideR50FPN
    torch.manual_seed(0)
    g = GuideR50FPN().cuda()
    img = torch.rand(shape, generator=torch.Generator().manual_seed(1)) * 255.0
    got = g(img.cuda())
    ref = _ref_forward(g, img)
    for i, r in enumerate(ref):
        o = got[f"p{i + 2}"].cpu().to(torch.float64)
        assert o.shape == r.shape, (i, o.shape, r.shape)
        assert (o - r).abs().max().item() <= 1e-3 * r.abs().max().item(), (i, (o - r).abs().max().item(), r.abs().max().item())
