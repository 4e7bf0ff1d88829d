"""The frozen guide network of stages 1 and 2: ``RCNN_FPN_only`` (afigan/modeling/meta_arch/rcnn_only.py:17-59).

A backbone-only feature extractor: ``forward(batched_inputs, img_dict_name) -> [{"features": {p2..p6}}]``.  The backbone comes from
``BACKBONE_REGISTRY.get(cfg.MODEL.GUIDE_BACKBONE.NAME)``; normalisation and the ``ImageList.from_tensors`` zero padding to the
backbone's ``size_divisibility`` (rcnn_only.py:36-39) are done here without detectron2."""
import torch
import torch.nn as nn

from .registry import BACKBONE_REGISTRY, GUIDE_ARCH_REGISTRY


def pad_to_batch(images, size_divisibility: int) -> torch.Tensor:
    """ImageList.from_tensors: zero-pad [C,H,W] tensors at the bottom / right to the batch maximum rounded up to size_divisibility."""
    hmax, wmax = max(t.shape[-2] for t in images), max(t.shape[-1] for t in images)
    if size_divisibility > 1:
        hmax = (hmax + size_divisibility - 1) // size_divisibility * size_divisibility
        wmax = (wmax + size_divisibility - 1) // size_divisibility * size_divisibility
    out = images[0].new_zeros((len(images), images[0].shape[0], hmax, wmax))
    for i, t in enumerate(images):
        out[i, :, :t.shape[-2], :t.shape[-1]].copy_(t)
    return out


@GUIDE_ARCH_REGISTRY.register()
class RCNN_FPN_only(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.device = torch.device(cfg.MODEL.DEVICE)
        self.backbone = self.build_backbone(cfg)
        self.input_format = cfg.INPUT.FORMAT
        assert len(cfg.MODEL.PIXEL_MEAN) == len(cfg.MODEL.PIXEL_STD)
        n = len(cfg.MODEL.PIXEL_MEAN)
        self.register_buffer("pixel_mean", torch.tensor(cfg.MODEL.PIXEL_MEAN, dtype=torch.float32).view(n, 1, 1), persistent=False)
        self.register_buffer("pixel_std", torch.tensor(cfg.MODEL.PIXEL_STD, dtype=torch.float32).view(n, 1, 1), persistent=False)
        self.to(self.device)

    def normalizer(self, x):
        return (x - self.pixel_mean) / self.pixel_std

    def forward(self, batched_inputs, img_dict_name="image"):
        images = [self.normalizer(x[img_dict_name].to(self.device).float()) for x in batched_inputs]
        batch = pad_to_batch(images, self.backbone.size_divisibility)
        return [{"features": self.backbone(batch)}]

    def build_backbone(self, cfg, input_shape=None):
        """rcnn_only.py:47-59: the backbone named by cfg.MODEL.GUIDE_BACKBONE.NAME."""
        if input_shape is None:
            from .fpn_sr import ShapeSpec
            input_shape = ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN), stride=None)
        return BACKBONE_REGISTRY.get(cfg.MODEL.GUIDE_BACKBONE.NAME)(cfg, input_shape)
