"""``GeneralizedRCNN_AFExtractor`` (afigan/modeling/meta_arch/rcnn_extractor.py:21-147): the stage-2 detector, a GeneralizedRCNN that runs on
``image_x0.5`` and ALSO hands back its FPN features, which the stage-2 loop trains against the guide network's.

The backbone is the AFI pyramid of this package (FPN_AFIGAN / PAFPN_AFIGAN through BACKBONE_REGISTRY); the proposal generator and the
ROI heads are detectron2 components outside this package's scope: ``from_config`` builds them with detectron2 when it is importable,
and the constructor takes any callables with their contracts otherwise (tests use small stand-ins)."""
import torch
import torch.nn as nn

from .rcnn_only import pad_to_batch
from .registry import BACKBONE_REGISTRY, Registry

try:                                                     # the reference registers into detectron2's META_ARCH_REGISTRY (:21)
    from detectron2.modeling import META_ARCH_REGISTRY
except Exception:
    META_ARCH_REGISTRY = Registry("META_ARCH")


class _Images:
    """What detectron2's ImageList gives the heads: the padded batch and the un-padded sizes."""

    def __init__(self, tensor, image_sizes):
        self.tensor, self.image_sizes = tensor, image_sizes

    def __len__(self):
        return len(self.image_sizes)


@META_ARCH_REGISTRY.register()
class GeneralizedRCNN_AFExtractor(nn.Module):
    def __init__(self, cfg=None, *, backbone=None, proposal_generator=None, roi_heads=None, pixel_mean=None, pixel_std=None,
                 device=None, input_format="BGR"):
        super().__init__()
        if cfg is not None and backbone is None:
            built = self._build_from_config(cfg)
            backbone, proposal_generator, roi_heads = built
            pixel_mean, pixel_std, device, input_format = cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD, cfg.MODEL.DEVICE, cfg.INPUT.FORMAT
        self.device = torch.device(device if device is not None else "cuda")
        self.backbone, self.proposal_generator, self.roi_heads = backbone, proposal_generator, roi_heads
        self.input_format = input_format
        assert len(pixel_mean) == len(pixel_std)
        n = len(pixel_mean)
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean, dtype=torch.float32).view(n, 1, 1), persistent=False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std, dtype=torch.float32).view(n, 1, 1), persistent=False)
        self.to(self.device)

    @staticmethod
    def _build_from_config(cfg):
        """rcnn_extractor.py:27-29,134-147: backbone from BACKBONE_REGISTRY, RPN / ROI heads from detectron2."""
        from .fpn_sr import ShapeSpec
        backbone = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN), stride=None))
        try:
            from detectron2.modeling.proposal_generator import build_proposal_generator
            from detectron2.modeling.roi_heads import build_roi_heads
        except Exception as e:
            from ._lib import AfiError
            raise AfiError(f"building the RPN / ROI heads from a config needs detectron2 ({type(e).__name__}: {e}); "
                           "pass proposal_generator= and roi_heads= instead")
        return backbone, build_proposal_generator(cfg, backbone.output_shape()), build_roi_heads(cfg, backbone.output_shape())

    def preprocess_image(self, batched_inputs):
        """rcnn_extractor.py:120-127: the detector sees the HALF-size image of the dual-scale mapper."""
        images = [(x["image_x0.5"].to(self.device).float() - self.pixel_mean) / self.pixel_std for x in batched_inputs]
        sizes = [tuple(t.shape[-2:]) for t in images]
        return _Images(pad_to_batch(images, self.backbone.size_divisibility), sizes)

    def forward(self, batched_inputs, img_dict_name="image"):
        if not self.training:
            return self.inference(batched_inputs)
        images = self.preprocess_image(batched_inputs)
        if "instances" in batched_inputs[0]:                                           # :45-50 (the x0.5 annotations)
            gt_instances = [x["instances_x0.5"].to(self.device) if hasattr(x["instances_x0.5"], "to") else x["instances_x0.5"] for x in batched_inputs]
        elif "targets" in batched_inputs[0]:
            gt_instances = [x["targets"] for x in batched_inputs]
        else:
            gt_instances = None
        features = self.backbone(images.tensor)                                        # :53
        processed_results = [{"features": features}]                                   # :55-56
        if self.proposal_generator:
            proposals, proposal_losses = self.proposal_generator(images, features, gt_instances)    # :58-59
        else:
            assert "proposals" in batched_inputs[0]
            proposals = [x["proposals"] for x in batched_inputs]
            proposal_losses = {}
        _, detector_losses = self.roi_heads(images, features, proposals, gt_instances)           # :65
        losses = {}
        losses.update(detector_losses)
        losses.update(proposal_losses)
        return losses, processed_results                                               # :67-70

    def inference(self, batched_inputs, detected_instances=None, do_postprocess=True):
        """rcnn_extractor.py:72-118: features -> proposals -> ROI-head predictions, then (do_postprocess) each image's instances rescaled
        to the input's "height" / "width" and wrapped as ``{"instances": r}`` -- the contract evaluators are written against."""
        assert not self.training
        images = self.preprocess_image(batched_inputs)
        features = self.backbone(images.tensor)
        if detected_instances is None:
            if self.proposal_generator:
                proposals, _ = self.proposal_generator(images, features, None)
            else:
                assert "proposals" in batched_inputs[0]
                proposals = [x["proposals"].to(self.device) if hasattr(x["proposals"], "to") else x["proposals"] for x in batched_inputs]
            results, _ = self.roi_heads(images, features, proposals, None)
        else:
            detected_instances = [x.to(self.device) if hasattr(x, "to") else x for x in detected_instances]        # :101
            results = self.roi_heads.forward_with_given_boxes(features, detected_instances)
        if do_postprocess:
            return self._postprocess(results, batched_inputs, images.image_sizes)                                   # :106-107
        return results

    @staticmethod
    def _postprocess(instances, batched_inputs, image_sizes):
        """rcnn_extractor.py:129-143: ``detector_postprocess(results, height, width)`` per image.  detectron2's own function when it is
        importable; otherwise its box part (scale by output / network size, clip, drop empty boxes: detectron2 v0.1.1
        modeling/postprocessing.py) on anything that carries ``image_size`` and ``pred_boxes`` -- and a loud error for instance fields this
        package cannot rescale without detectron2 (masks, keypoints) rather than a silently wrong scale."""
        try:
            from detectron2.modeling.postprocessing import detector_postprocess
        except Exception:
            detector_postprocess = _detector_postprocess_boxes
        out = []
        for results_per_image, input_per_image, image_size in zip(instances, batched_inputs, image_sizes):
            height = input_per_image.get("height", image_size[0])
            width = input_per_image.get("width", image_size[1])
            out.append({"instances": detector_postprocess(results_per_image, height, width)})
        return out


def _detector_postprocess_boxes(results, output_height, output_width):
    import copy
    from ._lib import AfiError
    for f in ("pred_masks", "pred_keypoints"):
        if getattr(results, f, None) is not None:
            raise AfiError(f"rescaling `{f}` to the input size needs detectron2's detector_postprocess (not importable here); "
                           "call inference(..., do_postprocess=False) for raw ROI-head results")
    h, w = results.image_size
    sx, sy = output_width / w, output_height / h
    r = copy.copy(results)
    boxes = getattr(results, "pred_boxes", None)
    if boxes is None:
        boxes = getattr(results, "proposal_boxes", None)
        name = "proposal_boxes"
    else:
        name = "pred_boxes"
    r.image_size = (output_height, output_width)
    if boxes is None:
        return r
    t = (boxes.tensor if hasattr(boxes, "tensor") else boxes).clone()
    t[:, 0::2] *= sx
    t[:, 1::2] *= sy
    t[:, 0::2].clamp_(min=0, max=output_width)
    t[:, 1::2].clamp_(min=0, max=output_height)
    keep = ((t[:, 2] - t[:, 0]) > 0) & ((t[:, 3] - t[:, 1]) > 0)
    if hasattr(boxes, "tensor"):
        nb = copy.copy(boxes)
        nb.tensor = t[keep]
        setattr(r, name, nb)
    else:
        setattr(r, name, t[keep])
    for f in ("scores", "pred_classes", "objectness_logits"):
        v = getattr(results, f, None)
        if v is not None and hasattr(v, "__getitem__") and len(v) == len(keep):
            setattr(r, f, v[keep])
    return r
