"""Dual-scale data path on the device.

Host-side mirror of the reference's ``afigan.engine.dataset_mapper.DatasetMapper`` (dataset_mapper.py:24-193) for the part
that feeds the stage-1/2 trainers: one decoded image becomes ``image`` (ResizeShortestEdge + RandomFlip) and ``image_x0.5``
(the ORIGINAL image resized straight to ``int(new_h*0.5) x int(new_w*0.5)`` with the same flip decision,
transform_gen.py:514-559), and the boxes follow both transform lists (afigan_utils.py:140-170, 234-262, 328-356).

The pixels are produced by ``csrc/resample.hip`` (Pillow-exact antialiased bilinear, flip and CHW store fused into the second
pass); sizes, random draws and the box arithmetic (a handful of float64 operations per box) stay on the host, as in the
reference.  Same dict keys as the reference, including its ``width_x0.5`` / ``heigth_x0.5`` [sic] entries, which read
``shape[1]`` and ``shape[2]`` of the CHW tensor and therefore hold half the HEIGHT and half the WIDTH (dataset_mapper.py:121-122).

Out of scope (not on the path): file reading / JPEG decode (``dataset_dict["image"]`` must hold the decoded uint8
HWC array), RandomCrop, masks, keypoints, proposals, semantic segmentation.  Random draws come from ``numpy.random``'s global
state in the reference's order (size, flip, then the two discarded draws of the x0.5 list), so a seeded run picks the same
sizes and flips as the reference mapper.
"""
import copy
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib, ops

XYXY_ABS, XYWH_ABS = 0, 1          # detectron2.structures.BoxMode values


def shortest_edge_size(h, w, size, max_size):
    """ResizeShortestEdge.get_transform (transform_gen.py:198-217): the resized (h, w) for a drawn short-edge ``size``."""
    scale = size * 1.0 / min(h, w)
    if h < w:
        newh, neww = size, scale * w
    else:
        newh, neww = scale * h, size
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh = newh * scale
        neww = neww * scale
    return int(newh + 0.5), int(neww + 0.5)


def _apply_box(boxes, h, w, new_h, new_w, flip_width):
    """fvcore Transform.apply_box through ResizeTransform(h, w, new_h, new_w) and HFlipTransform(flip_width) (or NoOp)."""
    idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
    b = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    c = b[:, idxs].reshape(-1, 2)
    c[:, 0] = c[:, 0] * (new_w * 1.0 / w)
    c[:, 1] = c[:, 1] * (new_h * 1.0 / h)
    c = c.reshape(-1, 4, 2)
    b = np.concatenate((c.min(axis=1), c.max(axis=1)), axis=1)
    if flip_width is not None:
        c = b[:, idxs].reshape(-1, 2)
        c[:, 0] = flip_width - c[:, 0]
        c = c.reshape(-1, 4, 2)
        b = np.concatenate((c.min(axis=1), c.max(axis=1)), axis=1)
    return b


def _instances(annos, h, w, new_h, new_w, flip_width, device):
    """transform_instance_annotations + annotations_to_instances + filter_empty_instances for boxes and classes."""
    keep = [a for a in annos if a.get("iscrowd", 0) == 0]
    boxes = []
    for a in keep:
        b = [float(v) for v in a["bbox"]]
        if a.get("bbox_mode", XYXY_ABS) == XYWH_ABS:
            b = [b[0], b[1], b[0] + b[2], b[1] + b[3]]
        boxes.append(b)
    b = torch.as_tensor(_apply_box(boxes, h, w, new_h, new_w, flip_width), dtype=torch.float32).reshape(-1, 4)
    b[:, 0::2].clamp_(min=0, max=new_w)                      # Boxes.clip
    b[:, 1::2].clamp_(min=0, max=new_h)
    cls = torch.tensor([a["category_id"] for a in keep], dtype=torch.int64)
    ne = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)                # Boxes.nonempty
    return SimpleNamespace(image_size=(new_h, new_w), gt_boxes=b[ne].to(device), gt_classes=cls[ne].to(device))


class DualScaleMapper:
    """``DatasetMapper(cfg, scale_ratio=[0.5], is_train)`` with the config values passed directly:
    ``min_size`` / ``max_size`` / ``sample_style`` = INPUT.MIN_SIZE_TRAIN / MAX_SIZE_TRAIN / MIN_SIZE_TRAIN_SAMPLING
    (or the *_TEST values and "choice" when ``is_train`` is False, afigan_utils.py:438-466).

    ``share_flip``: the x0.5 list re-uses the flip decision of the first list, as ``apply_transform_gens_overlap2`` is written
    to do (transform_gen.py:546-554) and as SURVEY.md specifies.  NOTE the reference's test there is
    ``isinstance(g, T.RandomFlip)`` with ``T = detectron2.data.transforms`` while its mapper instantiates the reference's own
    copy of ``RandomFlip`` (afigan_utils.py:26,438-466): with stock detectron2 the test is False and ``image_x0.5`` is flipped by an
    independent draw (it then disagrees with ``image`` for half of the samples).  ``share_flip=False`` reproduces that
    as-written behaviour draw for draw; both variants are pinned by tests/golden/dual_scale_mapper.npz."""

    def __init__(self, min_size=(800,), max_size=1333, sample_style="choice", scale_ratio=(0.5,), is_train=True, flip_prob=0.5,
                 share_flip=True, device="cuda"):
        assert sample_style in ("range", "choice"), sample_style
        if isinstance(min_size, int):
            min_size = (min_size, min_size)
        if sample_style == "range":
            assert len(min_size) == 2, f"more than 2 ({len(min_size)}) min_size(s) are provided for ranges"
        self.min_size, self.max_size, self.is_range = tuple(min_size), max_size, sample_style == "range"
        self.scale_ratio, self.is_train, self.flip_prob, self.device = tuple(scale_ratio), is_train, flip_prob, torch.device(device)
        self.share_flip = share_flip
        assert self.scale_ratio == (0.5,), "the reference hard-codes the 0.5 ratio in its transform list (transform_gen.py:542-543)"

    def _draw(self):
        if self.is_range:
            size = np.random.randint(self.min_size[0], self.min_size[1] + 1)
        else:
            size = np.random.choice(self.min_size)
        flip = bool(np.random.uniform(0, 1) < self.flip_prob) if self.is_train else False
        return int(size), flip

    def __call__(self, dataset_dict):
        d = copy.copy(dataset_dict)
        img = d.pop("image")
        if isinstance(img, np.ndarray):
            img = torch.from_numpy(np.ascontiguousarray(img))
        if img.dtype != torch.uint8 or img.dim() != 3:
            raise _lib.AfiError(f"expected a decoded uint8 HWC image, got {img.dtype} {tuple(img.shape)}")
        img = img.to(self.device, non_blocking=True).contiguous()
        h, w = img.shape[:2]
        size, flip = self._draw()
        _, flip_r = self._draw()                                  # the x0.5 list draws again; its size is overwritten (:542-543)
        if self.share_flip:
            flip_r = flip
        if size == 0:
            raise _lib.AfiError("short-edge size 0 (NoOpTransform) is not supported on the dual-scale path")
        new_h, new_w = shortest_edge_size(h, w, size, self.max_size)
        ratio = self.scale_ratio[0]
        rh, rw = int(new_h * ratio), int(new_w * ratio)
        d["image"], image_r = ops.dual_scale_u8(img, (new_h, new_w), (rh, rw), hflip=flip, hflip_r=flip_r, chw=True)
        shp = d["image"].shape
        d[f"width_x{ratio}"], d[f"heigth_x{ratio}"] = int(shp[1] * ratio), int(shp[2] * ratio)      # sic, dataset_mapper.py:121-122
        if not self.is_train:
            d.pop("annotations", None)
            return d
        d[f"image_x{ratio}"] = image_r
        if "annotations" in d:
            annos = d.pop("annotations")
            d["instances"] = _instances(annos, h, w, new_h, new_w, new_w if flip else None, self.device)
            d[f"instances_x{ratio}"] = _instances(annos, h, w, rh, rw, rw if flip_r else None, self.device)
        return d


def preprocess_images(batched_inputs, key="image", pixel_mean=(103.530, 116.280, 123.675), pixel_std=(1.0, 1.0, 1.0),
                      size_divisibility=32):
    """RCNN_FPN_only.forward up to ``images.tensor`` (rcnn_only.py:36-39): normalise each ``x[key]`` and pad into one batch."""
    return ops.normalize_pad([x[key] for x in batched_inputs], pixel_mean, pixel_std, size_divisibility)
