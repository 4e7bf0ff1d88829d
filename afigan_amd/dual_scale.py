"""Dual-scale data path on the device.

Host-side mirror of the reference's ``afigan.engine.dataset_mapper.DatasetMapper`` (dataset_mapper.py:24-193) for the part
that feeds the stage-1/2 trainers: one decoded image becomes ``image`` (ResizeShortestEdge + RandomFlip) and ``image_x0.5``
(the ORIGINAL image resized straight to ``int(new_h*0.5) x int(new_w*0.5)`` with the same flip decision,
transform_gen.py:514-559), and the boxes follow both transform lists (afigan_utils.py:140-170, 234-262, 328-356).

The pixels are produced by ``csrc/resample.hip`` (Pillow-exact antialiased bilinear, flip and CHW store fused into the second
pass); sizes, random draws and the box arithmetic (a handful of float64 operations per box) stay on the host, as in the
reference.  Same dict keys as the reference, including its ``width_x0.5`` / ``heigth_x0.5`` [sic] entries, which read
``shape[1]`` and ``shape[2]`` of the CHW tensor and therefore hold half the HEIGHT and half the WIDTH (dataset_mapper.py:121-122).

Annotations (``configs[3]`` is Mask R-CNN, ``MASK_ON``): boxes and POLYGON masks follow both transform lists
(``transform_instance_annotations`` -> ``apply_box`` / ``apply_polygons``, afigan_utils.py:140-183; ``annotations_to_instances`` with
``INPUT.MASK_FORMAT = "polygon"``, the detectron2 default no reference yaml changes, :234-262; ``filter_empty_instances``, :328-354), and
``INPUT.CROP`` (``RandomCrop`` + ``gen_crop_transform_with_instance``, dataset_mapper.py:42-45,96-109; transform_gen.py:220-264;
afigan_utils.py:379-406) is mirrored AS WRITTEN -- including that only ``image`` is cropped: ``image_x0.5`` is the UNCROPPED original
resized to half of the cropped image's resized size (dataset_mapper.py:98-105; no reference yaml enables the crop).

Out of scope (not on the path): file reading / JPEG decode (``dataset_dict["image"]`` must hold the decoded uint8
HWC array), bitmask / RLE masks (rasterised by pycocotools, a third-party package the reference does not vendor), keypoints, proposals,
semantic segmentation.  Random draws come from ``numpy.random``'s global state in the reference's order (crop size, crop instance,
crop origin, size, flip, then the two discarded draws of the x0.5 list), so a seeded run picks the same crops, sizes and flips as the
reference mapper.
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


def _xyxy(a):
    b = [float(v) for v in a["bbox"]]
    if a.get("bbox_mode", XYXY_ABS) == XYWH_ABS:            # BoxMode.convert(XYWH_ABS -> XYXY_ABS)
        b = [b[0], b[1], b[0] + b[2], b[1] + b[3]]
    return b


def _apply_coords(coords, tfms):
    """TransformList.apply_coords for the three transforms of this path, in list order: ("crop", x0, y0) -> coords - (x0, y0)
    (CropTransform); ("resize", h, w, new_h, new_w) -> coords * new / old (ResizeTransform); ("hflip", width) -> x = width - x."""
    c = np.array(coords, dtype=np.float64).reshape(-1, 2)
    for t in tfms:
        if t[0] == "crop":
            c[:, 0] -= t[1]
            c[:, 1] -= t[2]
        elif t[0] == "resize":
            c[:, 0] = c[:, 0] * (t[4] * 1.0 / t[2])
            c[:, 1] = c[:, 1] * (t[3] * 1.0 / t[1])
        elif t[0] == "hflip":
            c[:, 0] = t[1] - c[:, 0]
        else:
            raise _lib.AfiError(f"unknown transform {t[0]!r}")
    return c


def _apply_box_list(boxes, tfms):
    """Transform.apply_box transform by transform (each takes the four corners through apply_coords and re-forms the min/max box)."""
    idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
    b = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    for t in tfms:
        c = _apply_coords(b[:, idxs].reshape(-1, 2), [t]).reshape(-1, 4, 2)
        b = np.concatenate((c.min(axis=1), c.max(axis=1)), axis=1)
    return b


class PolygonMasks:
    """The part of detectron2.structures.PolygonMasks this path touches: ``polygons[i]`` = the list of float64 [x0, y0, x1, y1, ...] arrays
    of instance i; ``nonempty()`` (an instance with at least one polygon); ``get_bounding_boxes()`` (min / max over its vertices)."""

    def __init__(self, polygons):
        self.polygons = [[np.asarray(p, dtype=np.float64).reshape(-1) for p in inst] for inst in polygons]

    def __len__(self):
        return len(self.polygons)

    def __getitem__(self, keep):
        keep = keep.tolist() if hasattr(keep, "tolist") else list(keep)
        return PolygonMasks([p for p, k in zip(self.polygons, keep) if k])

    def nonempty(self):
        return torch.tensor([len(inst) > 0 for inst in self.polygons], dtype=torch.bool)

    def get_bounding_boxes(self):
        """(as detectron2 v0.1.1 computes it: the running minimum starts at +inf, the running MAXIMUM at zero)"""
        out = torch.zeros((len(self.polygons), 4), dtype=torch.float32)
        for i, inst in enumerate(self.polygons):
            lo = torch.tensor([float("inf"), float("inf")]); hi = torch.zeros(2)
            for p in inst:
                c = torch.from_numpy(p.reshape(-1, 2)).to(torch.float32)
                lo, hi = torch.min(lo, c.min(dim=0).values), torch.max(hi, c.max(dim=0).values)
            out[i, :2], out[i, 2:] = lo, hi
        return out


def _instances(annos, tfms, image_size, device, mask_on=False, tight_boxes=False):
    """transform_instance_annotations (afigan_utils.py:140-183) + annotations_to_instances (:234-262) + filter_empty_instances (:328-354)
    for boxes, classes and polygon masks.  tight_boxes: the boxes re-formed from the masks, as the mapper does when it crops."""
    new_h, new_w = image_size
    keep = [a for a in annos if a.get("iscrowd", 0) == 0]
    b = torch.as_tensor(_apply_box_list([_xyxy(a) for a in keep], tfms), dtype=torch.float32).reshape(-1, 4)
    b[:, 0::2].clamp_(min=0, max=new_w)                      # Boxes.clip
    b[:, 1::2].clamp_(min=0, max=new_h)
    cls = torch.tensor([a["category_id"] for a in keep], dtype=torch.int64)
    masks = None
    if mask_on and len(keep) and "segmentation" in keep[0]:
        polys = []
        for a in keep:
            segm = a["segmentation"]
            if not isinstance(segm, list):
                raise _lib.AfiError("only polygon segmentations are supported on this path (RLE / bitmask masks need pycocotools)")
            polys.append([_apply_coords(np.asarray(q).reshape(-1, 2), tfms).reshape(-1) for q in segm])
        masks = PolygonMasks(polys)
        if tight_boxes:
            b = masks.get_bounding_boxes()                   # dataset_mapper.py:156-157,174-175
    ne = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)                # Boxes.nonempty
    if masks is not None:
        ne = ne & masks.nonempty()
    inst = SimpleNamespace(image_size=(new_h, new_w), gt_boxes=b[ne].to(device), gt_classes=cls[ne].to(device))
    if masks is not None:
        inst.gt_masks = masks[ne]
    return inst


class RandomCrop:
    """transform_gen.py:220-264: crop_type "relative_range" / "relative" / "absolute", crop_size (h, w) as ratios or pixels."""

    def __init__(self, crop_type, crop_size):
        assert crop_type in ("relative_range", "relative", "absolute"), crop_type
        self.crop_type, self.crop_size = crop_type, tuple(crop_size)

    def get_crop_size(self, image_size):
        h, w = image_size
        if self.crop_type == "relative":
            ch, cw = self.crop_size
            return int(h * ch + 0.5), int(w * cw + 0.5)
        if self.crop_type == "relative_range":
            crop_size = np.asarray(self.crop_size, dtype=np.float32)
            ch, cw = crop_size + np.random.rand(2) * (1 - crop_size)
            return int(h * ch + 0.5), int(w * cw + 0.5)
        return self.crop_size


def gen_crop_transform_with_instance(crop_size, image_size, instance):
    """afigan_utils.py:379-406: a crop window of `crop_size` that contains the centre of `instance`'s box; returns (x0, y0, w, h)."""
    crop_size = np.asarray(crop_size, dtype=np.int32)
    bbox = _xyxy(instance)
    center_yx = (bbox[1] + bbox[3]) * 0.5, (bbox[0] + bbox[2]) * 0.5
    assert image_size[0] >= center_yx[0] and image_size[1] >= center_yx[1], "The annotation bounding box is outside of the image!"
    assert image_size[0] >= crop_size[0] and image_size[1] >= crop_size[1], "Crop size is larger than image size!"
    min_yx = np.maximum(np.floor(center_yx).astype(np.int32) - crop_size, 0)
    max_yx = np.maximum(np.asarray(image_size, dtype=np.int32) - crop_size, 0)
    max_yx = np.minimum(max_yx, np.ceil(center_yx).astype(np.int32))
    y0 = np.random.randint(min_yx[0], max_yx[0] + 1)
    x0 = np.random.randint(min_yx[1], max_yx[1] + 1)
    return int(x0), int(y0), int(crop_size[1]), int(crop_size[0])


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
                 share_flip=True, device="cuda", mask_on=False, mask_format="polygon", crop=None):
        """mask_on / mask_format = MODEL.MASK_ON / INPUT.MASK_FORMAT; crop = (INPUT.CROP.TYPE, INPUT.CROP.SIZE) when INPUT.CROP.ENABLED
        (training only, dataset_mapper.py:42-45), else None."""
        assert sample_style in ("range", "choice"), sample_style
        if isinstance(min_size, int):
            min_size = (min_size, min_size)
        if sample_style == "range":
            assert len(min_size) == 2, f"more than 2 ({len(min_size)}) min_size(s) are provided for ranges"
        self.min_size, self.max_size, self.is_range = tuple(min_size), max_size, sample_style == "range"
        self.scale_ratio, self.is_train, self.flip_prob, self.device = tuple(scale_ratio), is_train, flip_prob, torch.device(device)
        self.share_flip = share_flip
        assert self.scale_ratio == (0.5,), "the reference hard-codes the 0.5 ratio in its transform list (transform_gen.py:542-543)"
        if mask_on and mask_format != "polygon":
            raise _lib.AfiError('INPUT.MASK_FORMAT "bitmask" rasterises polygons with pycocotools (not vendored by the reference); "polygon" only')
        self.mask_on = mask_on
        self.crop_gen = RandomCrop(*crop) if (crop is not None and is_train) else None

    def _draw(self):
        if self.is_range:
            size = np.random.randint(self.min_size[0], self.min_size[1] + 1)
        else:
            size = np.random.choice(self.min_size)
        flip = bool(np.random.uniform(0, 1) < self.flip_prob) if self.is_train else False
        return int(size), flip

    def plan(self, h0, w0, annotations=None):
        """The host-side decisions of one sample, in the reference's draw order (dataset_mapper.py:96-109): the crop window around one
        instance (training with INPUT.CROP and annotations), the short-edge size and flips of both lists, both output sizes and the two
        transform lists the annotations follow.  Pure numpy: no device work."""
        crop = None
        if self.crop_gen is not None and annotations is None:
            # dataset_mapper.py:84-91: without an "annotations" key the reference prepends the RandomCrop to BOTH transform lists (image and
            # image_x0.5 each randomly cropped, two more draws).  That branch is not mirrored here: refuse it instead of handing back
            # uncropped data with another random-stream position
            raise _lib.AfiError("INPUT.CROP is enabled and the sample has no \"annotations\" key: the reference's crop of both images "
                                "(dataset_mapper.py:84-91) is not supported on the dual-scale path; pass annotations or disable the crop")
        if self.crop_gen is not None and len(annotations) == 0:
            raise ValueError("a must be non-empty")            # np.random.choice([]) in the reference (dataset_mapper.py:101)
        if self.crop_gen is not None:
            crop_size = self.crop_gen.get_crop_size((h0, w0))                    # (argument order of dataset_mapper.py:98-102: the size draws first,
            inst = annotations[np.random.randint(len(annotations))]              #  then np.random.choice(annotations) = one randint draw, then the origin)
            crop = gen_crop_transform_with_instance(crop_size, (h0, w0), inst)
        h, w = (crop[3], crop[2]) if crop else (h0, w0)
        size, flip = self._draw()
        _, flip_r = self._draw()                                  # the x0.5 list draws again; its size is overwritten (:542-543)
        if self.share_flip:
            flip_r = flip
        if size == 0:
            raise _lib.AfiError("short-edge size 0 (NoOpTransform) is not supported on the dual-scale path")
        new_h, new_w = shortest_edge_size(h, w, size, self.max_size)
        ratio = self.scale_ratio[0]
        rh, rw = int(new_h * ratio), int(new_w * ratio)
        tf = ([("crop", crop[0], crop[1])] if crop else []) + [("resize", h, w, new_h, new_w)] + ([("hflip", new_w)] if flip else [])      # crop_tfm + transforms (:108-109)
        tf_r = [("resize", h0, w0, rh, rw)] + ([("hflip", int(new_w * 0.5))] if flip_r else [])           # transforms_r: no crop; width int(w * 0.5) (:547-549)
        return SimpleNamespace(crop=crop, flip=flip, flip_r=flip_r, size=(new_h, new_w), size_r=(rh, rw), tf=tf, tf_r=tf_r)

    def __call__(self, dataset_dict):
        d = copy.copy(dataset_dict)
        img = d.pop("image")
        if isinstance(img, np.ndarray):
            img = torch.from_numpy(np.ascontiguousarray(img))
        if img.dtype != torch.uint8 or img.dim() != 3:
            raise _lib.AfiError(f"expected a decoded uint8 HWC image, got {img.dtype} {tuple(img.shape)}")
        img = img.to(self.device, non_blocking=True).contiguous()
        pl = self.plan(img.shape[0], img.shape[1], d.get("annotations"))
        (new_h, new_w), (rh, rw) = pl.size, pl.size_r
        ratio = self.scale_ratio[0]
        if pl.crop is None:
            d["image"], image_r = ops.dual_scale_u8(img, (new_h, new_w), (rh, rw), hflip=pl.flip, hflip_r=pl.flip_r, chw=True)
        else:       # as written (:98-105): `image` from the crop, `image_x0.5` from the UNCROPPED original at half the crop's resized size
            cx0, cy0, cw, ch = pl.crop
            d["image"] = ops.resize_bilinear_u8(img[cy0:cy0 + ch, cx0:cx0 + cw].contiguous(), new_h, new_w, hflip=pl.flip, chw=True)
            image_r = ops.resize_bilinear_u8(img, rh, rw, hflip=pl.flip_r, chw=True)
        shp = d["image"].shape
        d[f"width_x{ratio}"], d[f"heigth_x{ratio}"] = int(shp[1] * ratio), int(shp[2] * ratio)      # sic, dataset_mapper.py:121-122
        if not self.is_train:
            d.pop("annotations", None)
            return d
        d[f"image_x{ratio}"] = image_r
        if "annotations" in d:
            annos = d.pop("annotations")
            tight = self.crop_gen is not None
            d["instances"] = _instances(annos, pl.tf, pl.size, self.device, self.mask_on, tight)
            d[f"instances_x{ratio}"] = _instances(annos, pl.tf_r, pl.size_r, self.device, self.mask_on, tight)
        return d


def preprocess_images(batched_inputs, key="image", pixel_mean=(103.530, 116.280, 123.675), pixel_std=(1.0, 1.0, 1.0),
                      size_divisibility=32):
    """RCNN_FPN_only.forward up to ``images.tensor`` (rcnn_only.py:36-39): normalise each ``x[key]`` and pad into one batch."""
    return ops.normalize_pad([x[key] for x in batched_inputs], pixel_mean, pixel_std, size_divisibility)
