#!/usr/bin/env python3
"""Golden fixtures for the dual-scale data path.  Build container only (needs /root/reference and Pillow):

    python tests/golden/make_golden_dual_scale.py

Two fixture files, arrays only:

pil_resize.npz -- `np.asarray(Image.fromarray(img).resize((new_w, new_h), Image.BILINEAR))`, the call detectron2 v0.1.1's
  ResizeTransform.apply_image makes for uint8 images, evaluated by the Pillow installed here (version recorded in the file).
  Small cases keep input and output; the full-size cases keep the seed of the input and the output's SHA-256 plus a few rows.

dual_scale_mapper.npz -- the reference's OWN transform generators, loaded by path from
  /root/reference/afigan/engine/transform_gen.py: ResizeShortestEdge.get_transform (:198-217), RandomFlip.get_transform
  (:139-148), apply_transform_gens (:438-470) and apply_transform_gens_overlap2 (:514-559), driven exactly as
  DatasetMapper.__call__ does (dataset_mapper.py:85-107) under a seeded numpy.random.  The module imports fvcore / detectron2
  names at its top; neither is installed, so stand-ins carrying the published semantics are registered for exactly those
  names (HFlipTransform = np.flip(axis=1) / x -> width - x, ResizeTransform.apply_image = the Pillow call above /
  coords * new/old, TransformList = apply in order, Transform.apply_box = corners -> min/max).  Recorded per case: the images,
  the sizes, the flip decisions and the transformed boxes of both lists.
  Two variants are recorded, because the outcome depends on a class identity in the third-party package:
  transform_gen.py:546 tests `isinstance(g, T.RandomFlip)` with T = detectron2.data.transforms, while the mapper builds its
  generators from the reference's own copy of RandomFlip (afigan_utils.py:26,438-466).  With stock detectron2 those are two
  different classes, the test is False and the x0.5 list keeps its OWN flip draw ("as_written"); when the two names denote one
  class the flip of the first list is re-used ("shared", what the function is named and documented for, and what SURVEY.md
  specifies).
"""
import copy
import hashlib
import importlib.util
import os
import sys
import types

import numpy as np
from PIL import Image
import PIL

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def pil_resize(img, new_h, new_w):
    return np.asarray(Image.fromarray(img).resize((new_w, new_h), Image.BILINEAR))


SMALL = [((37, 53, 3), (80, 133)), ((37, 53, 3), (18, 26)), ((64, 48, 3), (64, 20)), ((64, 48, 3), (100, 48)), ((31, 17), (7, 5)),
         ((9, 9, 3), (1, 1)), ((5, 7, 3), (40, 3)), ((1, 1, 3), (4, 4)), ((60, 90, 3), (13, 21)), ((23, 29, 3), (23, 29)),
         ((50, 40), (33, 77)), ((40, 50, 3), (17, 9))]
LARGE = [((480, 640, 3), (800, 1067)), ((480, 640, 3), (400, 533)), ((427, 640, 3), (800, 1199)), ((1200, 1600, 3), (800, 1067)),
         ((1200, 1600, 3), (400, 533)), ((500, 375, 3), (1067, 800)), ((500, 375, 3), (533, 400))]


def make_pil():
    fx = {"pillow_version": np.array(PIL.__version__)}
    for i, (shp, (nh, nw)) in enumerate(SMALL):
        img = np.random.default_rng(100 + i).integers(0, 256, size=shp, dtype=np.uint8)
        fx[f"s{i}/in"], fx[f"s{i}/out"] = img, pil_resize(img, nh, nw)
    for i, (shp, (nh, nw)) in enumerate(LARGE):
        img = np.random.default_rng(200 + i).integers(0, 256, size=shp, dtype=np.uint8)
        out = pil_resize(img, nh, nw)
        fx[f"l{i}/shape"], fx[f"l{i}/size"] = np.array(shp), np.array([nh, nw])
        fx[f"l{i}/sha256"] = np.array(hashlib.sha256(out.tobytes()).hexdigest())
        fx[f"l{i}/rows"] = out[:: max(1, nh // 5)][:5]
    np.savez_compressed(os.path.join(HERE, "pil_resize.npz"), **fx)
    print("pil_resize.npz:", len(SMALL), "small +", len(LARGE), "large cases, Pillow", PIL.__version__)


def install_shims():
    """Stand-ins for the fvcore / detectron2 names transform_gen.py imports at module top."""
    class Transform:
        def apply_box(self, box):
            idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
            coords = np.asarray(box).reshape(-1, 4)[:, idxs].reshape(-1, 2)
            coords = self.apply_coords(coords).reshape((-1, 4, 2))
            return np.concatenate((coords.min(axis=1), coords.max(axis=1)), axis=1)

    class NoOpTransform(Transform):
        def apply_image(self, img): return img
        def apply_coords(self, coords): return coords

    class HFlipTransform(Transform):
        def __init__(self, width): self.width = width
        def apply_image(self, img): return np.flip(img, axis=1)
        def apply_coords(self, coords):
            coords[:, 0] = self.width - coords[:, 0]
            return coords

    class VFlipTransform(Transform):
        def __init__(self, height): self.height = height

    class ResizeTransform(Transform):
        def __init__(self, h, w, new_h, new_w, interp):
            self.h, self.w, self.new_h, self.new_w, self.interp = h, w, new_h, new_w, interp
        def apply_image(self, img, interp=None):
            assert img.shape[:2] == (self.h, self.w)
            return np.asarray(Image.fromarray(img).resize((self.new_w, self.new_h), interp if interp is not None else self.interp))
        def apply_coords(self, coords):
            coords[:, 0] = coords[:, 0] * (self.new_w * 1.0 / self.w)
            coords[:, 1] = coords[:, 1] * (self.new_h * 1.0 / self.h)
            return coords

    class TransformList:
        def __init__(self, transforms): self.transforms = transforms
        def apply_box(self, box):
            for t in self.transforms:
                box = t.apply_box(box)
            return box

    class _Unused(Transform):
        pass

    # fvcore semantics the annotation path needs (afigan_utils.py:140-183; dataset_mapper.py:96-109): polygons go through apply_coords one
    # by one, a CropTransform shifts coordinates by its origin and slices the image, `crop_tfm + TransformList` prepends
    Transform.apply_polygons = lambda self, polygons: [self.apply_coords(p) for p in polygons]

    class CropTransform(Transform):
        def __init__(self, x0, y0, w, h): self.x0, self.y0, self.w, self.h = x0, y0, w, h
        def apply_image(self, img): return img[self.y0:self.y0 + self.h, self.x0:self.x0 + self.w]
        def apply_coords(self, coords):
            coords[:, 0] -= self.x0
            coords[:, 1] -= self.y0
            return coords
        def __add__(self, other): return TransformList([self] + list(other.transforms))

    TransformList.apply_polygons = lambda self, polygons: _chain(self, polygons)

    def _chain(tl, polygons):
        for t in tl.transforms:
            polygons = t.apply_polygons(polygons)
        return polygons

    class D2RandomFlip:          # detectron2.data.transforms.RandomFlip: a class of its own, unrelated to the reference's copy
        pass

    fv = types.ModuleType("fvcore"); fvt = types.ModuleType("fvcore.transforms"); fvtt = types.ModuleType("fvcore.transforms.transform")
    for n, c in dict(BlendTransform=_Unused, CropTransform=CropTransform, HFlipTransform=HFlipTransform, NoOpTransform=NoOpTransform,
                     Transform=Transform, TransformList=TransformList, VFlipTransform=VFlipTransform).items():
        setattr(fvtt, n, c)
    d2 = types.ModuleType("detectron2"); d2d = types.ModuleType("detectron2.data"); d2t = types.ModuleType("detectron2.data.transforms")
    d2tt = types.ModuleType("detectron2.data.transforms.transform")
    d2tt.ExtentTransform, d2tt.ResizeTransform = _Unused, ResizeTransform
    d2t.ResizeTransform, d2t.HFlipTransform, d2t.VFlipTransform, d2t.RandomFlip = ResizeTransform, HFlipTransform, VFlipTransform, D2RandomFlip
    d2d.transforms = d2t
    sys.modules.update({"fvcore": fv, "fvcore.transforms": fvt, "fvcore.transforms.transform": fvtt, "detectron2": d2,
                        "detectron2.data": d2d, "detectron2.data.transforms": d2t, "detectron2.data.transforms.transform": d2tt})
    return HFlipTransform


def make_mapper():
    HFlip = install_shims()
    spec = importlib.util.spec_from_file_location("ref_transform_gen", os.path.join(REF, "afigan/engine/transform_gen.py"))
    tg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tg)
    fx = {}
    cases = [((48, 64, 3), (24, 28, 32), 44, "choice"), ((64, 48, 3), (24, 28, 32), 40, "choice"), ((50, 70, 3), (20, 30), 36, "range"),
             ((33, 47, 3), (25,), 60, "choice"), ((47, 33, 3), (40,), 45, "choice"), ((40, 40, 3), (21,), 30, "choice")]
    boxes = np.array([[3.0, 4.0, 20.5, 30.25], [0.0, 0.0, 47.0, 33.0], [10.0, 5.0, 10.0, 25.0], [30.0, 2.0, 46.5, 31.0]])
    n = 0
    for variant in ("as_written", "shared"):
        tg.T.RandomFlip = tg.RandomFlip if variant == "shared" else sys.modules["detectron2.data.transforms"].RandomFlip
        for ci, (shp, min_size, max_size, style) in enumerate(cases):
            for seed in range(4):
                img = np.random.default_rng(300 + ci).integers(0, 256, size=shp, dtype=np.uint8)
                gens = [tg.ResizeShortestEdge(min_size, max_size, style), tg.RandomFlip()]
                gens_r = copy.deepcopy(gens)
                np.random.seed(1000 * ci + seed)
                image, tfms = tg.apply_transform_gens(gens, img)                                   # dataset_mapper.py:85-87 / 103
                image_r, tfms_r = tg.apply_transform_gens_overlap2(gens_r, copy.deepcopy(img), tfms)   # :89-91 / 105
                k = f"{variant}/{ci}/{seed}"
                fx[k + "/in"], fx[k + "/image"], fx[k + "/image_r"] = img, np.ascontiguousarray(image), np.ascontiguousarray(image_r)
                fx[k + "/cfg"] = np.array(list(min_size) + [max_size, int(style == "range")])
                fx[k + "/flip"] = np.array([isinstance(tfms.transforms[-1], HFlip), isinstance(tfms_r.transforms[-1], HFlip)])
                fx[k + "/boxes_in"] = boxes
                fx[k + "/boxes"] = tfms.apply_box(boxes.copy())
                fx[k + "/boxes_r"] = tfms_r.apply_box(boxes.copy())
                fx[k + "/rng_after"] = np.array(np.random.uniform())      # how much of numpy's global stream the two lists consumed
                n += 1
    np.savez_compressed(os.path.join(HERE, "dual_scale_mapper.npz"), **fx)
    shared = sum(bool(fx[k][0] == fx[k][1]) for k in fx if k.startswith("as_written") and k.endswith("/flip"))
    print(f"dual_scale_mapper.npz: {n} cases; as written, the two lists agree on the flip in {shared} of {n // 2}")


def make_annos():
    """dual_scale_annos.npz -- the annotation side of the mapper with the reference's OWN code: RandomCrop.get_crop_size (transform_gen.py:245-264),
    gen_crop_transform_with_instance (afigan_utils.py:379-406) and transform_instance_annotations (:140-183), loaded by path and driven exactly as
    DatasetMapper.__call__ does (dataset_mapper.py:96-109,140-176) under a seeded numpy.random: the crop window, both images, and per instance the
    transformed box and polygons of both lists.  afigan_utils.py imports pycocotools / detectron2.structures / fvcore.common at its top: stand-ins
    for exactly those names (BoxMode.convert XYWH_ABS -> XYXY_ABS is the only one the recorded functions execute)."""
    HFlip = install_shims()
    spec = importlib.util.spec_from_file_location("ref_transform_gen", os.path.join(REF, "afigan/engine/transform_gen.py"))
    tg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tg)
    tg.T.RandomFlip = tg.RandomFlip                          # the "shared" flip variant

    class BoxMode:
        XYXY_ABS, XYWH_ABS = 0, 1
        @staticmethod
        def convert(box, from_mode, to_mode):
            b = [float(v) for v in box]
            assert to_mode == 0
            return [b[0], b[1], b[0] + b[2], b[1] + b[3]] if from_mode == 1 else b
    st = types.ModuleType("detectron2.structures")
    for n in ("BitMasks", "Boxes", "Instances", "Keypoints", "PolygonMasks", "RotatedBoxes", "polygons_to_bitmask"):
        setattr(st, n, type(n, (), {}))
    st.BoxMode = BoxMode
    cat = types.ModuleType("detectron2.data.catalog"); cat.MetadataCatalog = object()
    pm = types.ModuleType("pycocotools.mask")
    fio = types.ModuleType("fvcore.common.file_io"); fio.PathManager = object()
    eng = types.ModuleType("afigan.engine"); eng.transform_gen = tg
    sys.modules.update({"detectron2.structures": st, "detectron2.data.catalog": cat, "pycocotools": types.ModuleType("pycocotools"), "pycocotools.mask": pm,
                        "fvcore.common": types.ModuleType("fvcore.common"), "fvcore.common.file_io": fio, "afigan": types.ModuleType("afigan"),
                        "afigan.engine": eng, "afigan.engine.transform_gen": tg})
    spec = importlib.util.spec_from_file_location("ref_afigan_utils", os.path.join(REF, "afigan/engine/afigan_utils.py"))
    au = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(au)

    annos0 = [
        {"bbox": [10.0, 8.0, 30.0, 24.0], "bbox_mode": 1, "category_id": 3, "iscrowd": 0,
         "segmentation": [[12.0, 9.5, 38.0, 10.0, 39.5, 30.0, 11.0, 31.5], [20.0, 15.0, 25.0, 15.0, 22.5, 20.0]]},
        {"bbox": [40.0, 5.0, 62.0, 40.0], "bbox_mode": 0, "category_id": 7, "iscrowd": 0, "segmentation": [[41.0, 6.0, 61.0, 7.5, 55.0, 39.0, 43.5, 36.0]]},
        {"bbox": [0.0, 30.0, 18.0, 16.0], "bbox_mode": 1, "category_id": 1, "iscrowd": 1, "segmentation": [[1.0, 31.0, 17.0, 31.0, 17.0, 45.0, 1.0, 45.0]]},
        {"bbox": [50.0, 28.0, 13.5, 19.0], "bbox_mode": 1, "category_id": 5, "iscrowd": 0, "segmentation": [[50.5, 28.5, 63.0, 29.0, 62.0, 46.5, 51.0, 45.0]]},
    ]
    crops = [None, ("relative_range", (0.6, 0.7)), ("absolute", (30, 36)), ("relative", (0.5, 0.75))]
    fx, n = {}, 0
    for ci, crop in enumerate(crops):
        for seed in range(5):
            img = np.random.default_rng(400 + seed).integers(0, 256, size=(48, 64, 3), dtype=np.uint8)
            annos = copy.deepcopy(annos0)
            gens = [tg.ResizeShortestEdge((24, 28, 32), 50, "choice"), tg.RandomFlip()]
            gens_r = copy.deepcopy(gens)
            np.random.seed(7000 + 10 * ci + seed)
            image = img
            crop_tfm = None
            if crop is not None:                                                                   # dataset_mapper.py:96-102
                crop_gen = tg.RandomCrop(*crop)
                crop_tfm = au.gen_crop_transform_with_instance(crop_gen.get_crop_size(image.shape[:2]), image.shape[:2], np.random.choice(annos))
                image = crop_tfm.apply_image(image)
            image, transforms = tg.apply_transform_gens(gens, image)                               # :103
            image_r, transforms_r = tg.apply_transform_gens_overlap2(gens_r, copy.deepcopy(img), transforms)   # :105
            if crop_tfm is not None:
                transforms = crop_tfm + transforms                                                 # :108-109
            k = f"{ci}/{seed}"
            fx[k + "/in"], fx[k + "/image"], fx[k + "/image_r"] = img, np.ascontiguousarray(image), np.ascontiguousarray(image_r)
            fx[k + "/crop"] = np.array([crop_tfm.x0, crop_tfm.y0, crop_tfm.w, crop_tfm.h] if crop_tfm is not None else [-1, -1, -1, -1])
            fx[k + "/flip"] = np.array([any(isinstance(t, HFlip) for t in transforms.transforms), any(isinstance(t, HFlip) for t in transforms_r.transforms)])
            for tag, tl, shape in (("", transforms, image.shape[:2]), ("_r", transforms_r, image_r.shape[:2])):
                out = [au.transform_instance_annotations(copy.deepcopy(a), tl, shape) for a in annos if a.get("iscrowd", 0) == 0]      # :140-147,163-169
                fx[k + "/boxes" + tag] = np.array([o["bbox"] for o in out], dtype=np.float64)
                fx[k + "/poly_len" + tag] = np.array([len(q) for o in out for q in o["segmentation"]])
                fx[k + "/poly_cnt" + tag] = np.array([len(o["segmentation"]) for o in out])
                fx[k + "/poly" + tag] = np.concatenate([np.asarray(q, dtype=np.float64) for o in out for q in o["segmentation"]])
            fx[k + "/rng_after"] = np.array(np.random.uniform())
            n += 1
    import json
    fx["annotations_json"] = np.array(json.dumps(annos0))
    fx["crops_json"] = np.array(json.dumps(crops))
    np.savez_compressed(os.path.join(HERE, "dual_scale_annos.npz"), **fx)
    print(f"dual_scale_annos.npz: {n} cases ({len(crops)} crop settings x 5 seeds)")


if __name__ == "__main__":
    make_pil()
    make_mapper()
    make_annos()
