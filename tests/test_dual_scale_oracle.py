"""Pin the dual-scale oracle (oracle/dual_scale_oracle.py) against Pillow outputs and against outputs of the reference's own
transform generators (tests/golden/make_golden_dual_scale.py).  CPU-only."""
import hashlib
import os

import numpy as np
import pytest

from oracle import dual_scale_oracle as dso


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def test_resize_matches_pillow_fixtures(golden_dir):
    fx = _load(golden_dir, "pil_resize.npz")
    n = 0
    while f"s{n}/in" in fx:
        src, ref = fx[f"s{n}/in"], fx[f"s{n}/out"]
        out = dso.pil_resize_bilinear_u8(src, ref.shape[0], ref.shape[1])
        assert np.array_equal(out, ref), f"small case {n}: {src.shape} -> {ref.shape}"      # bit-exact
        n += 1
    assert n >= 10


def test_resize_full_size_matches_pillow_digest(golden_dir):
    fx = _load(golden_dir, "pil_resize.npz")
    for i in (0, 1):                                  # 480x640 -> 800x1067 and -> 400x533 (the two images of one sample)
        shp, (nh, nw) = tuple(fx[f"l{i}/shape"]), fx[f"l{i}/size"]
        img = np.random.default_rng(200 + i).integers(0, 256, size=shp, dtype=np.uint8)
        out = dso.pil_resize_bilinear_u8(img, int(nh), int(nw))
        assert hashlib.sha256(out.tobytes()).hexdigest() == str(fx[f"l{i}/sha256"])
        assert np.array_equal(out[:: max(1, int(nh) // 5)][:5], fx[f"l{i}/rows"])


def test_resize_matches_live_pillow_when_present():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(7)
    for _ in range(25):
        h, w = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        nh, nw = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
        assert np.array_equal(dso.pil_resize_bilinear_u8(img, nh, nw), ref), (h, w, nh, nw)


@pytest.mark.parametrize("variant", ["as_written", "shared"])
def test_mapper_logic_matches_reference_transform_gens(golden_dir, variant):
    fx = _load(golden_dir, "dual_scale_mapper.npz")
    n = 0
    for ci in range(6):
        for seed in range(4):
            k = f"{variant}/{ci}/{seed}"
            cfg = fx[k + "/cfg"]
            min_size, max_size, style = tuple(int(v) for v in cfg[:-2]), int(cfg[-2]), "range" if cfg[-1] else "choice"
            np.random.seed(1000 * ci + seed)
            size, flip, flip_r = dso.draw_transforms(min_size, max_size, style, share_flip=(variant == "shared"))
            assert np.random.uniform() == float(fx[k + "/rng_after"])           # same number of draws, same order
            assert [flip, flip_r] == [bool(v) for v in fx[k + "/flip"]]
            boxes = fx[k + "/boxes_in"]
            out = dso.dual_scale_map(fx[k + "/in"], boxes, [0] * len(boxes), [0] * len(boxes), size, max_size, flip, flip_r)
            assert np.array_equal(out["image"], fx[k + "/image"].transpose(2, 0, 1))
            assert np.array_equal(out["image_x0.5"], fx[k + "/image_r"].transpose(2, 0, 1))
            assert np.array_equal(out["boxes_raw"], fx[k + "/boxes"])
            assert np.array_equal(out["boxes_raw_x0.5"], fx[k + "/boxes_r"])
            n += 1
    assert n == 24


def test_reference_as_written_does_not_share_the_flip(golden_dir):
    """Documents the finding in make_golden_dual_scale.py: with two distinct RandomFlip classes the x0.5 list draws its own flip."""
    fx = _load(golden_dir, "dual_scale_mapper.npz")
    flips = [fx[k] for k in fx if k.startswith("as_written") and k.endswith("/flip")]
    assert any(f[0] != f[1] for f in flips)
    assert all(f[0] == f[1] for f in (fx[k] for k in fx if k.startswith("shared") and k.endswith("/flip")))


def test_instances_clip_and_filter():
    # a 40x60 image resized to 20x30 and flipped: hand-derived expectations
    img = np.zeros((40, 60, 3), np.uint8)
    boxes = [[10, 10, 30, 30], [50, 0, 70, 20], [5, 5, 5, 25], [0, 0, 60, 40]]
    out = dso.dual_scale_map(img, boxes, [1, 2, 3, 4], [0, 0, 0, 1], size=20, max_size=100, flip=True)
    assert out["image"].shape == (3, 20, 30) and out["image_x0.5"].shape == (3, 10, 15)
    assert (out["width_x0.5"], out["heigth_x0.5"]) == (10, 15)                       # sic: half height, half width
    # box 0: x 10..30 -> 5..15 -> flipped 15..25 ; y 5..15.  box 1: x 25..35 -> flipped -5..5 -> clipped 0..5.
    # box 2 is empty (zero width) and dropped, box 3 is crowd and dropped.
    assert np.allclose(out["boxes"], [[15, 5, 25, 15], [0, 0, 5, 10]])
    assert list(out["classes"]) == [1, 2]
    assert np.allclose(out["boxes_x0.5"], [[7.5, 2.5, 12.5, 7.5], [0, 0, 2.5, 5]])


def test_normalize_pad_matches_torch_semantics():
    import torch
    rng = np.random.default_rng(3)
    imgs = [rng.integers(0, 256, size=(3, 20, 33), dtype=np.uint8), rng.integers(0, 256, size=(3, 37, 18), dtype=np.uint8)]
    mean, std = [103.53, 116.28, 123.675], [57.375, 57.12, 58.395]
    out = dso.normalize_pad(imgs, mean, std, 32)
    assert out.shape == (2, 3, 64, 64)
    m, s = torch.tensor(mean).view(3, 1, 1), torch.tensor(std).view(3, 1, 1)
    for n, i in enumerate(imgs):
        ref = (torch.from_numpy(i) - m) / s                                          # rcnn_only.py:31
        assert np.array_equal(out[n, :, :i.shape[1], :i.shape[2]], ref.numpy())
        assert not out[n, :, i.shape[1]:, :].any() and not out[n, :, :, i.shape[2]:].any()


def _anno_cases(golden_dir):
    import json
    fx = dict(np.load(os.path.join(golden_dir, "dual_scale_annos.npz")))
    annos = json.loads(str(fx["annotations_json"]))
    crops = json.loads(str(fx["crops_json"]))
    return fx, annos, crops


def test_annotation_path_matches_reference_functions(golden_dir):
    """The host side of the mapper's annotation path against outputs of the reference's OWN functions (tests/golden/dual_scale_annos.npz,
    made by tests/golden/make_golden_dual_scale.py from transform_gen.py:220-264 RandomCrop, afigan_utils.py:379-406
    gen_crop_transform_with_instance and :140-183 transform_instance_annotations, driven like dataset_mapper.py:96-109,140-176 under seeded
    numpy.random): the crop window, the flip decisions, per instance the transformed box and polygons of BOTH transform lists, and the
    position of numpy's random stream afterwards -- without a GPU (DualScaleMapper.plan is pure numpy)."""
    import torch
    from afigan_amd.dual_scale import DualScaleMapper, _apply_box_list, _apply_coords, _instances, _xyxy
    fx, annos, crops = _anno_cases(golden_dir)
    n_crop = 0
    for ci, crop in enumerate(crops):
        for seed in range(5):
            k = f"{ci}/{seed}"
            mapper = DualScaleMapper((24, 28, 32), 50, "choice", share_flip=True, device="cpu", mask_on=True, crop=tuple(crop) if crop else None)
            np.random.seed(7000 + 10 * ci + seed)
            pl = mapper.plan(48, 64, annos)
            assert np.random.uniform() == float(fx[k + "/rng_after"]), k
            want_crop = [int(v) for v in fx[k + "/crop"]]
            assert (list(pl.crop) if pl.crop else [-1, -1, -1, -1]) == want_crop, k
            n_crop += pl.crop is not None
            assert [pl.flip, pl.flip_r] == [bool(v) for v in fx[k + "/flip"]], k
            assert pl.size == fx[k + "/image"].shape[:2] and pl.size_r == fx[k + "/image_r"].shape[:2], k
            keep = [a for a in annos if a.get("iscrowd", 0) == 0]
            for tag, tf in (("", pl.tf), ("_r", pl.tf_r)):
                np.testing.assert_allclose(_apply_box_list([_xyxy(a) for a in keep], tf), fx[k + "/boxes" + tag], rtol=0, atol=1e-9, err_msg=k + tag)
                polys = np.concatenate([_apply_coords(np.asarray(q).reshape(-1, 2), tf).reshape(-1) for a in keep for q in a["segmentation"]])
                np.testing.assert_allclose(polys, fx[k + "/poly" + tag], rtol=0, atol=1e-9, err_msg=k + tag)
                assert [len(a["segmentation"]) for a in keep] == [int(v) for v in fx[k + "/poly_cnt" + tag]]
            # Instances: clipped boxes (tight around the masks when the mapper crops), classes, polygon masks, empty instances dropped
            inst = _instances(annos, pl.tf, pl.size, "cpu", mask_on=True, tight_boxes=crop is not None)
            raw = torch.from_numpy(fx[k + "/boxes"]).float()
            if crop is None:
                b = raw.clone()
                b[:, 0::2].clamp_(0, pl.size[1]); b[:, 1::2].clamp_(0, pl.size[0])
                ne = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
                assert torch.equal(inst.gt_boxes, b[ne]) and len(inst.gt_masks) == int(ne.sum())
            else:                                            # dataset_mapper.py:156-157: gt_boxes = gt_masks.get_bounding_boxes()
                assert len(inst.gt_masks) == len(inst.gt_boxes) == len(inst.gt_classes)
                for bb, polys in zip(inst.gt_boxes, inst.gt_masks.polygons):
                    allc = np.concatenate([q.reshape(-1, 2) for q in polys])
                    # (detectron2's PolygonMasks.get_bounding_boxes starts its running maximum at ZERO: a polygon left of / above the crop keeps max = 0)
                    np.testing.assert_allclose(bb.numpy(), np.concatenate([allc.min(0), np.maximum(allc.max(0), 0.0)]).astype(np.float32), rtol=0, atol=0)
    assert n_crop == 15                                      # three crop settings x five seeds took the crop branch


def test_polygon_masks_and_unsupported_mask_formats():
    from afigan_amd import AfiError
    from afigan_amd.dual_scale import DualScaleMapper, PolygonMasks
    import torch
    m = PolygonMasks([[[0, 0, 4, 0, 4, 3]], [], [[1, 1, 2, 5, 3, 1], [7, 7, 9, 7, 8, 9.5]]])
    assert m.nonempty().tolist() == [True, False, True] and len(m[torch.tensor([True, False, True])]) == 2
    assert m.get_bounding_boxes()[2].tolist() == [1.0, 1.0, 9.0, 9.5]
    with pytest.raises(AfiError):
        DualScaleMapper(mask_on=True, mask_format="bitmask", device="cpu")


def test_crop_without_annotations_is_refused_not_skipped():
    """dataset_mapper.py:84-91,96-102: with INPUT.CROP on, a sample without an "annotations" key gets BOTH images cropped by the reference
    (not mirrored here: AfiError instead of silently uncropped data), and an empty annotation list raises from np.random.choice([])."""
    from afigan_amd import _lib
    from afigan_amd.dual_scale import DualScaleMapper
    m = DualScaleMapper((24, 28, 32), 50, "choice", device="cpu", crop=("relative", (0.8, 0.8)))
    with pytest.raises(_lib.AfiError, match="annotations"):
        m.plan(40, 60, None)
    with pytest.raises(ValueError, match="non-empty"):
        m.plan(40, 60, [])
    assert DualScaleMapper((24, 28, 32), 50, "choice", device="cpu").plan(40, 60, None).crop is None      # no crop configured: unchanged
