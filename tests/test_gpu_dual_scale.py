"""Dual-scale data path on the GPU (csrc/resample.hip through the C-ABI) against the CPU oracle and the Pillow / reference
fixtures.  uint8 results are held to BIT-EXACT equality; the box tensors (fp32) and the normalised batch (fp32, one correctly
rounded subtraction and division per element) to exact equality too."""
import hashlib
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_resize_matches_pillow_fixtures(golden_dir):
    from afigan_amd import ops
    fx = _load(golden_dir, "pil_resize.npz")
    n = 0
    while f"s{n}/in" in fx:
        src, ref = fx[f"s{n}/in"], fx[f"s{n}/out"]
        out = ops.resize_bilinear_u8(_dev(src), ref.shape[0], ref.shape[1], chw=False).cpu().numpy()
        assert np.array_equal(out, ref), f"small case {n}: {src.shape} -> {ref.shape}"
        n += 1
    assert n >= 10


def test_resize_full_size_matches_pillow_digests(golden_dir):
    from afigan_amd import ops
    fx = _load(golden_dir, "pil_resize.npz")
    i = 0
    while f"l{i}/shape" in fx:
        shp, (nh, nw) = tuple(int(v) for v in fx[f"l{i}/shape"]), (int(v) for v in fx[f"l{i}/size"])
        img = np.random.default_rng(200 + i).integers(0, 256, size=shp, dtype=np.uint8)
        out = ops.resize_bilinear_u8(_dev(img), nh, nw, chw=False).cpu().numpy()
        assert hashlib.sha256(out.tobytes()).hexdigest() == str(fx[f"l{i}/sha256"]), f"large case {i}"
        i += 1
    assert i == 7


def test_resize_sweep_matches_oracle_with_flip_and_layouts():
    from afigan_amd import ops
    from oracle import dual_scale_oracle as dso
    rng = np.random.default_rng(11)
    for it in range(40):
        h, w = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        nh, nw = int(rng.integers(1, 120)), int(rng.integers(1, 120))
        gray = it % 5 == 0
        img = rng.integers(0, 256, size=(h, w) if gray else (h, w, 3), dtype=np.uint8)
        ref = dso.pil_resize_bilinear_u8(img, nh, nw)
        flip, chw = bool(it & 1), bool(it & 2)
        if flip:
            ref = ref[:, ::-1]
        if chw and not gray:
            ref = ref.transpose(2, 0, 1)
        out = ops.resize_bilinear_u8(_dev(img), nh, nw, hflip=flip, chw=chw).cpu().numpy()
        assert np.array_equal(out, ref), (h, w, nh, nw, gray, flip, chw)
    # extreme ratios: 1 pixel out of many and many out of 1 pixel, saturated inputs (clip8 at both ends)
    for img, (nh, nw) in [(np.full((97, 131, 3), 255, np.uint8), (1, 1)), (np.zeros((1, 1, 3), np.uint8), (64, 64)),
                          (rng.integers(0, 256, size=(300, 7, 3), dtype=np.uint8), (3, 200))]:
        out = ops.resize_bilinear_u8(_dev(img), nh, nw, chw=False).cpu().numpy()
        assert np.array_equal(out, dso.pil_resize_bilinear_u8(img, nh, nw))


@pytest.mark.parametrize("variant", ["as_written", "shared"])
def test_mapper_matches_reference_transform_gens(golden_dir, variant):
    """The device mapper under the reference's own seeds: same sizes, flips, pixels and boxes as transform_gen.py produced."""
    from afigan_amd.dual_scale import DualScaleMapper
    fx = _load(golden_dir, "dual_scale_mapper.npz")
    for ci in range(6):
        for seed in range(4):
            k = f"{variant}/{ci}/{seed}"
            cfg = fx[k + "/cfg"]
            min_size, max_size, style = tuple(int(v) for v in cfg[:-2]), int(cfg[-2]), "range" if cfg[-1] else "choice"
            mapper = DualScaleMapper(min_size, max_size, style, share_flip=(variant == "shared"))
            annos = [{"bbox": list(b), "bbox_mode": 0, "category_id": i} for i, b in enumerate(fx[k + "/boxes_in"])]
            np.random.seed(1000 * ci + seed)
            d = mapper({"image": fx[k + "/in"], "annotations": annos, "file_name": "x.jpg"})
            assert np.random.uniform() == float(fx[k + "/rng_after"])
            assert np.array_equal(d["image"].cpu().numpy(), fx[k + "/image"].transpose(2, 0, 1))
            assert np.array_equal(d["image_x0.5"].cpu().numpy(), fx[k + "/image_r"].transpose(2, 0, 1))
            assert d["file_name"] == "x.jpg" and "annotations" not in d
            for key, bk, img_k in (("instances", "/boxes", "/image"), ("instances_x0.5", "/boxes_r", "/image_r")):
                H, W = fx[k + img_k].shape[:2]
                b = torch.from_numpy(fx[k + bk]).float()
                b[:, 0::2].clamp_(0, W); b[:, 1::2].clamp_(0, H)
                keep = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
                assert torch.equal(d[key].gt_boxes.cpu(), b[keep]) and d[key].image_size == (H, W)
                assert torch.equal(d[key].gt_classes.cpu(), torch.arange(len(b))[keep])


def test_mapper_full_size_matches_oracle_and_eval_mode():
    from afigan_amd.dual_scale import DualScaleMapper
    from oracle import dual_scale_oracle as dso
    img = np.random.default_rng(5).integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
    boxes = [[10.5, 20.0, 300.25, 400.0], [600.0, 0.0, 640.0, 480.0]]
    annos = [{"bbox": [b[0], b[1], b[2] - b[0], b[3] - b[1]], "bbox_mode": 1, "category_id": 7 + i, "iscrowd": 0} for i, b in enumerate(boxes)]
    np.random.seed(3)
    size, flip, flip_r = dso.draw_transforms((640, 672, 704, 736, 768, 800), 1333, "choice")
    ref = dso.dual_scale_map(img, boxes, [7, 8], [0, 0], size, 1333, flip, flip_r)
    np.random.seed(3)
    d = DualScaleMapper((640, 672, 704, 736, 768, 800), 1333, "choice")({"image": img, "annotations": annos})
    assert np.array_equal(d["image"].cpu().numpy(), ref["image"]) and np.array_equal(d["image_x0.5"].cpu().numpy(), ref["image_x0.5"])
    assert (d["width_x0.5"], d["heigth_x0.5"]) == (ref["width_x0.5"], ref["heigth_x0.5"])
    assert np.array_equal(d["instances"].gt_boxes.cpu().numpy(), ref["boxes"])
    assert np.array_equal(d["instances_x0.5"].gt_boxes.cpu().numpy(), ref["boxes_x0.5"])
    assert d["instances"].gt_classes.tolist() == [7, 8]
    # inference mapper: no flip, no x0.5 image, annotations dropped (dataset_mapper.py:129-137)
    np.random.seed(3)
    e = DualScaleMapper((800,), 1333, "choice", is_train=False)({"image": img, "annotations": annos})
    assert "image_x0.5" not in e and "annotations" not in e and "instances" not in e
    assert np.array_equal(e["image"].cpu().numpy(), dso.pil_resize_bilinear_u8(img, 800, 1067).transpose(2, 0, 1))


def test_normalize_pad_matches_oracle():
    from afigan_amd import dual_scale
    from oracle import dual_scale_oracle as dso
    rng = np.random.default_rng(9)
    imgs = [rng.integers(0, 256, size=(3, 40, 61), dtype=np.uint8), rng.integers(0, 256, size=(3, 70, 33), dtype=np.uint8)]
    for mean, std in (([103.53, 116.28, 123.675], [1.0, 1.0, 1.0]), ([123.675, 116.28, 103.53], [58.395, 57.12, 57.375])):
        ref = dso.normalize_pad(imgs, mean, std, 32)
        out = dual_scale.preprocess_images([{"image": _dev(i)} for i in imgs], "image", mean, std, 32)
        assert out.shape == (2, 3, 96, 64) and np.array_equal(out.cpu().numpy(), ref)


def test_unsupported_inputs_fail_loudly():
    from afigan_amd import ops, _lib
    with pytest.raises(_lib.AfiError):
        ops.resize_bilinear_u8(torch.zeros(4, 4, 3, dtype=torch.uint8), 8, 8)                       # CPU tensor: no fallback
    with pytest.raises(_lib.AfiError):
        ops.resize_bilinear_u8(torch.zeros(4, 4, 4, dtype=torch.uint8, device="cuda"), 8, 8)        # RGBA resizes premultiplied in Pillow
    with pytest.raises(_lib.AfiError):
        ops.resize_bilinear_u8(torch.zeros(4, 4, 3, dtype=torch.float32, device="cuda"), 8, 8)


def test_mapper_with_masks_and_random_crop_matches_reference_functions(golden_dir):
    """DualScaleMapper(mask_on=True, crop=...) end to end on the device under the reference's own seeds (tests/golden/dual_scale_annos.npz):
    the cropped + resized `image`, the UNCROPPED `image_x0.5` (dataset_mapper.py:98-105, as written), boxes, polygon masks, random stream."""
    import json
    from afigan_amd.dual_scale import DualScaleMapper
    fx = _load(golden_dir, "dual_scale_annos.npz")
    annos = json.loads(str(fx["annotations_json"]))
    crops = json.loads(str(fx["crops_json"]))
    for ci, crop in enumerate(crops):
        for seed in range(5):
            k = f"{ci}/{seed}"
            mapper = DualScaleMapper((24, 28, 32), 50, "choice", share_flip=True, mask_on=True, crop=tuple(crop) if crop else None)
            np.random.seed(7000 + 10 * ci + seed)
            d = mapper({"image": fx[k + "/in"], "annotations": [dict(a) for a in annos]})
            assert np.random.uniform() == float(fx[k + "/rng_after"]), k
            assert np.array_equal(d["image"].cpu().numpy(), fx[k + "/image"].transpose(2, 0, 1)), k
            assert np.array_equal(d["image_x0.5"].cpu().numpy(), fx[k + "/image_r"].transpose(2, 0, 1)), k
            for key, tag in (("instances", ""), ("instances_x0.5", "_r")):
                inst = d[key]
                assert inst.image_size == fx[k + "/image" + tag].shape[:2]
                lens = [int(v) for v in fx[k + "/poly_len" + tag]]
                cnts = [int(v) for v in fx[k + "/poly_cnt" + tag]]
                flat = fx[k + "/poly" + tag]
                want, o, j = [], 0, 0
                for c in cnts:                                # the fixture's polygons per (non-crowd) instance
                    inst_p = []
                    for _ in range(c):
                        inst_p.append(flat[o:o + lens[j]]); o += lens[j]; j += 1
                    want.append(inst_p)
                got = inst.gt_masks.polygons
                assert len(got) <= len(want)
                gi = 0
                for w in want:                                # every kept instance carries the reference's polygons, in order
                    if gi < len(got) and len(got[gi]) == len(w) and all(np.allclose(a, b, rtol=0, atol=1e-9) for a, b in zip(got[gi], w)):
                        gi += 1
                assert gi == len(got), k
