"""CPU restatement of the dual-scale data path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product path
(afigan_amd.dual_scale + csrc/resample.hip) never does.

What it restates
  * the reference's mapper: DatasetMapper.__call__ (afigan/engine/dataset_mapper.py:69-193): `image` goes through
    ResizeShortestEdge + RandomFlip, `image_x0.5` is the ORIGINAL image resized straight to
    (int(new_h*0.5), int(new_w*0.5)) with the flip decision shared (afigan/engine/transform_gen.py:514-559),
    boxes go through the same transform lists, are clipped and empty ones dropped (afigan_utils.py:140-170,234-262,328-356);
  * ResizeShortestEdge.get_transform (afigan/engine/transform_gen.py:171-217);
  * the arithmetic under ResizeTransform.apply_image, which lives in a third-party dependency that is NOT under
    /root/reference: detectron2 v0.1.1 (README.md:55) calls `PIL.Image.fromarray(img).resize((new_w, new_h), BILINEAR)` for
    uint8 images, i.e. Pillow's ImagingResample (src/libImaging/Resample.c): triangle filter whose support grows with the
    down-scale factor, coefficients computed in double and rounded to 22-bit fixed point, a horizontal pass rounded to uint8 and
    then a vertical pass.  The reference pins no Pillow version (requirements.txt pins only timm and dataclasses); this file is
    PINNED against Pillow 12.2.0 -- the version importable in the build container -- through tests/golden/pil_resize.npz (made by
    tests/golden/make_golden_dual_scale.py) and, where Pillow is importable, against live calls (tests/test_oracle_golden.py);
  * fvcore's Transform semantics for boxes (ResizeTransform.apply_coords: x*new_w/w, y*new_h/h; HFlipTransform.apply_coords:
    x -> width - x; Transform.apply_box: transform the four corners, take min/max) and detectron2's Boxes.clip / nonempty and
    ImageList.from_tensors -- published third-party behaviour.  The size / flip / half-size logic and the random-draw order ARE
    pinned: tests/golden/dual_scale_mapper.npz holds outputs of the reference's own transform_gen.py run under a seeded
    numpy.random (in both flip-sharing variants, see make_golden_dual_scale.py).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def resample_coeffs(in_size: int, out_size: int):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter over the whole axis (box = 0..in_size).
    Returns (ksize, bounds[out,2] int32, kk[out,ksize] int32)."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            if a < 0.0:
                a = -a
            w = 1.0 - a if a < 1.0 else 0.0
            k[x] = w
            ww += w
        if ww != 0.0:
            k[:xmax] = k[:xmax] / ww
        for x in range(ksize):
            v = k[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _pass_axis0(img: np.ndarray, out_size: int) -> np.ndarray:
    """One resampling pass along axis 0 of a uint8 array (any trailing shape); int32 accumulation, clip8 as Pillow."""
    in_size = img.shape[0]
    ksize, bounds, kk = resample_coeffs(in_size, out_size)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        xmin, xmax = bounds[xx]
        w = kk[xx, :xmax].astype(np.int64).reshape((-1,) + (1,) * (img.ndim - 1))
        acc = (src[xmin:xmin + xmax] * w).sum(axis=0) + (1 << (PRECISION_BITS - 1))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def pil_resize_bilinear_u8(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """`np.asarray(Image.fromarray(img).resize((new_w, new_h), Image.BILINEAR))` for uint8 [H,W] or [H,W,C]:
    horizontal pass first (skipped when the width is unchanged), then the vertical one (Resample.c, ImagingResampleInner)."""
    assert img.dtype == np.uint8 and img.ndim in (2, 3)
    out = img
    if new_w != img.shape[1]:
        out = np.swapaxes(_pass_axis0(np.ascontiguousarray(np.swapaxes(out, 0, 1)), new_w), 0, 1)
    if new_h != img.shape[0]:
        out = _pass_axis0(np.ascontiguousarray(out), new_h)
    return np.ascontiguousarray(out)


def shortest_edge_size(h: int, w: int, size: int, max_size: int):
    """ResizeShortestEdge.get_transform, transform_gen.py:198-217 (the random draw of `size` is the caller's)."""
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
    """fvcore Transform.apply_box through [ResizeTransform(h, w, new_h, new_w), HFlipTransform(flip_width) | NoOp]."""
    b = np.asarray(boxes, np.float64).reshape(-1, 4)
    idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
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


def _instances(boxes_xyxy, classes, crowd, h, w, new_h, new_w, flip_width):
    keep = [i for i in range(len(classes)) if not crowd[i]]
    b = _apply_box(np.asarray(boxes_xyxy, np.float64).reshape(-1, 4)[keep], h, w, new_h, new_w, flip_width).astype(np.float32)
    b[:, 0::2] = np.clip(b[:, 0::2], 0, new_w)          # Boxes.clip
    b[:, 1::2] = np.clip(b[:, 1::2], 0, new_h)
    cls = np.asarray(classes, np.int64)[keep]
    ne = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)     # Boxes.nonempty(threshold=0)
    return b[ne], cls[ne]


def draw_transforms(min_size, max_size, sample_style, is_train=True, share_flip=True, flip_prob=0.5):
    """The numpy.random draws of the two transform lists, in the reference's order (transform_gen.py:191-196, :139-141 via
    apply_transform_gens :438-470 and apply_transform_gens_overlap2 :514-559): size, flip for `image`; then size (overwritten by
    :542-543) and flip again for `image_x0.5`.  share_flip=True: the x0.5 list re-uses the first flip (:546-554 when the
    isinstance test holds); False: it keeps its own draw (what happens with stock detectron2, see make_golden_dual_scale.py)."""
    def one():
        if sample_style == "range":
            size = np.random.randint(min_size[0], min_size[1] + 1)
        else:
            size = np.random.choice(min_size)
        flip = bool(np.random.uniform(0, 1) < flip_prob) if is_train else False
        return int(size), flip
    size, flip = one()
    _, flip2 = one()
    return size, flip, (flip if share_flip else flip2)


def dual_scale_map(image: np.ndarray, boxes_xyxy, classes, crowd, size: int, max_size: int, flip: bool, flip_r=None, ratio: float = 0.5):
    """dataset_mapper.py:69-193 for one decoded uint8 HWC image with XYXY_ABS boxes; `size`, `flip` (and `flip_r` for the
    x0.5 list, default: shared) are the random draws.  Returns the dict entries the stage-1/2 trainers read."""
    if flip_r is None:
        flip_r = flip
    h, w = image.shape[:2]
    new_h, new_w = shortest_edge_size(h, w, size, max_size)
    img = pil_resize_bilinear_u8(image, new_h, new_w)
    rh, rw = int(new_h * ratio), int(new_w * ratio)                # transform_gen.py:542-543
    img_r = pil_resize_bilinear_u8(image, rh, rw)                  # from the ORIGINAL, not from `img`
    if flip:
        img = img[:, ::-1]
    if flip_r:
        img_r = img_r[:, ::-1]
    out = {"image": np.ascontiguousarray(img.transpose(2, 0, 1)), "image_x0.5": np.ascontiguousarray(img_r.transpose(2, 0, 1))}
    chw = out["image"].shape
    # dataset_mapper.py:121-122 reads shape[1], shape[2] of the CHW tensor: "width" is the height and "heigth" [sic] the width
    out["width_x0.5"], out["heigth_x0.5"] = int(chw[1] * ratio), int(chw[2] * ratio)
    out["boxes_raw"] = _apply_box(boxes_xyxy, h, w, new_h, new_w, new_w if flip else None)
    out["boxes_raw_x0.5"] = _apply_box(boxes_xyxy, h, w, rh, rw, rw if flip_r else None)
    out["boxes"], out["classes"] = _instances(boxes_xyxy, classes, crowd, h, w, new_h, new_w, new_w if flip else None)
    out["boxes_x0.5"], out["classes_x0.5"] = _instances(boxes_xyxy, classes, crowd, h, w, rh, rw, rw if flip_r else None)
    return out


def normalize_pad(images_chw, mean, std, divisibility: int) -> np.ndarray:
    """RCNN_FPN_only.forward (rcnn_only.py:36-39): (x - mean) / std per image in fp32, then ImageList.from_tensors:
    zero-pad every image (bottom/right) to the batch maximum rounded up to `divisibility`."""
    mean = np.asarray(mean, np.float32).reshape(-1, 1, 1)
    std = np.asarray(std, np.float32).reshape(-1, 1, 1)
    hm = max(i.shape[1] for i in images_chw)
    wm = max(i.shape[2] for i in images_chw)
    if divisibility > 0:
        hm = int(math.ceil(hm / divisibility) * divisibility)
        wm = int(math.ceil(wm / divisibility) * divisibility)
    out = np.zeros((len(images_chw), images_chw[0].shape[0], hm, wm), np.float32)
    for n, i in enumerate(images_chw):
        out[n, :, :i.shape[1], :i.shape[2]] = (i.astype(np.float32) - mean) / std
    return out
