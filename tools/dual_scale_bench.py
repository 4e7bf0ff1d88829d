"""Dual-scale data path: one COCO-sized sample (480x640x3 uint8, resident in HBM) -> `image` 800x1067 and
`image_x0.5` 400x533 (uint8 CHW) through DualScaleMapper, then normalise + pad of a 2-image batch.  Prints one JSON object."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def run(iters=200, warm=20, cpu_iters=10):
    from afigan_amd.dual_scale import DualScaleMapper, preprocess_images
    img = torch.from_numpy(np.random.default_rng(0).integers(0, 256, size=(480, 640, 3), dtype=np.uint8)).cuda()
    mapper = DualScaleMapper((800,), 1333, "choice")
    np.random.seed(0)
    def one():
        return mapper({"image": img})
    for _ in range(warm): one()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record()
    for _ in range(iters): d = one()
    b.record(); torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters * 1e3
    ms = a.elapsed_time(b) / iters
    # the same sample handed over as a host array (pageable numpy, as a loader worker would): H2D copy included
    host = img.cpu().numpy()
    for _ in range(warm): mapper({"image": host})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): mapper({"image": host})
    torch.cuda.synchronize()
    ms_host = (time.perf_counter() - t0) / iters * 1e3
    batch = [one(), one()]
    for _ in range(warm): preprocess_images(batch)
    a.record()
    for _ in range(iters): x = preprocess_images(batch)
    b.record(); torch.cuda.synchronize()
    ms_pre = a.elapsed_time(b) / iters
    H0, W0, H1, W1 = 480, 640, 800, 1067
    H2, W2 = int(H1 * 0.5), int(W1 * 0.5)
    # algorithmic bytes of one sample: source read once per resize, intermediate written and read, result written
    alg = sum(3 * (H0 * W0 + 2 * H0 * w + h * w) for h, w in ((H1, W1), (H2, W2)))
    out = {"workload": "480x640x3 uint8 -> image 800x1067 + image_x0.5 400x533 (uint8 CHW), Pillow-exact", "ms_per_sample": round(ms, 4),
           "ms_per_sample_wall": round(wall, 4), "ms_per_sample_from_host_array": round(ms_host, 4), "samples_per_s": round(1e3 / ms, 1), "algorithmic_bytes_per_sample": alg,
           "achieved_GBps": round(alg / ms / 1e6, 1), "hbm_peak_GBps": 8000,
           "normalize_pad_2x3x800x1088_ms": round(ms_pre, 4),
           "normalize_pad_GBps": round((2 * 3 * 800 * 1067 + 4 * x.numel()) / ms_pre / 1e6, 1)}
    try:                                                                    # the library the reference calls, on this host's CPU
        from PIL import Image
        src = img.cpu().numpy()
        t0 = time.perf_counter()
        for _ in range(cpu_iters):
            np.ascontiguousarray(np.asarray(Image.fromarray(src).resize((W1, H1), Image.BILINEAR)).transpose(2, 0, 1))
            np.ascontiguousarray(np.asarray(Image.fromarray(src).resize((W2, H2), Image.BILINEAR)).transpose(2, 0, 1))
        out["pillow_cpu_ms_per_sample"] = round((time.perf_counter() - t0) / cpu_iters * 1e3, 3)
        out["pillow_version"] = __import__("PIL").__version__
    except ImportError:
        pass
    return out


if __name__ == "__main__":
    print(json.dumps(run()), flush=True)
