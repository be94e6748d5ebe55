#!/usr/bin/env python3
"""Throughput of the GPU input pipeline (fea_resize_u8: resize + ToTensor + Normalize, and
the mask path) on images of the shipped data's shape (1105x800 -> 512x512), inputs resident
in HBM, next to the CPU transform (Pillow + torch, what the reference's DataLoader workers
run, custom_dataset.py:23-28) on one core.  Prints one JSON line."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from fea_amd.data import ImagePreprocessor  # noqa: E402


def main():
    h, w, S, n = 800, 1105, 512, 256
    rng = np.random.default_rng(0)
    imgs = [torch.from_numpy(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).cuda() for _ in range(8)]
    masks = [torch.from_numpy(rng.integers(0, 256, (S, S), dtype=np.uint8)).cuda() for _ in range(8)]
    pre = ImagePreprocessor(S)
    x = torch.empty((n, 3, S, S), dtype=torch.float32, device="cuda")
    t = torch.zeros((n, 1, S, S), dtype=torch.uint8, device="cuda")

    def run():
        for i in range(n):
            pre.image(imgs[i % 8], out=x[i])
            pre.mask(masks[i % 8], out=t[i, 0])

    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (reps * n)
    # algorithmic bytes per (image, mask) pair: read the sources once, write the results once
    alg = h * w * 3 + 3 * S * S * 4 + S * S + S * S
    out = {"metric": "input pipeline images/sec (1105x800 RGB -> 512x512 fp32 NCHW + mask)", "value": 1e3 / ms,
           "unit": "images/s", "ms_per_image": ms, "dtype": "u8",
           "roofline": {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": alg / (ms * 1e-3) / 1e9 / 8000.0, "traffic": None},
           "note": "two launches per image (a few hundred KB each): launch-latency bound, not HBM bound"}
    try:
        from PIL import Image
        a = imgs[0].cpu().numpy()
        pil = Image.fromarray(a)
        mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
        std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
        torch.set_num_threads(1)
        t0 = time.time()
        k = 0
        while time.time() - t0 < 5.0:
            r = np.asarray(pil.resize((S, S), Image.BILINEAR))
            torch.from_numpy(r.copy()).permute(2, 0, 1).contiguous().float().div(255).sub_(mean).div_(std)
            k += 1
        out["cpu_baseline"] = {"value": k / (time.time() - t0), "unit": "images/s", "cores": 1, "kind": "reference",
                               "sample": "Pillow resize + ToTensor + Normalize of one 1105x800 image for 5 s (decode excluded on both sides)"}
    except ImportError:
        out["cpu_baseline"] = None
    print(json.dumps(out))


if __name__ == "__main__":
    main()
