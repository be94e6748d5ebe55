#!/usr/bin/env python3
"""Inference throughput on one MI355X (BASELINE.json configs[4] shape and the 512x512 batch):
  * eval forward, batch 64 @512x512 bf16, BatchNorm folded vs not
  * 2048x2048 images through TiledSegmenter (core 512 + 2*96 halo) incl. mask + IoU counts
Synthetic inputs resident in HBM.  Prints one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fea_amd  # noqa: E402
from fea_amd.inference import TiledSegmenter  # noqa: E402

GF512 = 385.51      # forward GFLOP per 512x512 image


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    torch.manual_seed(0)
    m = fea_amd.FloorPlanSegmentationModel(compute_dtype="bf16").cuda().eval()
    eng = m._engine()
    x = torch.randn(64, 3, 512, 512, device="cuda")
    out = {}
    with torch.no_grad():
        for fold in (True, False):
            eng.fold_bn = fold
            ms = timed(lambda: eng.forward(x, False), 5)
            out["fwd512_b64_" + ("folded" if fold else "bn_kernels")] = {
                "ms": ms, "images_per_s": 64e3 / ms, "tflops": 64 * GF512 / ms}
        eng.fold_bn = True
        del x
        NB = 4
        big = torch.randn(NB, 3, 2048, 2048, device="cuda")
        tgt = (torch.rand(NB, 6, 2048, 2048, device="cuda") < 0.1).to(torch.uint8)
        from fea_amd.inference import window_plan
        for core in (512, 1024, (1024, 2048)):
            seg = TiledSegmenter(m, core=core)          # windows per forward: the default pixel budget
            ms = timed(lambda: seg.segment(big, tgt), 3)
            wh, ww, plan = window_plan(2048, 2048, core)
            out[f"tiled2048_core{core}"] = {"ms_per_image": ms / NB, "images_per_s": NB * 1e3 / ms, "useful_tflops": NB * 16 * GF512 / ms,
                                            "images_per_call": NB, "windows_per_forward": max(1, seg.PIXELS_PER_FORWARD // (wh * ww)),
                                            "windows_per_image": len(plan), "window": [wh, ww],
                                            "computed_over_useful_pixels": len(plan) * wh * ww / 2048.0 ** 2}
    print(json.dumps({"metric": "UNet inference throughput bf16", "data": "synthetic", **out}))


if __name__ == "__main__":
    main()
