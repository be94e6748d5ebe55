#!/usr/bin/env python3
"""Test infrastructure (not product code): decodes a few of the reference's own (floor plan, mask) pairs into a small
fixture, tests/golden/real_pairs.npz.

    python oracle/make_real_pairs.py            # needs /root/reference (development container only)

The pairs are `floor_plans2/<name>` <-> `dataset_masks/<name>` (the only realistic (image, mask) pairs;
custom_dataset.py:65-90 is the reference loader, which opens images with `Image.open(path).convert('RGB')`).  The fixture
holds DATA only: the decoded RGB arrays and grey masks of the six pairs with the smallest floor-plan images.
"""
import os
import sys

import numpy as np
from PIL import Image

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "real_pairs.npz")
NPAIRS = 6


def main():
    fp, mk = os.path.join(REF, "floor_plans2"), os.path.join(REF, "dataset_masks")
    names = sorted(set(os.listdir(fp)) & set(os.listdir(mk)))
    sized = []
    for n in names:
        with Image.open(os.path.join(fp, n)) as im:
            sized.append((im.size[0] * im.size[1], n))
    sized.sort()
    arrays = {}
    picked = [n for _, n in sized[:NPAIRS]]
    for i, n in enumerate(picked):
        with Image.open(os.path.join(fp, n)) as im:
            arrays[f"image_{i}"] = np.asarray(im.convert("RGB")).copy()          # custom_dataset.py:68
        with Image.open(os.path.join(mk, n)) as im:
            arrays[f"mask_{i}"] = np.asarray(im.convert("L")).copy()
        print(n, arrays[f"image_{i}"].shape, arrays[f"mask_{i}"].shape, f"fg {(arrays[f'mask_{i}'] > 127).mean():.3f}")
    arrays["names"] = np.array(picked)
    np.savez_compressed(OUT, **arrays)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    sys.exit(main())
