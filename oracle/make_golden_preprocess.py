"""Generate tests/golden/preprocess_golden.npz with Pillow itself (the library the reference's
transform runs, custom_dataset.py:23-28,69-71).  Run in the development container only:

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_preprocess.py

Each case stores the uint8 input and Pillow's `resize(..., BILINEAR)` output; two cases are
crops of the reference's own data files (floor_plans2/ are GIF palette images behind a .jpg
name: `convert('RGB')` as custom_dataset.py:68 does; dataset_masks/ are 8-bit grey JPEGs).
One case also stores the float32 result of ToTensor + Normalize computed with torch.
"""
import os
import sys

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden", "preprocess_golden.npz")
REF = "/root/reference"

CASES = [  # name, in_h, in_w, channels, out_h, out_w
    ("down_rgb", 97, 131, 3, 64, 48),
    ("up_rgb", 40, 40, 3, 64, 64),
    ("mixed_grey", 33, 77, 1, 33, 50),
    ("tall_rgb", 100, 60, 3, 16, 60),
    ("tiny_up_rgb", 5, 7, 3, 32, 32),
    ("big_down_grey", 300, 211, 1, 32, 32),
]


def pil_resize(a, out_h, out_w):
    return np.asarray(Image.fromarray(a).resize((out_w, out_h), Image.BILINEAR))


def main():
    rng = np.random.default_rng(20240607)
    out = {}
    names = []
    for name, h, w, c, oh, ow in CASES:
        a = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        if c == 1:
            a = a[:, :, 0]
        out[name + "/in"] = a
        out[name + "/out"] = pil_resize(a, oh, ow)
        names.append(name)
    fp = sorted(os.listdir(os.path.join(REF, "floor_plans2")))[0]
    img = np.asarray(Image.open(os.path.join(REF, "floor_plans2", fp)).convert("RGB"))
    crop = np.ascontiguousarray(img[200:360, 300:500])            # 160 x 200 crop of a real plan
    out["plan_crop/in"] = crop
    out["plan_crop/out"] = pil_resize(crop, 96, 96)
    names.append("plan_crop")
    t = torch.from_numpy(out["plan_crop/out"].copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    out["plan_crop/normalized"] = t.sub(mean).div(std).numpy()
    mk = np.asarray(Image.open(os.path.join(REF, "dataset_masks", fp)).convert("L"))
    crop = np.ascontiguousarray(mk[100:260, 150:310])
    out["mask_crop/in"] = crop
    out["mask_crop/out"] = pil_resize(crop, 64, 64)
    names.append("mask_crop")
    out["names"] = np.array(names)
    out["pillow_version"] = np.array(Image.__version__ if hasattr(Image, "__version__") else "?")
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    sys.exit(main())
