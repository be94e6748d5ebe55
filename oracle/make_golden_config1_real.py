"""BASELINE configs[0] on a REAL shipped floor plan: reference UNet (data_model/models.py), fp32, eval, batch 1, 256x256,
input = one file of the reference's floor_plans2/ through the reference's transform (custom_dataset.py:23-28,65-71:
convert('RGB') -> Resize -> ToTensor -> Normalize, restated with Pillow + torch because torchvision is absent).

Runs only in the development container (imports /root/reference/data_model/models.py).  Writes
  tests/golden/config1_real.npz   the decoded uint8 RGB image (data) and Pillow's 256x256 resize of it
  tests/golden/config1_real.json  digests of the normalised input and of the reference's logits, seed-0 weights

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_config1_real.py
"""
import json
import os
import sys

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/data_model")

from models import FloorPlanSegmentationModel  # noqa: E402  (the reference)
from oracle import unet_oracle as O  # noqa: E402

NAME = "135396gra_floorplan_1.jpg"          # 525 x 800 GIF palette image behind a .jpg name
SIZE = 256


def main():
    img = np.asarray(Image.open(os.path.join("/root/reference/floor_plans2", NAME)).convert("RGB"))
    small = np.asarray(Image.fromarray(img).resize((SIZE, SIZE), Image.BILINEAR))
    t = torch.from_numpy(small.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    x = t.sub(mean).div(std).unsqueeze(0)
    torch.manual_seed(0)
    model = FloorPlanSegmentationModel().eval()
    torch.set_num_threads(8)
    with torch.no_grad():
        y = model(x)
    out = os.path.join(ROOT, "tests", "golden")
    np.savez_compressed(os.path.join(out, "config1_real.npz"), image=img, resized=small)
    rec = dict(name="config1_real_256", source="floor_plans2/" + NAME, seed=0, h=SIZE, w=SIZE, image_shape=list(img.shape),
               pillow=Image.__version__, torch=torch.__version__, x=O.digest(x, 64), logits=O.digest(y, 64),
               pred_sum=int((y > 0).sum()))
    with open(os.path.join(out, "config1_real.json"), "w") as f:
        json.dump(rec, f, indent=0)
    print("wrote config1_real.npz", os.path.getsize(os.path.join(out, "config1_real.npz")), "bytes; pred_sum", rec["pred_sum"])


if __name__ == "__main__":
    main()
