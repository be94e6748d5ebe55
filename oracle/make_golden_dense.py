"""Generate tests/golden/unet_dense.npz from the REFERENCE module: FULL tensors, not digests.

Runs only in the development container (it imports /root/reference/data_model/models.py, which never travels to
the GPU box).  tests/golden/unet_golden.json pins logits, gradients and parameters through digests (sum / abs-sum / L2 /
8-16 strided samples); this file adds, for the strict case `train_32` (the well-conditioned seed recorded there), the
reference's own values element by element:

  * logits of the first training step and eval-mode logits after it, whole tensors;
  * the gradient of every parameter after the first backward -- whole tensors up to 65,536 elements (every bias and
    BatchNorm vector, enc1.*, dec1.3, upconv1, out), and for the larger weights the first 4,096 elements plus 4,096
    evenly strided ones;
  * BatchNorm running statistics after the step, whole;
  * the parameters after clip_grad_norm_(1.0) + AdamW, the same element sets.

Data only.  tests/test_oracle_golden.py checks oracle/unet_oracle.py against it on the CPU,
tests/test_unet_gpu.py::test_against_reference_dense_tensors checks the HIP fp32 path.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_dense.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/data_model")

from models import FloorPlanSegmentationModel  # noqa: E402  (the reference)
from oracle import unet_oracle as O  # noqa: E402

WHOLE = 65536
HEAD = 4096


def element_set(numel: int) -> np.ndarray:
    """Flat indices kept for a tensor of `numel` elements (all of them up to WHOLE)."""
    if numel <= WHOLE:
        return np.arange(numel, dtype=np.int64)
    stride = numel // HEAD
    return np.unique(np.concatenate([np.arange(HEAD, dtype=np.int64), np.arange(HEAD, dtype=np.int64) * stride]))


def take(t: torch.Tensor) -> np.ndarray:
    f = t.detach().reshape(-1).to(torch.float32).numpy()
    return f[element_set(f.size)].copy()


def main():
    torch.set_num_threads(8)
    with open(os.path.join(ROOT, "tests", "golden", "unet_golden.json")) as f:
        case = json.load(f)["train_32"]
    seed, n, h, w = case["seed"], case["n"], case["h"], case["w"]
    torch.manual_seed(seed)
    model = FloorPlanSegmentationModel(in_channels=case["in_channels"], num_classes=case["num_classes"]).train()
    x, tgt = O.synthetic_batch(n, h, w, num_classes=case["num_classes"], seed=seed + 100, in_channels=case["in_channels"])
    x = torch.randn(n, case["in_channels"], h, w, generator=torch.Generator().manual_seed(seed + 200))      # kind "randn"
    out = {"meta": np.array(json.dumps(dict(case="train_32", seed=seed, n=n, h=h, w=w, whole=WHOLE, head=HEAD,
                                            torch=torch.__version__)))}
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
    opt.zero_grad()
    logits = model(x)
    loss = O.bce_dice_loss(logits, tgt)
    loss.backward()
    assert abs(float(loss) - case["train"][0]["loss"]) < 1e-6, "not the case the digests were made from"
    out["logits"] = logits.detach().numpy()
    out["loss"] = np.array(float(loss), dtype=np.float64)
    for k, p in model.named_parameters():
        out["grad/" + k] = take(p.grad)
    for k, b in model.named_buffers():
        out["buffer/" + k] = take(b.float())
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    for k, p in model.named_parameters():
        out["after/" + k] = take(p)
    model.eval()
    with torch.no_grad():
        out["eval_logits"] = model(x).numpy()
    path = os.path.join(ROOT, "tests", "golden", "unet_dense.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
