"""Real-data end-to-end check on the reference's own (floor plan, mask) pairs (class-count note:
`num_classes=1` on floor_plans2/ <-> dataset_masks/; custom_dataset.py:23-28,65-90 is the reference loader).

Fixture: tests/golden/real_pairs.npz = six decoded pairs of the reference's data (oracle/make_real_pairs.py; data only).
They are written back as lossless PNGs under the reference's directory names, so that the product's loader
(`FloorPlanPairs`: PIL decode + `convert('RGB')`, GPU resize / normalise / binarise) runs as it would on the real tree.

  input pipeline  -> bit-exact against the CPU restatement of Resize -> ToTensor -> Normalize and `> 127`;
  5 UNetTrainer steps (fp32, num_classes = 1, BCE + Dice, clip 1.0, AdamW) -> loss per step within 2e-3 of the oracle;
  final eval-mode mask + IoU counts -> exact for the HIP logits, and the masks agree with the oracle's.
"""
import os

import numpy as np
import pytest
import torch

import fea_amd
from fea_amd import functional as Fn
from fea_amd.data import FloorPlanPairs
from fea_amd.train import UNetTrainer
from oracle import preprocess_oracle as P
from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu
FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "real_pairs.npz")
SIZE = 128


def _write_tree(tmp_path):
    from PIL import Image
    data = np.load(FIX)
    fp, mk = tmp_path / "floor_plans2", tmp_path / "dataset_masks"
    fp.mkdir()
    mk.mkdir()
    names = []
    for i, n in enumerate(data["names"]):
        name = os.path.splitext(str(n))[0] + ".png"
        Image.fromarray(data[f"image_{i}"]).save(fp / name)
        Image.fromarray(data[f"mask_{i}"]).save(mk / name)
        names.append(name)
    order = sorted(range(len(names)), key=lambda i: names[i])          # FloorPlanPairs sorts by file name
    return str(fp), str(mk), [data[f"image_{i}"] for i in order], [data[f"mask_{i}"] for i in order]


def test_real_pairs_pipeline_training_and_iou(tmp_path):
    fp, mk, images, masks = _write_tree(tmp_path)
    pairs = FloorPlanPairs(fp, mk, img_size=SIZE, num_classes=1)
    assert len(pairs) == 6
    (x, t), = list(pairs.batches(6, shuffle=False))
    assert x.shape == (6, 3, SIZE, SIZE) and t.shape == (6, 1, SIZE, SIZE) and t.dtype == torch.uint8
    # input pipeline: bit-exact against the oracle of the reference's transform on the same decoded arrays
    x_ref = torch.from_numpy(np.stack([P.transform(im, SIZE) for im in images]))
    t_ref = torch.from_numpy(np.stack([P.mask_target(m, SIZE) for m in masks])).unsqueeze(1)
    assert torch.equal(x.cpu(), x_ref) and torch.equal(t.cpu(), t_ref)
    fg = t_ref.float().mean().item()
    assert 0.05 < fg < 0.3, fg                                         # real masks: 8-19 % foreground

    torch.manual_seed(31)
    m = fea_amd.FloorPlanSegmentationModel(num_classes=1, compute_dtype="fp32").cuda()
    params, bufs = O.split_state({k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    lr = 1e-3
    tr = UNetTrainer(m, dict(learning_rate=lr))
    state = {}
    hip, ref = [], []
    for step in range(5):
        loss = tr.step(x, t)
        lo, _, grads, nb = O.loss_and_grads(params, bufs, x_ref, t_ref, training=True)
        clipped, norm = O.clip_grad_norm(grads, 1.0)
        params = O.adamw_step(params, clipped, state, lr=lr)
        bufs = nb
        hip.append(loss.item())
        ref.append(lo.item())
        assert abs(hip[-1] - ref[-1]) < 2e-3 * abs(ref[-1]), (step, hip, ref)
        assert abs(tr.last_grad_norm.item() - norm.item()) < 5e-2 * norm.item(), (step, tr.last_grad_norm.item(), norm.item())
    assert hip[-1] < hip[0] and ref[-1] < ref[0], (hip, ref)

    m.eval()
    with torch.no_grad():
        logits = m(x)
    mask, inter, union = Fn.mask_iou_counts(logits, t)
    pm, pi, pu = O.mask_iou_counts(logits.cpu(), t_ref)                # exact for the HIP logits (integer work)
    assert torch.equal(mask.cpu(), pm) and torch.equal(inter.cpu(), pi) and torch.equal(union.cpu(), pu)
    lo_e, _ = O.forward(params, bufs, x_ref, False)
    agree = ((logits.cpu() > 0) == (lo_e > 0)).float().mean().item()
    assert agree > 0.98, agree
    _, miou_h = O.iou_from_counts(pi, pu)
    _, ri, ru = O.mask_iou_counts(lo_e, t_ref)
    assert abs(float(miou_h) - float(O.iou_from_counts(ri, ru)[1])) < 5e-2
