"""Generate tests/golden/*.json|npz from the REFERENCE module.

Runs only in the development container (it imports
/root/reference/data_model/models.py, which never travels to the GPU box).
The fixtures are data: seeded inputs and digests (sum / abs-sum / L2 / strided
samples) of the reference's outputs, gradients, buffers and post-optimizer
parameters.  `tests/test_oracle_golden.py` checks oracle/unet_oracle.py
against them.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
"""
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/data_model")

from models import FloorPlanSegmentationModel  # noqa: E402  (the reference)
from oracle import unet_oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def dig(t, n=16):
    return O.digest(t, n)


def make_inputs(kind, seed, n, h, w, num_classes, in_channels):
    """'plan' = floor-plan-shaped synthetic batch (flat regions => max-pool
    ties, so gradients are tie-break sensitive); 'randn' = the same targets
    with a tie-free gaussian image, used where gradients are compared."""
    x, tgt = O.synthetic_batch(n, h, w, num_classes=num_classes, seed=seed + 100, in_channels=in_channels)
    if kind == "randn":
        g = torch.Generator().manual_seed(seed + 200)
        x = torch.randn(n, in_channels, h, w, generator=g)
    return x, tgt


def well_conditioned(seed, n, h, w, num_classes, in_channels, kind):
    """ReLU / max-pool make the gradient piecewise smooth; an activation
    within ~1e-7 of a kink flips under fp32 rounding and moves deep-layer
    gradients by ~1e-3.  Such knife-edge seeds cannot pin an fp32
    restatement to 1e-4, so STRICT gradient goldens (small cases) only use
    seeds where the reference's fp32 and fp64 gradients and the restatement's
    fp32 gradients all agree to 2e-5.  At 64x96x3 and above practically every
    seed has such a flip (100 of 100 tried), so larger cases are recorded
    with strict=False and their gradients are compared loosely; strict
    gradient parity at size is checked in the GPU tests with the kinks
    pinned (oracle `forced=` masks)."""
    x, tgt = make_inputs(kind, seed, n, h, w, num_classes, in_channels)
    gs = []
    for dt in (torch.float32, torch.float64):
        torch.manual_seed(seed)
        m = FloorPlanSegmentationModel(in_channels=in_channels, num_classes=num_classes).to(dt).train()
        loss = O.bce_dice_loss(m(x.to(dt)), tgt)
        loss.backward()
        gs.append([p.grad.double() for k, p in m.named_parameters() if not k.endswith("bias")])
    # the fp32 restatement rounds differently again (explicit BN formula), so
    # it must sit on the same side of every kink too
    sd = O.init_state_dict(seed, in_channels, num_classes)
    params, bufs = O.split_state(sd)
    _, _, og, _ = O.loss_and_grads(params, bufs, x, tgt)
    gs.append([g.double() for k, g in og.items() if not k.endswith("bias")])
    worst = max(((a - b).norm() / b.norm()).item() for a, b in zip(gs[0], gs[1]))
    worst2 = max(((a - b).norm() / b.norm()).item() for a, b in zip(gs[2], gs[1]))
    return max(worst, worst2)


def run_case(name, seed, n, h, w, steps, num_classes=6, in_channels=3, kind="randn", strict=True):
    strict = strict and kind == "randn"
    if strict:
        for s in range(seed, seed + 1000, 10):
            c = well_conditioned(s, n, h, w, num_classes, in_channels, kind)
            print(name, "seed", s, "fp32-vs-fp64 grad rel err", c)
            if c < 2e-5:
                seed = s
                break
        else:
            raise RuntimeError("no well-conditioned seed")
    torch.manual_seed(seed)
    model = FloorPlanSegmentationModel(in_channels=in_channels, num_classes=num_classes)
    model.train()
    x, tgt = make_inputs(kind, seed, n, h, w, num_classes, in_channels)
    case = OrderedDict(name=name, seed=seed, n=n, h=h, w=w, num_classes=num_classes,
                       in_channels=in_channels, steps=steps, kind=kind, strict=strict, torch=torch.__version__)
    case["init"] = OrderedDict((k, dig(v, 8)) for k, v in model.state_dict().items())
    case["x"] = dig(x)
    case["target_sum"] = int(tgt.sum())
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
    case["train"] = []
    for s in range(steps):
        opt.zero_grad()
        logits = model(x)
        loss = O.bce_dice_loss(logits, tgt)
        loss.backward()
        rec = OrderedDict(step=s, loss=float(loss), logits=dig(logits))
        if s == 0:
            rec["grads"] = OrderedDict((k, dig(p.grad, 8)) for k, p in model.named_parameters())
            rec["buffers"] = OrderedDict((k, dig(b.float(), 8)) for k, b in model.named_buffers())
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        rec["grad_norm"] = float(norm)
        opt.step()
        if s in (0, steps - 1):
            rec["params_after"] = OrderedDict((k, dig(p, 8)) for k, p in model.named_parameters())
        case["train"].append(rec)
    model.eval()
    with torch.no_grad():
        le = model(x)
    case["eval_logits"] = dig(le)
    pred, inter, union = O.mask_iou_counts(le, tgt)
    case["eval_iou"] = dict(inter=inter.tolist(), union=union.tolist(), pred_sum=int(pred.sum()))
    return case


def dlogits_case(name, seed, n, h, w):
    """Backward of the UNet alone for a random upstream gradient."""
    torch.manual_seed(seed)
    model = FloorPlanSegmentationModel()
    model.train()
    g = torch.Generator().manual_seed(seed + 7)
    x = torch.randn(n, 3, h, w, generator=g)
    dl = torch.randn(n, 6, h, w, generator=g) / (n * h * w)
    logits = model(x)
    logits.backward(dl)
    return OrderedDict(name=name, seed=seed, n=n, h=h, w=w, x=dig(x), dlogits=dig(dl), logits=dig(logits),
                       grads=OrderedDict((k, dig(p.grad, 8)) for k, p in model.named_parameters()))


def config1_case():
    """BASELINE configs[0]: fp32 eval forward, 256x256, batch 1 (synthetic
    floor-plan image; seed-0 weights)."""
    torch.manual_seed(0)
    model = FloorPlanSegmentationModel().eval()
    x, tgt = O.synthetic_batch(1, 256, 256, seed=4321)
    with torch.no_grad():
        y = model(x)
    return OrderedDict(name="config1_eval_256", seed=0, n=1, h=256, w=256, x=dig(x), logits=dig(y, 64),
                       pred_sum=int((y > 0).sum()))


def main():
    gold = OrderedDict()
    gold["train_32"] = run_case("train_32", seed=0, n=2, h=32, w=32, steps=3)
    gold["train_64"] = run_case("train_64", seed=1, n=3, h=64, w=96, steps=2, strict=False)
    gold["plan_64"] = run_case("plan_64", seed=4, n=2, h=64, w=64, steps=1, kind="plan")
    gold["train_c1"] = run_case("train_c1", seed=2, n=1, h=32, w=32, steps=1, num_classes=1, in_channels=1)
    gold["dlogits_32"] = dlogits_case("dlogits_32", seed=3, n=2, h=32, w=32)
    gold["config1"] = config1_case()
    with open(os.path.join(OUT, "unet_golden.json"), "w") as f:
        json.dump(gold, f, indent=0)
    print("wrote", os.path.join(OUT, "unet_golden.json"), os.path.getsize(os.path.join(OUT, "unet_golden.json")), "bytes")


if __name__ == "__main__":
    main()
