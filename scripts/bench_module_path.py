#!/usr/bin/env python3
"""The reference-shaped training loop on the drop-in module (train_direct.py:96-106): model(x) -> loss -> loss.backward() ->
clip_grad_norm_ -> optimizer.step(), against UNetTrainer.step on the same batch.  Batch 64 @512x512 bf16, synthetic."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fea_amd
from fea_amd import functional as Fn
from fea_amd.data import synthetic_batch
from fea_amd.train import UNetTrainer
from fea_amd.optim import ClipAdamW

def timed(fn, steps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

x, t = synthetic_batch(64, 512, 512, seed=1234)
x, t = x.cuda(), t.cuda()
out = {}
torch.manual_seed(0)
m = fea_amd.FloorPlanSegmentationModel(compute_dtype="bf16").cuda().train()
for name, opt in (("torch AdamW(fused) + clip_grad_norm_", torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-5, fused=True)),
                  ("ClipAdamW", ClipAdamW(list(m.named_parameters()), m._engine(), lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0))):
    def step():
        opt.zero_grad(set_to_none=False)
        loss = Fn.bce_dice_loss(m(x), t)
        loss.backward()
        if "torch" in name: torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
    out["module path, " + name] = timed(step)
torch.manual_seed(0)
m2 = fea_amd.FloorPlanSegmentationModel(compute_dtype="bf16").cuda()
tr = UNetTrainer(m2)
out["UNetTrainer.step"] = timed(lambda: tr.step(x, t))
print(json.dumps({k: round(v, 2) for k, v in out.items()}))
