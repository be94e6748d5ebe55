"""UNet training step on the HIP path, mirroring the loop shape of the
reference's DirectTrainer.train_epoch (data_model/train_direct.py:59-125):
zero_grad -> forward -> loss -> backward -> clip_grad_norm_(1.0) -> AdamW step.

The reference trains the whole FloorPlanTo3DPipeline on self-supervised
regularisers and has no segmentation loss; this trainer optimises the UNet
alone on the build-defined BCE+Dice loss.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.distributed as dist

from . import _lib as L
from . import functional as Fn
from .engine import param_order


class GradBuckets:
    """Bucketed gradient all-reduce (RCCL via torch.distributed) overlapped
    with backward.  The engine's flat gradient buffer is laid out in the order
    backward finishes gradients, so a bucket is a contiguous slice that is
    complete once its last parameter is reported ready; it is then reduced on
    a side stream while the main stream keeps computing."""

    def __init__(self, engine, bucket_mb: float = 32.0, group=None):
        self.eng = engine
        self.group = group
        self.world = dist.get_world_size(group)
        engine._ensure_flat_grad()
        order = param_order(engine.m)
        limit = int(bucket_mb * (1 << 20) / 4)
        self.buckets = []          # (start, end, last_param_name)
        start = 0
        cur_end = 0
        for n in order:
            off, numel = engine.grad_offsets[n]
            end = off + (numel + 63) // 64 * 64
            cur_end = end
            if cur_end - start >= limit:
                self.buckets.append((start, cur_end, n))
                start = cur_end
        if cur_end > start:
            self.buckets.append((start, cur_end, order[-1]))
        self.trigger = {last: i for i, (_, _, last) in enumerate(self.buckets)}
        self.on_gpu = engine.flat_grad.is_cuda
        self.stream = torch.cuda.Stream() if self.on_gpu else None
        self.pending = []
        engine.grad_ready = self.on_ready

    def _reduce(self, s, e):
        sl = self.eng.flat_grad[s:e]
        dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group)
        sl.mul_(1.0 / self.world)

    def on_ready(self, name: str):
        i = self.trigger.get(name)
        if i is None:
            return
        s, e, _ = self.buckets[i]
        if not self.on_gpu:                      # gloo / CPU tensors: synchronous
            self._reduce(s, e)
            return
        ev = torch.cuda.Event()
        ev.record()                              # gradients of this bucket are enqueued up to here
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            self._reduce(s, e)
        done = torch.cuda.Event()
        done.record(self.stream)
        self.pending.append(done)

    def wait(self):
        for ev in self.pending:
            torch.cuda.current_stream().wait_event(ev)
        self.pending.clear()


class UNetTrainer:
    """config keys follow train_direct.py:166-185 where they apply:
    learning_rate, weight_decay, num_epochs (cosine T_max), max_grad_norm."""

    def __init__(self, model, config: Optional[dict] = None, distributed: bool = False, bucket_mb: float = 32.0):
        cfg = dict(learning_rate=1e-4, weight_decay=1e-5, num_epochs=50, max_grad_norm=1.0)
        cfg.update(config or {})
        self.cfg = cfg
        self.model = model
        self.eng = model._engine()
        self.params = list(model.parameters())
        # AdamW(lr, weight_decay) + CosineAnnealingLR(T_max=epochs, eta_min=1e-6): train_direct.py:28-39
        self.opt = torch.optim.AdamW(self.params, lr=cfg["learning_rate"], weight_decay=cfg["weight_decay"], fused=True)
        self.sched = torch.optim.lr_scheduler.CosineAnnealingLR(self.opt, T_max=cfg["num_epochs"], eta_min=1e-6)
        self.loss_ws = {}
        self.buckets = None
        self.distributed = distributed
        self._grads_bound = False

    def _bind_grads(self):
        self.eng._ensure_flat_grad()
        names = [n for n, _ in self.model.named_parameters()]
        for n, p in zip(names, self.params):
            p.grad = self.eng.grad_views[n]
        if self.distributed and self.buckets is None:
            self.buckets = GradBuckets(self.eng)
        self._grads_bound = True

    def step(self, x: torch.Tensor, target_u8: torch.Tensor):
        """One optimisation step; returns the loss as a 1-element device tensor."""
        m = self.model
        m.train()
        logits = self.eng.forward(x, True)
        loss, dl = Fn.bce_dice_loss_and_grad(logits, target_u8, ws=self.loss_ws)
        if not self._grads_bound:
            self.eng._ensure_device(x)
            self._bind_grads()
        self.eng.backward(dl)                     # writes p.grad (views of the flat buffer)
        if self.buckets is not None:
            self.buckets.wait()
        # clip_grad_norm_(params, 1.0): train_direct.py:105 -- on the flat buffer (padding is zero)
        flat = self.eng.flat_grad
        total = torch.linalg.vector_norm(flat)
        flat.mul_(torch.clamp(self.cfg["max_grad_norm"] / (total + 1e-6), max=1.0))
        self.opt.step()
        self.last_grad_norm = total
        return loss

    def epoch_end(self):
        self.sched.step()
