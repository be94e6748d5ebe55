"""UNet training step on the HIP path, mirroring the loop shape of the
reference's DirectTrainer.train_epoch (data_model/train_direct.py:59-125):
zero_grad -> forward -> loss -> backward -> clip_grad_norm_(1.0) -> AdamW step.

The reference trains the whole FloorPlanTo3DPipeline on self-supervised
regularisers and has no segmentation loss; this trainer optimises the UNet
alone on the build-defined BCE+Dice loss.
"""
from __future__ import annotations

import json
import math
import os
import time
from typing import Iterable, Optional, Tuple

import torch
import torch.distributed as dist

from . import _lib as L
from . import trace
from . import functional as Fn
from .engine import param_order
from .optim import ClipAdamW


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
        self.has_avg = self.on_gpu and dist.get_backend(group) == "nccl"      # RCCL; gloo (CPU tests, one-GPU rehearsals) has no AVG
        self.stream = torch.cuda.Stream() if self.on_gpu else None
        self.pending = []
        engine.grad_ready = self.on_ready

    @trace.traced("grad.bucket_allreduce")
    def _reduce(self, s, e):
        sl = self.eng.flat_grad[s:e]
        if self.has_avg:
            # RCCL averages inside the collective: no second pass over the bucket on the communication stream
            dist.all_reduce(sl, op=dist.ReduceOp.AVG, group=self.group)
        elif self.on_gpu:
            # gloo with device gradients (rehearsal of the N-rank path on a box with fewer GPUs than ranks; its
            # device-tensor all-reduce hangs on this ROCm build): through the host, on the communication stream
            h = sl.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            sl.copy_(h.mul_(1.0 / self.world))
        else:                                    # gloo has no AVG
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
    """config keys follow train_direct.py:166-185 where they apply: learning_rate,
    weight_decay, batch_size, num_epochs (cosine T_max), checkpoint_dir, log_dir, save_every;
    plus max_grad_norm (the reference hard-codes 1.0, train_direct.py:105).  Checkpoints
    use the reference's dict schema (train_direct.py:127-135); scalars the reference sends
    to TensorBoard go to `<log_dir>/metrics.jsonl`, one JSON object per step."""

    def __init__(self, model, config: Optional[dict] = None, distributed: bool = False, bucket_mb: float = 32.0):
        cfg = dict(learning_rate=1e-4, weight_decay=1e-5, batch_size=4, num_epochs=50, max_grad_norm=1.0,
                   checkpoint_dir="./checkpoints", log_dir="./logs", save_every=5,
                   optimizer="hip")          # "hip": fea_amd.optim.ClipAdamW (two launches); "torch": torch's fused AdamW
        cfg.update(config or {})
        self.cfg = cfg
        self.model = model
        self.eng = model._engine()
        self.params = list(model.parameters())
        # AdamW(lr, weight_decay) + CosineAnnealingLR(T_max=epochs, eta_min=1e-6): train_direct.py:28-39
        if cfg["optimizer"] == "hip":
            self.opt = ClipAdamW(list(model.named_parameters()), self.eng, lr=cfg["learning_rate"],
                                 weight_decay=cfg["weight_decay"], max_grad_norm=cfg["max_grad_norm"])
        elif cfg["optimizer"] == "torch":
            self.opt = torch.optim.AdamW(self.params, lr=cfg["learning_rate"], weight_decay=cfg["weight_decay"], fused=True)
        else:
            raise ValueError(f"optimizer must be 'hip' or 'torch', got {cfg['optimizer']!r}")
        self.sched = torch.optim.lr_scheduler.CosineAnnealingLR(self.opt, T_max=cfg["num_epochs"], eta_min=1e-6)
        self.loss_ws = {}
        # Fused head + loss kernels (fea_head_loss_sums / _bwd: logits and dlogits stay in registers, 2 GB less HBM
        # traffic per batch-64 step).  On by default where their matrix-pipe form applies (bf16, H*W % 16 == 0:
        # 1.7 ms against 2.35 ms for the four separate kernels); the 8-lane VALU form of the other cases is slower than
        # the separate kernels (3.05 ms) and stays opt-in.  FEA_AMD_FUSE_HEAD_LOSS=0 / 1 forces either.
        self.fuse_head_loss = os.environ.get("FEA_AMD_FUSE_HEAD_LOSS")
        self.buckets = None
        self.distributed = distributed
        self._grads_bound = False
        if (distributed and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
                and not os.environ.get("FEA_AMD_PERSISTENT_SPLIT")):
            # The all-reduce's channel kernels take CUs while backward runs.  A persistent kernel (one workgroup per CU,
            # static tile ranges) whose workgroup finds its CU taken waits for a peer to finish: 1.6-1.8x per launch with
            # 8-32 CUs held (scripts/cu_steal_test.py).  Four workgroups per CU's worth of tiles: 1.02-1.14x, for
            # +0.6 % on an undisturbed step.  Set before the first forward: table and workspace sizes depend on it.
            # (World size 1 keeps the single-GPU setting and its bit-identical results.)
            L.call("fea_set_persistent_split", 4)
            self.eng.plans.clear()
            self.eng._p_rows_cache.clear()

    def _bind_grads(self):
        self.eng._ensure_flat_grad()
        names = [n for n, _ in self.model.named_parameters()]
        for n, p in zip(names, self.params):
            p.grad = self.eng.grad_views[n]
        self._bound = [(p, self.eng.grad_views[n].data_ptr()) for n, p in zip(names, self.params)]
        if self.distributed and self.buckets is None:
            self.buckets = GradBuckets(self.eng)
        self._grads_bound = True

    def _grads_still_bound(self) -> bool:
        """False after `opt.zero_grad()` / `model.zero_grad()` (set_to_none=True is torch's default) or any other
        re-assignment of p.grad: the step would then run forward and backward and update nothing."""
        return all(p.grad is not None and p.grad.data_ptr() == ptr for p, ptr in self._bound)

    @trace.traced("train.step")
    def step(self, x: torch.Tensor, target_u8: torch.Tensor):
        """One optimisation step; returns the loss as a 1-element device tensor."""
        m = self.model
        m.train()
        fused = self.eng.fused_loss_supported(x.shape[2], x.shape[3]) and (
            self.fuse_head_loss == "1" if self.fuse_head_loss in ("0", "1")
            else bool(self.fuse_head_loss) if isinstance(self.fuse_head_loss, bool)
            else (self.eng.dtype == torch.bfloat16 and (x.shape[2] * x.shape[3]) % 16 == 0))
        logits = self.eng.forward(x, True, head=not fused)
        if not self._grads_bound or not self._grads_still_bound():
            self.eng._ensure_device(x)
            self._bind_grads()
        if isinstance(self.opt, ClipAdamW):
            # this step's backward is the one the optimiser steps on, whatever else ran on the engine since the last
            # step (a gradient check, a module-path backward)
            self.opt._bw_seen = self.eng.backward_count
        if fused:
            # head + BCE/Dice + their backward fused: the logits never go to HBM (writes p.grad: views of the flat buffer)
            self.eng.backward(None, target=target_u8)
            loss = self.eng._last[0].loss.clone()      # the plan's scalar is rewritten by the next step
        else:
            loss, dl = Fn.bce_dice_loss_and_grad(logits, target_u8, ws=self.loss_ws)
            loss = loss.clone()
            self.eng.backward(dl)
        if self.buckets is not None:
            self.buckets.wait()
        if isinstance(self.opt, ClipAdamW):
            # clip_grad_norm_ + AdamW in two launches (train_direct.py:105-106)
            self.last_grad_norm = self.opt.step()
            return loss
        # clip_grad_norm_(params, 1.0): train_direct.py:105 -- on the flat buffer (padding is zero)
        flat = self.eng.flat_grad
        total = torch.linalg.vector_norm(flat)
        flat.mul_(torch.clamp(self.cfg["max_grad_norm"] / (total + 1e-6), max=1.0))
        self.opt.step()
        self.last_grad_norm = total
        return loss

    def epoch_end(self):
        self.sched.step()

    # ---- epoch loop, checkpoints, metrics (train_direct.py:59-163) ----------------------
    def _is_rank0(self) -> bool:
        return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0

    def train_epoch(self, epoch: int, loader: Iterable[Tuple[torch.Tensor, torch.Tensor]], log=None) -> float:
        """One pass over `loader` (batches of (image [N,3,H,W] float, target [N,classes,H,W]
        uint8), already on the device or not).  Returns the mean loss (train_direct.py:125).
        The loss is read back once per epoch, not per step (no sync in the step loop)."""
        dev = next(self.model.parameters()).device
        losses = []
        t0 = time.time()
        n_img = 0
        for batch_idx, (x, t) in enumerate(loader):
            x = x.to(dev, non_blocking=True)
            t = t.to(dev, non_blocking=True)
            losses.append(self.step(x, t))
            n_img += x.shape[0]
        if not losses:
            return float("nan")
        vals = torch.stack([l.reshape(()) for l in losses]).cpu()        # one sync
        dt = time.time() - t0
        if log is not None and self._is_rank0():
            for batch_idx, v in enumerate(vals.tolist()):
                log.write(json.dumps({"epoch": epoch, "global_step": epoch * len(losses) + batch_idx,
                                      "Loss/total": v}) + "\n")
            log.write(json.dumps({"epoch": epoch, "Loss/epoch_mean": vals.mean().item(), "lr": self.sched.get_last_lr()[0],
                                  "images_per_sec": n_img / dt if dt > 0 else None}) + "\n")
            log.flush()
        return vals.mean().item()

    def save_checkpoint(self, epoch: int, filepath: str):
        """Same dict schema as train_direct.py:127-135."""
        torch.cuda.synchronize()
        torch.save({
            "epoch": epoch,
            "model_state_dict": self.model.state_dict(),
            "optimizer_state_dict": self.opt.state_dict(),
            "scheduler_state_dict": self.sched.state_dict(),
        }, filepath)

    def load_checkpoint(self, filepath: str) -> int:
        """Resume from a checkpoint written by save_checkpoint (or by the reference's
        trainer for the model part: `module.` / `segmentation_model.` prefixes are accepted).
        Returns the epoch stored in it."""
        ckpt = torch.load(filepath, map_location="cpu", weights_only=True)
        self.model.load_state_dict(ckpt["model_state_dict"], strict=False)
        self.eng.invalidate()
        try:
            self.opt.load_state_dict(ckpt["optimizer_state_dict"])
            self.sched.load_state_dict(ckpt["scheduler_state_dict"])
        except (KeyError, ValueError):
            pass            # a pipeline checkpoint of the reference: its optimizer covers other modules too
        return int(ckpt.get("epoch", -1))

    def train(self, make_loader, start_epoch: int = 0):
        """The loop of DirectTrainer.train (train_direct.py:137-163).  `make_loader(epoch)`
        returns the epoch's iterable of batches."""
        cfg = self.cfg
        rank0 = self._is_rank0()
        log = None
        if rank0:
            os.makedirs(cfg["checkpoint_dir"], exist_ok=True)
            os.makedirs(cfg["log_dir"], exist_ok=True)
            log = open(os.path.join(cfg["log_dir"], "metrics.jsonl"), "a")
        try:
            for epoch in range(start_epoch, cfg["num_epochs"]):
                self.train_epoch(epoch, make_loader(epoch), log)
                self.sched.step()
                if rank0 and (epoch + 1) % cfg["save_every"] == 0:
                    self.save_checkpoint(epoch, os.path.join(cfg["checkpoint_dir"], f"checkpoint_epoch_{epoch}.pth"))
            if rank0:
                self.save_checkpoint(cfg["num_epochs"] - 1, os.path.join(cfg["checkpoint_dir"], "final_model.pth"))
        finally:
            if log is not None:
                log.close()
