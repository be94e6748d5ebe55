"""clip_grad_norm_ + AdamW of the training loop (reference train_direct.py:28-32 `optim.AdamW(params,
lr, weight_decay)`, :105 `clip_grad_norm_(params, max_norm=1.0)`, :106 `optimizer.step()`) as two HIP
launches over the engine's flat gradient buffer (`fea_grad_sumsq`, `fea_adamw_clip_step`).

`ClipAdamW` is a `torch.optim.Optimizer`: LR schedulers, `param_groups`, `state_dict()` /
`load_state_dict()` behave as with `torch.optim.AdamW` and its checkpoints are interchangeable with
it (same parameter order, same per-parameter state keys `step`, `exp_avg`, `exp_avg_sq`).  The
moments live in two flat buffers laid out like the gradient buffer; the per-parameter state tensors
are views of them.

Which gradients a step uses (torch.optim semantics): `p.grad` of every parameter.  Where `p.grad`
IS the engine's slice of the flat buffer (UNetTrainer binds them so) nothing is copied; a `p.grad`
that autograd accumulated elsewhere (the drop-in module called once per floor before one
`loss.backward()`, models.py:190, or gradient accumulation) is copied into the flat buffer first.
Parameters with `requires_grad=False` or `grad is None` are skipped: no decay, no moment update, not
in the clip norm -- as torch.optim.AdamW / clip_grad_norm_ do.
"""
from __future__ import annotations

from typing import Iterable, Tuple

import torch

from . import _lib as L
from . import trace


class ClipAdamW(torch.optim.Optimizer):
    def __init__(self, named_params: Iterable[Tuple[str, torch.nn.Parameter]], engine, lr=1e-4, betas=(0.9, 0.999),
                 eps=1e-8, weight_decay=1e-5, max_grad_norm=1.0):
        named = list(named_params)
        super().__init__([p for _, p in named], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._names = [n for n, _ in named]
        self._eng = engine
        self.max_grad_norm = float(max_grad_norm) if max_grad_norm else 0.0
        self._t = 0
        self._ptrs = None
        self.sumsq = None
        self._active = None          # which parameters the segment table covers
        self._bw_seen = 0

    # ------------------------------------------------------------------ buffers
    def _params(self):
        return [p for g in self.param_groups for p in g["params"]]

    def _setup(self):
        eng = self._eng
        eng._ensure_flat_grad()
        flat = eng.flat_grad
        if not flat.is_cuda:
            raise L.FeaError("ClipAdamW runs on a ROCm device only (no CPU fallback)")
        params = self._params()
        old = {p: dict(self.state[p]) for p in params if p in self.state and "exp_avg" in self.state[p]}
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=flat.device)
        self.part = torch.empty(max(L.call("fea_sumsq_blocks", flat.numel()), 1), dtype=torch.float32, device=flat.device)
        for name, p in zip(self._names, params):
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise L.FeaError(f"parameter {name} must be a contiguous fp32 tensor")
            off, numel = eng.grad_offsets[name]
            st = self.state[p]
            st["exp_avg"] = self.m[off:off + numel].view(p.shape)
            st["exp_avg_sq"] = self.v[off:off + numel].view(p.shape)
            st["step"] = torch.tensor(float(self._t))
            if p in old:                         # moments loaded before the first step
                st["exp_avg"].copy_(old[p]["exp_avg"])
                st["exp_avg_sq"].copy_(old[p]["exp_avg_sq"])
        self._ptrs = [p.data_ptr() for p in params]
        self._active = None

    def _segments(self, active):
        """Device table of (parameter chunk address, offset in the flat buffers, count) over the active parameters."""
        chunk = L.call("fea_opt_chunk")
        rows = []
        for name, p, on in zip(self._names, self._params(), active):
            if not on:
                continue
            off, numel = self._eng.grad_offsets[name]
            for c0 in range(0, numel, chunk):
                rows.append((p.data_ptr() + 4 * c0, off + c0, min(chunk, numel - c0)))
        self.seg = torch.tensor(rows, dtype=torch.int64).reshape(-1, 3).to(self._eng.flat_grad.device)
        self._active = active

    def _ready(self) -> bool:
        return self._ptrs is not None and self._ptrs == [p.data_ptr() for p in self._params()] \
            and self.m.device == self._eng.flat_grad.device

    # --------------------------------------------------------------------- step
    @torch.no_grad()
    @trace.traced("clip+adamw.step")
    def step(self, closure=None):
        """One clipped AdamW step on the gradients currently in the engine's flat buffer.
        Returns the (unclipped) total gradient norm as a 1-element device tensor."""
        if closure is not None:
            raise L.FeaError("ClipAdamW.step does not take a closure")
        if len(self.param_groups) != 1:
            raise L.FeaError("ClipAdamW supports one parameter group (the reference uses one, train_direct.py:28-32)")
        if not self._ready():
            self._setup()
        grp = self.param_groups[0]
        eng = self._eng
        flat = eng.flat_grad
        # gather this step's gradients into the flat buffer (no copy where p.grad already is its slice)
        active, aliased = [], False
        for name, p in zip(self._names, self._params()):
            view = eng.grad_views[name]
            on = p.requires_grad and p.grad is not None
            active.append(on)
            if not on:
                view.zero_()                     # keeps it out of the clip norm
            elif p.grad.data_ptr() == view.data_ptr():
                aliased = True
            else:
                view.copy_(p.grad)
        if aliased and eng.backward_count - self._bw_seen > 1:
            raise L.FeaError(f"{eng.backward_count - self._bw_seen} backward passes wrote the engine's gradient buffer since the "
                             "last step and p.grad aliases it: only the last one's gradients are there.  Accumulate through "
                             "autograd (do not bind p.grad to engine.grad_views) or step after every backward")
        self._bw_seen = eng.backward_count
        active = tuple(active)
        if active != self._active:
            self._segments(active)
        if self.seg.shape[0] == 0:
            return torch.zeros(1, device=flat.device)
        S = L.stream_ptr()
        self._t += 1
        L.call("fea_grad_sumsq", flat, flat.numel(), self.part, self.sumsq, S)
        L.call("fea_adamw_clip_step", self.seg, self.seg.shape[0], flat, self.m, self.v, self.sumsq,
               self.max_grad_norm, float(grp["lr"]), float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"]),
               float(grp["weight_decay"]), self._t, S)
        return self.sumsq.sqrt().float()

    # --------------------------------------------------------------- checkpoints
    def state_dict(self):
        if self._ptrs is not None:
            for p in self._params():
                self.state[p]["step"] = torch.tensor(float(self._t))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)       # fills self.state with loaded (non-view) tensors
        steps = [float(st["step"]) for st in self.state.values() if "step" in st]
        self._t = int(steps[0]) if steps else 0
        self._ptrs = None                         # next step(): rebuild the flat buffers from the loaded moments
