"""Drop-in replacement for the reference's UNet.

`FloorPlanSegmentationModel` keeps the reference's constructor, forward
signature, sub-module attribute names and `state_dict` layout
(/root/reference/data_model/models.py:6-46), so checkpoints and callers
(`FloorPlanTo3DPipeline`, models.py:183,190) work unchanged.  Parameters live in
ordinary `nn.Conv2d` / `nn.BatchNorm2d` / `nn.ConvTranspose2d` containers
(identical init RNG consumption, fp32 OIHW master weights, optimizer
behaviour); the arithmetic of `forward` and of its backward runs in the HIP
kernels of libfea_unet.so through `fea_amd.engine.UNetEngine`.

There is no CPU execution path: calling the module on a CPU tensor raises.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import _lib as L
from .engine import UNetEngine, named_tensors


def _resolve_dtype(compute_dtype) -> torch.dtype:
    if compute_dtype is None:
        compute_dtype = os.environ.get("FEA_AMD_DTYPE", "fp32")
    if isinstance(compute_dtype, torch.dtype):
        if compute_dtype in (torch.float32, torch.bfloat16):
            return compute_dtype
        raise ValueError(f"unsupported compute dtype {compute_dtype}")
    table = {"fp32": torch.float32, "float32": torch.float32, "f32": torch.float32,
             "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}
    try:
        return table[str(compute_dtype).lower()]
    except KeyError:
        raise ValueError(f"unsupported compute dtype {compute_dtype!r} (use 'fp32' or 'bf16')") from None


class _Lease:
    """Keeps a forward's plan (its saved activations) out of reuse until the backward has
    run or the autograd graph holding it has been dropped."""

    def __init__(self, plan):
        self.plan = plan
        plan.leases += 1

    def release(self):
        if self.plan is not None:
            self.plan.leases -= 1
            self.plan = None

    def __del__(self):
        self.release()


class _UNetFunction(torch.autograd.Function):
    """One autograd node for the whole UNet: forward enqueues the HIP forward,
    backward the HIP backward, and returns the parameter gradients in the
    order the parameters were passed."""

    @staticmethod
    def forward(ctx, module, eng, x, *params):
        eng.m = module                       # the module (or DataParallel replica) whose tensors this call uses
        logits = eng.forward(x, module.training)
        ctx.eng = eng
        ctx.names = module._param_names
        ctx.needs = [p.requires_grad for p in params]
        ctx.plan = eng._last[0]
        ctx.lease = _Lease(ctx.plan)
        # hand out a copy: the plan's logits buffer is overwritten by the next forward
        # (the caller may keep the logits longer than the graph, e.g. for metrics)
        return logits.clone()

    @staticmethod
    def backward(ctx, dlogits):
        grads = ctx.eng.backward(dlogits, plan=ctx.plan)
        # copies: the flat gradient buffer is rewritten by the next backward (a second
        # forward of the same graph, models.py:190); autograd accumulates what we return and
        # may keep it as .grad, so it must not alias the engine's buffer
        out = tuple(grads[n].clone() if need else None for n, need in zip(ctx.names, ctx.needs))
        ctx.lease.release()
        return (None, None, None) + out


class FloorPlanSegmentationModel(nn.Module):
    """U-Net for segmenting walls, doors, windows, rooms (reference
    models.py:6).  `compute_dtype` ('fp32' default, or 'bf16'; env
    FEA_AMD_DTYPE) selects the storage type of activations and packed
    weights inside the HIP path; parameters, gradients and logits are fp32."""

    def __init__(self, in_channels=3, num_classes=6, *, compute_dtype=None):
        super().__init__()
        # construction order == reference order (models.py:10-24): it fixes
        # which RNG draws initialise which weight and the state_dict order
        self.enc1 = self._conv_block(in_channels, 64)
        self.enc2 = self._conv_block(64, 128)
        self.enc3 = self._conv_block(128, 256)
        self.enc4 = self._conv_block(256, 512)
        self.bottleneck = self._conv_block(512, 1024)
        self.upconv4 = nn.ConvTranspose2d(1024, 512, 2, stride=2)
        self.dec4 = self._conv_block(1024, 512)
        self.upconv3 = nn.ConvTranspose2d(512, 256, 2, stride=2)
        self.dec3 = self._conv_block(512, 256)
        self.upconv2 = nn.ConvTranspose2d(256, 128, 2, stride=2)
        self.dec2 = self._conv_block(256, 128)
        self.upconv1 = nn.ConvTranspose2d(128, 64, 2, stride=2)
        self.dec1 = self._conv_block(128, 64)
        self.out = nn.Conv2d(64, num_classes, 1)
        self.pool = nn.MaxPool2d(2)
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.compute_dtype = _resolve_dtype(compute_dtype)
        self._param_names = [n for n, _ in self.named_parameters()]
        # engines by (device, compute dtype).  Not sub-modules, not in state_dict.  The dict OBJECT is
        # shared with the replicas nn.DataParallel makes (their __dict__ is a shallow copy), so the
        # replica on device k finds device k's engine again on the next step; an engine resolves
        # its parameters from the module that calls it (see engine.named_tensors).
        self.__dict__["_engs"] = {}

    def _conv_block(self, in_c, out_c):
        return nn.Sequential(
            nn.Conv2d(in_c, out_c, 3, padding=1),
            nn.BatchNorm2d(out_c),
            nn.ReLU(inplace=True),
            nn.Conv2d(out_c, out_c, 3, padding=1),
            nn.BatchNorm2d(out_c),
            nn.ReLU(inplace=True),
        )

    # -- engine -----------------------------------------------------------
    def _engine(self, device=None) -> UNetEngine:
        """The engine (plans, packed weights, gradient buffer) of `device` (default: where the
        parameters live).  One per device: under nn.DataParallel (train_direct.py:23-25) every
        replica thread works with its own device's engine."""
        if device is None:
            device = self.out.weight.device
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        key = (str(device), self.compute_dtype)
        eng = self._engs.get(key)
        if eng is None and device.type == "cuda":
            # an engine handed out while the module was still on the CPU (UNetTrainer(model) before model.cuda()) binds
            # its device at its first forward: it becomes this device's engine, not a second one with plans of its own
            stale = self._engs.get(("cpu", self.compute_dtype))
            if stale is not None and (stale.device is None or stale.device == device):
                eng = self._engs[key] = self._engs.pop(("cpu", self.compute_dtype))
        if eng is None:
            eng = UNetEngine(self, self.compute_dtype)
            self._engs[key] = eng
        return eng

    def train(self, mode: bool = True):
        # leaving / entering eval mode: packed weights may be stale (fused
        # optimizers do not bump Parameter._version)
        for eng in self.__dict__.get("_engs", {}).values():
            eng.invalidate()
        return super().train(mode)

    def set_compute_dtype(self, compute_dtype):
        self.compute_dtype = _resolve_dtype(compute_dtype)
        return self

    def forward(self, x):
        if not x.is_cuda:
            raise L.FeaError("fea_amd.FloorPlanSegmentationModel runs on a ROCm device only; "
                             "there is no CPU fallback (use the reference module for CPU execution)")
        # by attribute, not self.parameters(): a DataParallel replica has none registered
        tens = named_tensors(self, self._param_names)
        w = tens["out.weight"]
        if w.dtype != torch.float32:
            raise L.FeaError("parameters must stay fp32 (use compute_dtype='bf16' for bf16 arithmetic)")
        if w.device != x.device:
            raise L.FeaError(f"input on {x.device} but parameters on {w.device}")
        in_dtype = x.dtype
        eng = self._engine(x.device)
        if torch.is_grad_enabled() and any(t.requires_grad for t in tens.values()):
            logits = _UNetFunction.apply(self, eng, x, *tens.values())
        else:
            eng.m = self
            logits = eng.forward(x, self.training).clone()
        return logits if in_dtype == torch.float32 else logits.to(in_dtype)

    # -- checkpoints --------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True, assign=False):
        """Accepts the reference's spellings too: keys prefixed with `module.`
        (saved from nn.DataParallel, train_direct.py:25,130) and/or
        `segmentation_model.` (saved from FloorPlanTo3DPipeline, models.py:183);
        keys of the pipeline's other sub-modules are ignored."""
        own = set(self.state_dict().keys())
        cleaned = {}
        for k, v in state_dict.items():
            kk = k
            changed = True
            while changed:
                changed = False
                for pre in ("module.", "segmentation_model."):
                    if kk.startswith(pre):
                        kk = kk[len(pre):]
                        changed = True
            if kk in own:
                cleaned[kk] = v
            elif kk == k and strict:
                cleaned[kk] = v           # let nn.Module report the unexpected key
        return super().load_state_dict(cleaned, strict=strict, assign=assign)
