"""Segmentation loss, mask binarisation and IoU on the HIP kernels
(build-defined, the reference has none)."""
from __future__ import annotations

import torch

from . import _lib as L


class _BceDice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        n, c, h, w = logits.shape
        lg = logits.detach().contiguous().float()
        tg = target.contiguous()
        if tg.dtype != torch.uint8:
            tg = (tg > 0).to(torch.uint8)
        dev = lg.device
        part = torch.empty(L.call("fea_loss_blocks", n, c, h * w) * 4, dtype=torch.float32, device=dev)
        sums = torch.empty((c, 4), dtype=torch.float64, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        S = L.stream_ptr()
        L.call("fea_bce_dice_sums", lg, tg, sums, part, n, c, h * w, S)
        L.call("fea_bce_dice_grad", lg, tg, sums, loss, None, 1.0, n, c, h * w, S)
        ctx.save_for_backward(lg, tg, sums)
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        lg, tg, sums = ctx.saved_tensors
        n, c, h, w = lg.shape
        dl = torch.empty_like(lg)
        L.call("fea_bce_dice_grad", lg, tg, sums, None, dl, 1.0, n, c, h * w, L.stream_ptr())
        return dl * gout, None


def bce_dice_loss(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """BCEWithLogits(mean) + 1 - mean_c Dice_c  (see oracle/unet_oracle.py:bce_dice_loss)."""
    if not logits.is_cuda:
        raise L.FeaError("fea_amd.functional runs on a ROCm device only")
    return _BceDice.apply(logits, target)


def bce_dice_loss_and_grad(logits: torch.Tensor, target_u8: torch.Tensor, dlogits: torch.Tensor = None, ws=None):
    """Fused fast path for training loops: returns (loss[1], dlogits) without autograd."""
    n, c, h, w = logits.shape
    dev = logits.device
    if ws is None:
        ws = {}
    key = (n, c, h, w)
    if ws.get("key") != key:
        ws["key"] = key
        ws["part"] = torch.empty(L.call("fea_loss_blocks", n, c, h * w) * 4, dtype=torch.float32, device=dev)
        ws["sums"] = torch.empty((c, 4), dtype=torch.float64, device=dev)
        ws["loss"] = torch.empty(1, dtype=torch.float32, device=dev)
        ws["dl"] = torch.empty((n, c, h, w), dtype=torch.float32, device=dev)
    if dlogits is None:
        dlogits = ws["dl"]
    S = L.stream_ptr()
    L.call("fea_bce_dice_sums", logits, target_u8, ws["sums"], ws["part"], n, c, h * w, S)
    L.call("fea_bce_dice_grad", logits, target_u8, ws["sums"], ws["loss"], dlogits, 1.0, n, c, h * w, S)
    return ws["loss"], dlogits


def binarise_mask(grey_u8: torch.Tensor) -> torch.Tensor:
    """target = grey > 127 (uint8 {0,1})."""
    g = grey_u8.contiguous()
    out = torch.empty_like(g)
    L.call("fea_binarise", g, out, g.numel(), L.stream_ptr())
    return out


def mask_iou_counts(logits: torch.Tensor, target_u8: torch.Tensor, want_mask: bool = True, inter=None, union=None):
    """pred = logits > 0; per-class int64 intersection / union counts
    (accumulated into `inter`/`union` when given)."""
    n, c, h, w = logits.shape
    lg = logits.detach().contiguous().float()
    tg = target_u8.contiguous()
    dev = lg.device
    mask = torch.empty((n, c, h, w), dtype=torch.uint8, device=dev) if want_mask else None
    if inter is None:
        inter = torch.zeros(c, dtype=torch.int64, device=dev)
    if union is None:
        union = torch.zeros(c, dtype=torch.int64, device=dev)
    L.call("fea_mask_iou", lg, tg, mask, inter, union, n, c, h * w, L.stream_ptr())
    return mask, inter, union


def threshold_mask(logits: torch.Tensor) -> torch.Tensor:
    """pred = logits > 0 as uint8 {0,1} (inference.py:39-40 without ground truth): fea_mask_iou with no target."""
    n, c, h, w = logits.shape
    lg = logits.detach().contiguous().float()
    mask = torch.empty((n, c, h, w), dtype=torch.uint8, device=lg.device)
    L.call("fea_mask_iou", lg, None, mask, None, None, n, c, h * w, L.stream_ptr())
    return mask


def iou_from_counts(inter: torch.Tensor, union: torch.Tensor):
    iou = torch.where(union > 0, inter.double() / union.clamp(min=1).double(), torch.ones_like(inter, dtype=torch.float64))
    return iou, iou.mean()
