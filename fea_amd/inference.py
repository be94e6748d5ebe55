"""Inference entry points on the HIP path.

`Inference` mirrors the shape of the reference's `Inference` class
(data_model/inference.py:9-44): load a checkpoint, `.eval()`, forward under
`torch.no_grad()`.  The reference runs whole 512x512 images at batch 1; for
high-resolution plans (BASELINE.json configs[4]: 2048x2048) `TiledSegmenter`
cuts each image into overlapping windows, runs them as batches and stitches
the logits, then thresholds (`logits > 0`) and counts per-class intersection /
union on the device.

Why tiling is exact: with BatchNorm in eval mode every layer of the UNet is a
per-pixel affine map or a local operator, so a logit depends on the input only
inside its receptive field (radius 92 px: SURVEY.md 3d).  A window therefore
reproduces the whole-image logits on every pixel that is at least HALO = 96 px
away from window edges that are NOT image edges (at a true image edge the
window's zero padding is the image's zero padding).  Windows have one size
(core + 2*HALO) and are shifted inwards at the image border instead of being
padded, and their origins are multiples of 16, so the four 2x2 max-pools see
the same phase as in the whole image.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from . import functional as Fn
from .models import FloorPlanSegmentationModel

HALO = 96            # >= receptive-field radius 92, multiple of 16


def window_plan(H: int, W: int, core, halo: int = HALO) -> Tuple[int, int, List[Tuple[int, int, int, int, int, int]]]:
    """Windows covering an H x W image.  `core` is an int or (core_h, core_w).  Returns
    (win_h, win_w, [(wy, wx, cy0, cy1, cx0, cx1)]): a window is the win_h x win_w crop at (wy, wx); it owns the core
    rectangle [cy0,cy1) x [cx0,cx1) (image coordinates).  Cores partition the image; every core pixel is >= halo away
    from each window edge that is not an image edge.  A direction in which core + 2*halo covers the image is not cut:
    (1024, 2048) cuts a 2048 x 2048 image into two 1216 x 2048 windows (19 % extra pixels instead of 41 % for four
    1216 x 1216 windows)."""
    ch, cw = (core, core) if isinstance(core, int) else core
    if H % 16 or W % 16 or ch % 16 or cw % 16 or halo % 16:
        raise ValueError("H, W, core and halo must be multiples of 16")
    wh = H if ch + 2 * halo >= H else ch + 2 * halo
    ww = W if cw + 2 * halo >= W else cw + 2 * halo
    ys = [(0, H)] if wh == H else [(c0, min(H, c0 + ch)) for c0 in range(0, H, ch)]
    xs = [(0, W)] if ww == W else [(c0, min(W, c0 + cw)) for c0 in range(0, W, cw)]
    out = []
    for cy0, cy1 in ys:
        wy = min(max(cy0 - halo, 0), H - wh)
        for cx0, cx1 in xs:
            wx = min(max(cx0 - halo, 0), W - ww)
            out.append((wy, wx, cy0, cy1, cx0, cx1))
    return wh, ww, out


def reduce_counts(inter: torch.Tensor, union: torch.Tensor, group=None):
    """Sum the per-class int64 (intersection, union) counts over the ranks of `group`
    (images are sharded over ranks; this is the only exchange step of the inference path)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        both = torch.stack([inter, union])
        dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)
        return both[0], both[1]
    return inter, union


class TiledSegmenter:
    """Batched, tiled eval-mode inference + mask / IoU for large images.  A window computes
    (core + 2*halo)^2 pixels for core^2 useful ones: core 1024 (1.41x) runs 2048x2048 images
    45 % faster than core 512 (1.89x) on one MI355X (scripts/bench_infer.py)."""

    # windows per forward when the caller does not say: as many as make one batch-64 @512x512 forward's worth of pixels
    # (fewer, larger launches: 2 -> 4 windows of 1216x2048 per forward is 141 -> 152 images/s at 2048x2048)
    PIXELS_PER_FORWARD = 64 * 512 * 512

    def __init__(self, model: FloorPlanSegmentationModel, core=1024, halo: int = HALO, window_batch: Optional[int] = None):
        self.model = model.eval()
        self.core, self.halo, self.window_batch = core, halo, window_batch

    @torch.no_grad()
    def logits(self, x: torch.Tensor) -> torch.Tensor:
        """x [N,C,H,W] on the device -> stitched logits [N,classes,H,W] fp32."""
        n, _, H, W = x.shape
        wh, ww, plan = window_plan(H, W, self.core, self.halo)
        if len(plan) == 1:
            return self.model(x)
        out = torch.empty((n, self.model.num_classes, H, W), dtype=torch.float32, device=x.device)
        jobs = [(i,) + p for i in range(n) for p in plan]
        wb = self.window_batch or max(1, self.PIXELS_PER_FORWARD // (wh * ww))
        wb = -(-len(jobs) // -(-len(jobs) // wb))          # equal chunks: 8 windows under a limit of 6 run as 4 + 4, not 6 + 2
        for b0 in range(0, len(jobs), wb):
            chunk = jobs[b0:b0 + wb]
            crops = torch.stack([x[i, :, wy:wy + wh, wx:wx + ww] for (i, wy, wx, *_r) in chunk])
            lg = self.model(crops.contiguous())
            for k, (i, wy, wx, cy0, cy1, cx0, cx1) in enumerate(chunk):
                out[i, :, cy0:cy1, cx0:cx1] = lg[k, :, cy0 - wy:cy1 - wy, cx0 - wx:cx1 - wx]
        return out

    @torch.no_grad()
    def segment(self, x: torch.Tensor, target_u8: Optional[torch.Tensor] = None, group=None):
        """-> (mask uint8 [N,classes,H,W], inter int64[classes], union int64[classes]).
        Under torch.distributed every rank passes ITS shard of the images; the counts are
        summed over ranks (one int64 all-reduce: the path's only exchange step)."""
        lg = self.logits(x)
        if target_u8 is None:
            return Fn.threshold_mask(lg), None, None
        mask, inter, union = Fn.mask_iou_counts(lg, target_u8, want_mask=True)
        inter, union = reduce_counts(inter, union, group)
        return mask, inter, union


class Inference:
    """load -> eval -> no_grad forward (reference data_model/inference.py:9-44, UNet part)."""

    def __init__(self, checkpoint_path: Optional[str] = None, device: str = "cuda", compute_dtype="bf16",
                 in_channels: int = 3, num_classes: int = 6):
        self.device = torch.device(device)
        self.model = FloorPlanSegmentationModel(in_channels, num_classes, compute_dtype=compute_dtype)
        if checkpoint_path is not None:
            ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
            sd = ckpt.get("model_state_dict", ckpt)          # train_direct.py:127-135 schema, or a bare state_dict
            self.model.load_state_dict(sd, strict=False)
        self.model.to(self.device).eval()

    @torch.no_grad()
    def predict(self, floor_plans: List[torch.Tensor]) -> List[torch.Tensor]:
        """One logits tensor per floor image tensor [B,3,H,W] (models.py:190)."""
        return [self.model(fp.to(self.device)) for fp in floor_plans]
