"""Input pipeline on the HIP path.

Mirrors the reference's per-image transform and pairing logic
(data_model/custom_dataset.py:23-28,65-90): images are decoded on the CPU with
PIL (`Image.open(path).convert('RGB')`, custom_dataset.py:68 -- the shipped
floor plans are GIF palette images behind a .jpg name), uploaded as uint8 and
resized / scaled / normalised on the GPU by `fea_resize_u8`, bit-identical to
`Resize((S,S)) -> ToTensor() -> Normalize(mean, std)` on a PIL image.  Masks
(`dataset_masks/`, 8-bit grey) take the same resize and are thresholded at 127.

`synthetic_batch` makes the 'floor-plan shaped' tensors of SURVEY.md 8(d) for benchmarks
and smoke runs (same generator as the test oracle's, kept textually identical so that the
CPU baseline and the HIP path see the same tensors, but owned by the product package:
nothing under fea_amd/ imports the oracle).
"""
from __future__ import annotations

import collections
import os
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L

IMAGENET_MEAN = (0.485, 0.456, 0.406)     # custom_dataset.py:26
IMAGENET_STD = (0.229, 0.224, 0.225)      # custom_dataset.py:27


class ImagePreprocessor:
    """Resize((S,S)) + ToTensor + Normalize for decoded uint8 images, on the device."""

    def __init__(self, img_size: int = 512, mean: Sequence[float] = IMAGENET_MEAN, std: Sequence[float] = IMAGENET_STD,
                 device: str = "cuda"):
        self.size = int(img_size)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.FeaError("fea_amd.data runs on a ROCm device only (no CPU fallback)")
        L.load()
        lut = torch.empty(3 * 256, dtype=torch.float32)
        L.call("fea_normalize_lut", torch.tensor(mean, dtype=torch.float32), torch.tensor(std, dtype=torch.float32), 3, lut)
        self.lut = lut.to(self.device)
        self._tables: Dict[Tuple[int, int], Tuple[int, torch.Tensor, torch.Tensor]] = {}
        self._tmp: Optional[torch.Tensor] = None

    def tables(self, in_size: int, out_size: int):
        """(ksize, bounds, coeffs) on the device for one (input, output) length; cached."""
        key = (in_size, out_size)
        t = self._tables.get(key)
        if t is None:
            ks = L.call("fea_resize_ksize", in_size, out_size)
            if ks <= 0:
                raise L.FeaError(f"bad resize {in_size} -> {out_size}")
            b = torch.empty(2 * out_size, dtype=torch.int32)
            k = torch.empty(out_size * ks, dtype=torch.int32)
            L.call("fea_resize_coeffs", in_size, out_size, b, k)
            t = (ks, b.to(self.device), k.to(self.device))
            self._tables[key] = t
        return t

    def _run(self, src: torch.Tensor, dst: torch.Tensor, epilogue: int, H: int, W: int, threshold: int = 127):
        if src.dtype != torch.uint8 or src.dim() not in (2, 3):
            raise ValueError("expected a uint8 [H,W] or [H,W,C] image")
        src = src.to(self.device, non_blocking=True).contiguous()
        h, w = src.shape[:2]
        C = 1 if src.dim() == 2 else src.shape[2]
        xks, xb, xk = self.tables(w, W)
        yks, yb, yk = self.tables(h, H)
        need = h * W * C
        if self._tmp is None or self._tmp.numel() < need:
            self._tmp = torch.empty(need, dtype=torch.uint8, device=self.device)
        L.call("fea_resize_u8", src, h, w, C, self._tmp, dst, H, W, xb, xk, xks, yb, yk, yks, epilogue,
               self.lut if epilogue == 1 else None, threshold, L.stream_ptr())
        return dst

    def resize(self, img_u8: torch.Tensor, H: Optional[int] = None, W: Optional[int] = None) -> torch.Tensor:
        """PIL `resize((W,H), BILINEAR)`: uint8 [h,w(,C)] -> uint8 [H,W(,C)]."""
        H, W = H or self.size, W or self.size
        shape = (H, W) if img_u8.dim() == 2 else (H, W, img_u8.shape[2])
        return self._run(img_u8, torch.empty(shape, dtype=torch.uint8, device=self.device), 0, H, W)

    def image(self, rgb_u8: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Decoded RGB uint8 [h,w,3] -> float32 [3,S,S], the module's input (custom_dataset.py:23-28)."""
        if rgb_u8.dim() != 3 or rgb_u8.shape[2] != 3:
            raise ValueError("expected an RGB uint8 [H,W,3] image (use .convert('RGB') as the reference does)")
        S = self.size
        if out is None:
            out = torch.empty((3, S, S), dtype=torch.float32, device=self.device)
        return self._run(rgb_u8, out, 1, S, S)

    def mask(self, grey_u8: torch.Tensor, out: Optional[torch.Tensor] = None, threshold: int = 127) -> torch.Tensor:
        """Grey uint8 [h,w] -> uint8 {0,1} [S,S] = resize > threshold."""
        if grey_u8.dim() != 2:
            raise ValueError("expected a grey uint8 [H,W] mask")
        S = self.size
        if out is None:
            out = torch.empty((S, S), dtype=torch.uint8, device=self.device)
        return self._run(grey_u8, out, 2, S, S, threshold)


def decode_rgb(path: str) -> torch.Tensor:
    from PIL import Image
    with Image.open(path) as im:
        return torch.from_numpy(np.asarray(im.convert("RGB")).copy())      # custom_dataset.py:68


def decode_grey(path: str) -> torch.Tensor:
    from PIL import Image
    with Image.open(path) as im:
        return torch.from_numpy(np.asarray(im.convert("L")).copy())


class FloorPlanPairs:
    """floor_plans2/<name> <-> dataset_masks/<name> pairs (same file names).  Decoding runs
    in a small thread pool ahead of the GPU; batches come out as device tensors
    (x [N,3,S,S] float32, target [N,num_classes,S,S] uint8 with the mask in channel
    `mask_channel`, zeros elsewhere -- the shipped masks have one channel)."""

    def __init__(self, floor_plans_dir: str, masks_dir: str, img_size: int = 512, num_classes: int = 6,
                 mask_channel: int = 0, device: str = "cuda", workers: int = 4):
        names = sorted(set(os.listdir(floor_plans_dir)) & set(os.listdir(masks_dir)))
        self.items: List[Tuple[str, str]] = [(os.path.join(floor_plans_dir, n), os.path.join(masks_dir, n)) for n in names]
        if not self.items:
            raise FileNotFoundError(f"no file name is present in both {floor_plans_dir} and {masks_dir}")
        self.pre = ImagePreprocessor(img_size, device=device)
        self.num_classes, self.mask_channel, self.workers = num_classes, mask_channel, workers

    def __len__(self):
        return len(self.items)

    def batches(self, batch_size: int, shuffle: bool = True, seed: int = 0, rank: int = 0, world: int = 1
                ) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        """Batches of this rank's shard (items rank, rank+world, ... of the epoch's permutation)."""
        order = list(range(len(self.items)))
        if shuffle:
            order = torch.randperm(len(order), generator=torch.Generator().manual_seed(seed)).tolist()
        order = order[rank::world]
        S, dev = self.pre.size, self.pre.device
        # bounded prefetch: at most two batches of decoded images are held (or being decoded) ahead
        # of the GPU; each decoded pair is dropped as soon as it has been handed to the device
        window = max(2 * batch_size, self.workers)
        with ThreadPoolExecutor(max_workers=self.workers) as pool:
            futs = collections.deque()
            nxt = 0

            def refill():
                nonlocal nxt
                while nxt < len(order) and len(futs) < window:
                    i = order[nxt]
                    futs.append((pool.submit(decode_rgb, self.items[i][0]), pool.submit(decode_grey, self.items[i][1])))
                    nxt += 1

            refill()
            for b0 in range(0, len(order), batch_size):
                nb = min(batch_size, len(order) - b0)
                x = torch.empty((nb, 3, S, S), dtype=torch.float32, device=dev)
                t = torch.zeros((nb, self.num_classes, S, S), dtype=torch.uint8, device=dev)
                for j in range(nb):
                    fi, fm = futs.popleft()
                    self.pre.image(fi.result(), out=x[j])
                    self.pre.mask(fm.result(), out=t[j, self.mask_channel])
                    del fi, fm
                    refill()
                yield x, t


def synthetic_batch(n, h, w, num_classes=6, seed=1234, in_channels=3):
    """'Floor-plan shaped' synthetic input: white page, dark
    axis-aligned strokes, a few light-grey rectangles, ImageNet-normalised
    (custom_dataset.py:26-27); target channel 0 = strokes, others = random
    rectangles, ~10 % foreground.  Deterministic in (seed, shape)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.ones(n, 1, h, w)
    tgt = torch.zeros(n, num_classes, h, w, dtype=torch.uint8)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g).item())
    for i in range(n):
        for _ in range(ri(3, 8)):
            x0, y0 = ri(0, w - 8), ri(0, h - 8)
            x1, y1 = min(w, x0 + ri(8, max(8, w // 3))), min(h, y0 + ri(8, max(8, h // 3)))
            img[i, 0, y0:y1, x0:x1] = 0.85 + 0.1 * torch.rand(1, generator=g).item()
            c = ri(1, max(1, num_classes - 1)) % num_classes
            tgt[i, c, y0:y1, x0:x1] = 1
        for _ in range(ri(20, 60)):
            th = ri(2, 6)
            val = 0.3 * torch.rand(1, generator=g).item()
            if ri(0, 1):
                y0, x0 = ri(0, h - th), ri(0, w - 8)
                ln = ri(min(32, w // 2), w)
                img[i, 0, y0:y0 + th, x0:min(w, x0 + ln)] = val
                tgt[i, 0, y0:y0 + th, x0:min(w, x0 + ln)] = 1
            else:
                y0, x0 = ri(0, h - 8), ri(0, w - th)
                ln = ri(min(32, h // 2), h)
                img[i, 0, y0:min(h, y0 + ln), x0:x0 + th] = val
                tgt[i, 0, y0:min(h, y0 + ln), x0:x0 + th] = 1
    img = img.expand(n, in_channels, h, w).clone()
    mean = torch.tensor([0.485, 0.456, 0.406])
    std = torch.tensor([0.229, 0.224, 0.225])
    reps = (in_channels + 2) // 3
    mean = mean.repeat(reps)[:in_channels].view(1, -1, 1, 1)
    std = std.repeat(reps)[:in_channels].view(1, -1, 1, 1)
    return (img - mean) / std, tgt
