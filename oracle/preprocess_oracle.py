"""CPU restatement of the reference's image transform -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench scripts' CPU legs may import this file; the
product path (fea_amd/data.py -> libfea_unet.so) never does.

What it restates (reference data_model/custom_dataset.py:23-28, applied at :69-71):

    transforms.Resize((img_size, img_size))      # on a PIL image
    transforms.ToTensor()                        # uint8 HWC -> float32 CHW / 255
    transforms.Normalize(mean, std)              # (t - mean) / std, float32

The arithmetic of Resize lives in a third-party dependency that is not part of
/root/reference: torchvision (`torchvision>=0.15.0`, requirements.txt:3) forwards a PIL image
to Pillow's `Image.resize(size, BILINEAR)` (`pillow>=9.5.0`, requirements.txt:5).  Pillow's
published algorithm (src/libImaging/Resample.c, unchanged since 3.x) is restated here:

  * separable, horizontal pass first, the intermediate image is rounded to uint8;
  * per output sample a window of `support = max(scale, 1)` input samples either side of
    the centre `(i + 0.5) * scale`, triangle weights evaluated at
    `(x - centre + 0.5) / max(scale, 1)`, normalised to sum 1 in double precision;
  * weights converted to fixed point with 22 fractional bits, rounding half away from
    zero; accumulation in int32 starting from 2^21; result `>> 22`, clipped to [0, 255].

PINNED: oracle/make_golden_preprocess.py runs Pillow itself (12.2.0 in this image) on seeded
inputs and on crops of the reference's own data files and stores input + Pillow's output
in tests/golden/preprocess_golden.npz; tests/test_preprocess_cpu.py checks this file against
them bit for bit.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
IMAGENET_MEAN = (0.485, 0.456, 0.406)      # custom_dataset.py:26
IMAGENET_STD = (0.229, 0.224, 0.225)       # custom_dataset.py:27


def resize_coeffs(in_size: int, out_size: int):
    """-> (ksize, bounds int32[out,2] = (first input sample, count), coeffs int32[out,ksize])."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale                       # bilinear: support 1
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = []
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            v = 1.0 - a if a < 1.0 else 0.0
            w.append(v)
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _pass(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """One separable pass along `axis` (0 = vertical, 1 = horizontal) of an [H,W,C] uint8 image."""
    in_size = img.shape[axis]
    _, bounds, kk = resize_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for i in range(out_size):
        x0, n = bounds[i]
        acc = np.tensordot(kk[i, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """Pillow `Image.resize((out_w, out_h), BILINEAR)` of a uint8 [H,W] or [H,W,C] image."""
    squeeze = img.ndim == 2
    a = img[:, :, None] if squeeze else img
    if a.shape[1] != out_w:
        a = _pass(a, out_w, 1)
    if a.shape[0] != out_h:
        a = _pass(a, out_h, 0)
    return a[:, :, 0] if squeeze else a


def to_tensor_normalize(img_u8: np.ndarray, mean=IMAGENET_MEAN, std=IMAGENET_STD) -> np.ndarray:
    """ToTensor + Normalize in float32, exactly as torch does it: u8 -> f32, / 255, - mean, / std."""
    t = img_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    m = np.asarray(mean, np.float32)[:, None, None]
    s = np.asarray(std, np.float32)[:, None, None]
    return ((t - m) / s).astype(np.float32)


def transform(img_u8: np.ndarray, img_size: int = 512) -> np.ndarray:
    """The whole transform of custom_dataset.py:23-28 for a decoded RGB uint8 [H,W,3] image."""
    return to_tensor_normalize(resize_bilinear_u8(img_u8, img_size, img_size))


def mask_target(grey_u8: np.ndarray, img_size: int = 512) -> np.ndarray:
    """Build-defined target: resize the grey mask like the image, then > 127."""
    return (resize_bilinear_u8(grey_u8, img_size, img_size) > 127).astype(np.uint8)
