"""Synthetic 'floor-plan shaped' batches for benchmarks and smoke runs.

Same generator as the test oracle's (kept textually identical so that the CPU
baseline and the HIP path see the same tensors), but owned by the product
package: nothing under fea_amd/ imports the oracle.
"""
import torch


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
