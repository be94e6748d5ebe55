"""CPU oracle for the UNet hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) restatement of the reference's
`FloorPlanSegmentationModel` forward/backward and of the optimizer step that
drives it.  It is the *checker* for the HIP path: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.
Nothing under `fea_amd/` imports it, and the product path never falls back to
it (the product raises if the HIP library is missing).

Parity status: PINNED.  `oracle/make_golden.py` imports the reference module
(`/root/reference/data_model/models.py`) in the development container and
writes digests of its outputs into `tests/golden/`; `tests/test_oracle_golden.py`
checks this restatement against those digests on every CPU test run.  The
reference itself ships no tests or golden vectors for this path.

Reference lines restated (all in `/root/reference/data_model/`):
  models.py:8-24   construction order + default init      -> `init_state_dict`
  models.py:26-34  conv3x3+BN+ReLU x2 block                -> `_conv_block`
  models.py:36-46  UNet dataflow (pool / upconv / cat)     -> `forward`
  train_direct.py:28-32,105-106  clip_grad_norm_ + AdamW   -> `adamw_step`
The segmentation loss / mask / IoU have no
reference implementation ("parity unpinned" at the reference); they are
defined here once and the HIP kernels are checked against these definitions.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # nn.BatchNorm2d default (models.py:29,32)
BN_MOMENTUM = 0.1

_BLOCKS = ["enc1", "enc2", "enc3", "enc4", "bottleneck", "dec4", "dec3", "dec2", "dec1"]


def block_channels(in_channels: int = 3):
    """(name, C_in, C_out) for the nine conv blocks, in forward order."""
    return [
        ("enc1", in_channels, 64), ("enc2", 64, 128), ("enc3", 128, 256),
        ("enc4", 256, 512), ("bottleneck", 512, 1024),
        ("dec4", 1024, 512), ("dec3", 512, 256), ("dec2", 256, 128), ("dec1", 128, 64),
    ]


# --------------------------------------------------------------------------
# A1: construction order and default initialisation (models.py:8-24)
# --------------------------------------------------------------------------
def _uniform(shape, bound, gen):
    # nn.init.uniform_ / kaiming_uniform_ draw with Tensor.uniform_ on the
    # default CPU generator; a fresh empty tensor + uniform_ consumes the same
    # stream.
    t = torch.empty(shape, dtype=torch.float32)
    if gen is None:
        return t.uniform_(-bound, bound)
    return t.uniform_(-bound, bound, generator=gen)


def _conv_init(cout, cin, k, gen, transposed=False):
    """nn.Conv2d / nn.ConvTranspose2d.reset_parameters: kaiming_uniform(a=sqrt 5)
    == U(+-1/sqrt(fan_in)); bias U(+-1/sqrt(fan_in)).  fan_in is computed from
    weight.size(1)*k*k, which for ConvTranspose2d ([C_in, C_out, k, k]) is
    C_out*k*k."""
    wshape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
    fan_in = wshape[1] * k * k
    gain = math.sqrt(2.0 / (1 + math.sqrt(5.0) ** 2))
    std = gain / math.sqrt(fan_in)
    wbound = math.sqrt(3.0) * std
    w = _uniform(wshape, wbound, gen)
    bbound = 1.0 / math.sqrt(fan_in)
    b = _uniform((cout,), bbound, gen)
    return w, b


def init_state_dict(seed: int | None, in_channels: int = 3, num_classes: int = 6) -> "OrderedDict[str, torch.Tensor]":
    """State dict with the reference's key names, shapes and RNG draw order:
    enc1..4, bottleneck, upconv4, dec4, upconv3, dec3, upconv2, dec2, upconv1,
    dec1, out (models.py:10-23).  With `seed` not None the global CPU RNG is
    seeded exactly like `torch.manual_seed(seed)` before building the module."""
    if seed is not None:
        torch.manual_seed(seed)
    gen = None
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def conv_block(name, cin, cout):
        for idx, (ci, co) in ((0, (cin, cout)), (3, (cout, cout))):
            w, b = _conv_init(co, ci, 3, gen)
            sd[f"{name}.{idx}.weight"] = w
            sd[f"{name}.{idx}.bias"] = b
            bn = idx + 1
            sd[f"{name}.{bn}.weight"] = torch.ones(co)
            sd[f"{name}.{bn}.bias"] = torch.zeros(co)
            sd[f"{name}.{bn}.running_mean"] = torch.zeros(co)
            sd[f"{name}.{bn}.running_var"] = torch.ones(co)
            sd[f"{name}.{bn}.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def upconv(name, cin, cout):
        w, b = _conv_init(cout, cin, 2, gen, transposed=True)
        sd[f"{name}.weight"] = w
        sd[f"{name}.bias"] = b

    ch = {n: (ci, co) for n, ci, co in block_channels(in_channels)}
    for n in ("enc1", "enc2", "enc3", "enc4", "bottleneck"):
        conv_block(n, *ch[n])
    for k, cin in ((4, 1024), (3, 512), (2, 256), (1, 128)):
        upconv(f"upconv{k}", cin, cin // 2)
        conv_block(f"dec{k}", *ch[f"dec{k}"])
    w, b = _conv_init(num_classes, 64, 1, gen)
    sd["out.weight"] = w
    sd["out.bias"] = b
    # nn.Module.state_dict() orders by registration: enc1..4, bottleneck,
    # upconv4, dec4, ... , out -- which is the order built above.
    return sd


def split_state(sd: Dict[str, torch.Tensor]) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """(trainable params, buffers) from a state dict (strips `module.` /
    `segmentation_model.` prefixes the reference's checkpoints may carry,
    SURVEY.md §3b)."""
    params, bufs = OrderedDict(), OrderedDict()
    for k, v in sd.items():
        stripped = True
        while stripped:
            stripped = False
            for pre in ("module.", "segmentation_model."):
                if k.startswith(pre):
                    k = k[len(pre):]
                    stripped = True
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            bufs[k] = v
        else:
            params[k] = v
    return params, bufs


# --------------------------------------------------------------------------
# A2-A8: forward
# --------------------------------------------------------------------------
def _batch_norm(x, gamma, beta, rm, rv, training, new_bufs, prefix):
    """nn.BatchNorm2d (models.py:29,32): train = batch mean / biased var,
    running stats updated with momentum 0.1 and the UNBIASED variance; eval =
    running stats."""
    if training:
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        with torch.no_grad():
            new_bufs[prefix + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean.detach()
            new_bufs[prefix + ".running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var.detach() * (n / max(n - 1, 1))
    else:
        mean, var = rm, rv
        new_bufs[prefix + ".running_mean"] = rm
        new_bufs[prefix + ".running_var"] = rv
    inv = torch.rsqrt(var + BN_EPS)
    return (x - mean[None, :, None, None]) * (inv * gamma)[None, :, None, None] + beta[None, :, None, None]


def _conv_block(x, p, b, name, training, new_bufs, taps=None, forced=None):
    for idx in (0, 3):
        x = F.conv2d(x, p[f"{name}.{idx}.weight"], p[f"{name}.{idx}.bias"], padding=1)
        if taps is not None:
            taps[f"{name}.{idx}.conv"] = x
        bn = f"{name}.{idx + 1}"
        x = _batch_norm(x, p[bn + ".weight"], p[bn + ".bias"], b[bn + ".running_mean"],
                        b[bn + ".running_var"], training, new_bufs, bn)
        if training:
            new_bufs[bn + ".num_batches_tracked"] = b[bn + ".num_batches_tracked"] + 1
        else:
            new_bufs[bn + ".num_batches_tracked"] = b[bn + ".num_batches_tracked"]
        if taps is not None:
            taps[f"{name}.{idx}.bn"] = x             # pre-activation (input of the ReLU)
        if forced is not None:
            # kinks pinned: the ReLU mask comes from the implementation under
            # test, so both sides differentiate the same smooth branch
            x = x * forced[f"{name}.{idx}.mask"].to(x.dtype)
        else:
            x = torch.relu(x)
        if taps is not None:
            taps[f"{name}.{idx}.act"] = x
    return x


def _upconv(x, w, bias):
    """nn.ConvTranspose2d(ci, co, 2, stride=2) (models.py:15-21):
    y[n,co,2i+a,2j+b] = bias[co] + sum_ci x[n,ci,i,j] * w[ci,co,a,b]."""
    return F.conv_transpose2d(x, w, bias, stride=2)


def kink_disagreements(taps_hip, taps_free, rel_band=1e-5):
    """How far the kink decisions of an implementation under test (`taps_hip`: its activated outputs) are from this
    oracle's own (`taps_free`: taps of a free forward(), with `.bn` pre-activations).  Pinning the kinks (forced_from_taps)
    is only legitimate where the two disagree inside a rounding band: returns, per layer,
      relu:  (number of mask disagreements, how many of them have |pre-activation| >= rel_band * RMS(pre-activation), numel,
              worst |pre-activation| / RMS among the disagreements)
      pool:  (number of arg-max disagreements, how many of them have a window gap (max - runner-up) >= rel_band * RMS, windows,
              worst gap / RMS among the disagreements)
    A kernel that mis-masks pixels away from the kink shows up in the second and fourth figures."""
    out = {}
    for k, act in taps_hip.items():
        if not k.endswith(".act"):
            continue
        pre = taps_free[k[:-4] + ".bn"].detach()
        rms = pre.pow(2).mean().sqrt().item()
        dis = (act > 0) != (pre > 0)
        worst = (pre.abs()[dis].max().item() / rms) if dis.any() else 0.0
        out[k[:-4] + ".relu"] = (int(dis.sum()), int((dis & (pre.abs() >= rel_band * rms)).sum()), pre.numel(), worst)
    for blk in ("enc1", "enc2", "enc3", "enc4"):
        a_free = taps_free[f"{blk}.3.act"].detach()
        _, idx_hip = F.max_pool2d(taps_hip[f"{blk}.3.act"].float(), 2, return_indices=True)
        _, idx_free = F.max_pool2d(a_free, 2, return_indices=True)
        n, c, h, w = a_free.shape
        win = a_free.view(n, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4)
        top2 = win.topk(2, dim=-1).values
        gap = top2[..., 0] - top2[..., 1]
        rms = a_free.pow(2).mean().sqrt().item()
        dis = idx_hip != idx_free
        worst = (gap[dis].max().item() / rms) if dis.any() else 0.0
        out[f"{blk}.pool"] = (int(dis.sum()), int((dis & (gap >= rel_band * rms)).sum()), dis.numel(), worst)
    return out


def forced_from_taps(taps):
    """Kink decisions (ReLU masks, max-pool arg-max) of an implementation under
    test, from its activated outputs `taps["<layer>.act"]` (NCHW)."""
    forced = {}
    for k, v in taps.items():
        if k.endswith(".act"):
            forced[k[:-4] + ".mask"] = v > 0
    for blk in ("enc1", "enc2", "enc3", "enc4"):
        _, idx = F.max_pool2d(taps[f"{blk}.3.act"].float(), 2, return_indices=True)
        forced[f"{blk}.pool_idx"] = idx
    return forced


def forward(params, bufs, x, training: bool = True, taps=None, forced=None):
    """models.py:36-46.  Returns (logits, new_buffers).  `x` is NCHW fp32,
    H and W multiples of 16.  With `forced` (see forced_from_taps) the ReLU
    masks and pool arg-max are taken from there instead of from this run's
    own activations: the function is then smooth in its inputs, which is what
    makes a 1e-4 gradient comparison meaningful (oracle/make_golden.py:
    well_conditioned explains why it otherwise is not)."""
    nb: Dict[str, torch.Tensor] = OrderedDict()

    def make_pool(blk):
        if forced is None:
            return lambda t: F.max_pool2d(t, 2)
        idx = forced[f"{blk}.pool_idx"]
        return lambda t: torch.gather(t.flatten(2), 2, idx.flatten(2)).view(idx.shape)

    cb = lambda t, name: _conv_block(t, params, bufs, name, training, nb, taps, forced)
    e1 = cb(x, "enc1")
    e2 = cb(make_pool("enc1")(e1), "enc2")
    e3 = cb(make_pool("enc2")(e2), "enc3")
    e4 = cb(make_pool("enc3")(e3), "enc4")
    bt = cb(make_pool("enc4")(e4), "bottleneck")
    d = bt
    for k, skip in ((4, e4), (3, e3), (2, e2), (1, e1)):
        up = _upconv(d, params[f"upconv{k}.weight"], params[f"upconv{k}.bias"])
        if taps is not None:
            taps[f"upconv{k}"] = up
        d = cb(torch.cat([up, skip], dim=1), f"dec{k}")
    logits = F.conv2d(d, params["out.weight"], params["out.bias"])
    return logits, nb


# --------------------------------------------------------------------------
# L1-L4: build-defined loss / mask / IoU (no reference implementation)
# --------------------------------------------------------------------------
def bce_dice_loss(logits, target):
    """L = BCEWithLogits(mean) + (1 - mean_c dice_c),
    dice_c = (2*sum(p*t) + 1) / (sum(p) + sum(t) + 1) over (N,H,W), p = sigmoid(x).
    `target` is {0,1} of the logits' shape (any dtype)."""
    t = target.to(logits.dtype)
    bce = F.binary_cross_entropy_with_logits(logits, t, reduction="mean")
    p = torch.sigmoid(logits)
    inter = (p * t).sum(dim=(0, 2, 3))
    denom = p.sum(dim=(0, 2, 3)) + t.sum(dim=(0, 2, 3))
    dice = (2 * inter + 1) / (denom + 1)
    return bce + (1 - dice.mean())


def binarise_mask(grey_u8):
    """L1: target = grey > 127."""
    return (grey_u8 > 127).to(torch.uint8)


def mask_iou_counts(logits, target):
    """L1/L4: pred = logits > 0; per class int64 intersection / union counts."""
    pred = logits > 0
    t = target > 0
    inter = (pred & t).sum(dim=(0, 2, 3)).to(torch.int64)
    union = (pred | t).sum(dim=(0, 2, 3)).to(torch.int64)
    return pred.to(torch.uint8), inter, union


def iou_from_counts(inter, union):
    iou = torch.where(union > 0, inter.double() / union.clamp(min=1).double(), torch.ones_like(inter, dtype=torch.float64))
    return iou, iou.mean()


# --------------------------------------------------------------------------
# A9/A10: backward via autograd on the restated forward; clip + AdamW by hand
# --------------------------------------------------------------------------
def loss_and_grads(params, bufs, x, target, training=True, forced=None):
    p = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in params.items())
    logits, nb = forward(p, bufs, x, training, forced=forced)
    loss = bce_dice_loss(logits, target)
    grads = torch.autograd.grad(loss, list(p.values()))
    return loss.detach(), logits.detach(), OrderedDict(zip(p.keys(), grads)), nb


def grads_from_dlogits(params, bufs, x, dlogits, training=True, forced=None):
    """Backward of the UNet alone for a given upstream gradient (what the
    drop-in's autograd.Function.backward must reproduce)."""
    p = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in params.items())
    logits, nb = forward(p, bufs, x, training, forced=forced)
    grads = torch.autograd.grad(logits, list(p.values()), grad_outputs=dlogits)
    return logits.detach(), OrderedDict(zip(p.keys(), grads)), nb


def clip_grad_norm(grads, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_ (train_direct.py:105): total L2 norm,
    scale by min(1, max_norm / (norm + 1e-6))."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return OrderedDict((k, g * coef) for k, g in grads.items()), total


def adamw_step(params, grads, state, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5):
    """torch.optim.AdamW single step (train_direct.py:28-32,106): decoupled
    weight decay, bias-corrected moments."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    b1, b2 = betas
    out = OrderedDict()
    for k, p in params.items():
        g = grads[k]
        m = state.setdefault("m", {}).get(k, torch.zeros_like(p))
        v = state.setdefault("v", {}).get(k, torch.zeros_like(p))
        p = p * (1 - lr * weight_decay)
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        state["m"][k], state["v"][k] = m, v
        bc1 = 1 - b1 ** t
        bc2 = 1 - b2 ** t
        denom = v.sqrt() / math.sqrt(bc2) + eps
        out[k] = p - (lr / bc1) * m / denom
    return out


def cosine_lr(base_lr, epoch, t_max, eta_min=1e-6):
    """CosineAnnealingLR closed form (train_direct.py:35-39)."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / t_max)) / 2


# --------------------------------------------------------------------------
# digests used by the golden fixtures
# --------------------------------------------------------------------------
def digest(t: torch.Tensor, nsample: int = 64):
    """Size-independent summary of a tensor: sum, abs-sum, L2 and `nsample`
    evenly strided elements (float64 lists, JSON-serialisable)."""
    f = t.detach().double().flatten()
    n = f.numel()
    if n == 0:
        return {"n": 0, "sum": 0.0, "abs": 0.0, "l2": 0.0, "samples": []}
    idx = torch.linspace(0, n - 1, min(nsample, n)).long()
    return {"n": n, "sum": f.sum().item(), "abs": f.abs().sum().item(),
            "l2": f.norm().item(), "samples": f[idx].tolist()}


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
