"""Shared helpers for parity tests (digests, error metrics)."""
import json
import os

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "unet_golden.json")


def load_golden():
    with open(GOLDEN) as f:
        return json.load(f)


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    """||a-b|| / ||b|| in float64 (b = reference)."""
    a, b = a.detach().double().flatten().cpu(), b.detach().double().flatten().cpu()
    d = (a - b).norm().item()
    n = b.norm().item()
    return d / n if n > 0 else d


def max_rel(a: torch.Tensor, b: torch.Tensor) -> float:
    """max|a-b| / max|b| (peak-normalised max error)."""
    a, b = a.detach().double().flatten().cpu(), b.detach().double().flatten().cpu()
    pk = b.abs().max().item()
    d = (a - b).abs().max().item()
    return d / pk if pk > 0 else d


def digest_err(gold: dict, t: torch.Tensor) -> float:
    """Worst relative deviation of tensor `t` from a golden digest: L2 and
    abs-sum relative error, and strided samples relative to the sample peak."""
    from oracle.unet_oracle import digest
    d = digest(t, len(gold["samples"]))
    assert d["n"] == gold["n"], (d["n"], gold["n"])
    errs = []
    for key in ("l2", "abs"):
        ref = gold[key]
        errs.append(abs(d[key] - ref) / ref if ref > 0 else abs(d[key]))
    gs = torch.tensor(gold["samples"], dtype=torch.float64)
    ds = torch.tensor(d["samples"], dtype=torch.float64)
    scale = max(gs.abs().max().item(), gold["l2"] / max(gold["n"], 1) ** 0.5, 1e-30)
    errs.append(((gs - ds).abs().max() / scale).item())
    return max(errs)


DENSE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "unet_dense.npz")


def load_dense():
    """tests/golden/unet_dense.npz (oracle/make_golden_dense.py): FULL reference tensors of case train_32."""
    import numpy as np
    z = np.load(DENSE, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return z, meta


def dense_take(t: torch.Tensor, meta: dict) -> torch.Tensor:
    """The element set oracle/make_golden_dense.py:element_set keeps of a tensor: every element up to meta['whole'],
    else the first meta['head'] elements plus meta['head'] evenly strided ones."""
    f = t.detach().reshape(-1).float().cpu()
    n = f.numel()
    if n <= meta["whole"]:
        return f
    head = meta["head"]
    idx = torch.unique(torch.cat([torch.arange(head), torch.arange(head) * (n // head)]))
    return f[idx]


def dense_err(got: torch.Tensor, ref) -> float:
    """max |got - ref| over the kept elements / RMS of ref (element-by-element: no digest, no norm ratio that a
    few wrong elements could hide in)."""
    r = torch.as_tensor(ref).double().flatten()
    g = got.double().flatten()
    assert g.numel() == r.numel(), (g.numel(), r.numel())
    rms = r.pow(2).mean().sqrt().item()
    return ((g - r).abs().max().item()) / max(rms, 1e-30)
