#!/usr/bin/env python3
"""Per-call timing of one training step (HIP events around every C-ABI call):
which layer shapes are slow.  Usage: python scripts/layer_prof.py [--batch 64] [--size 512]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import fea_amd
    from fea_amd import _lib as L
    from fea_amd.data import synthetic_batch
    from fea_amd.train import UNetTrainer
    torch.manual_seed(0)
    model = fea_amd.FloorPlanSegmentationModel(compute_dtype=a.dtype).cuda()
    tr = UNetTrainer(model)
    x, t = synthetic_batch(a.batch, a.size, a.size, seed=1234)
    x, t = x.cuda(), t.cuda()
    for _ in range(2):
        tr.step(x, t)
    torch.cuda.synchronize()
    L.PROFILE = []
    for _ in range(a.reps):
        tr.step(x, t)
    torch.cuda.synchronize()
    prof, L.PROFILE = L.PROFILE, None
    per_step = len(prof) // a.reps
    rows = []
    for i in range(per_step):
        name, _, _, args = prof[i]
        ms = sum(prof[i + r * per_step][1].elapsed_time(prof[i + r * per_step][2]) for r in range(a.reps)) / a.reps
        ints = [v for v in args if isinstance(v, int) and 0 < v < (1 << 24)]
        desc, fl = "", 0.0
        if name in ("fea_conv3x3", "fea_conv1x1", "fea_conv3x3p"):
            n, h, w, cin, cout = args[7:12]
            fl = 2.0 * n * h * w * cin * cout * (1 if name == "fea_conv1x1" else 9)
            desc = f"N{n} {h}x{w} {cin}->{cout}"
        elif name in ("fea_conv3x3_dgrad_bnred", "fea_conv3x3p_dgrad_bnred"):
            n, h, w, cin, cout = args[12:17]
            fl = 2.0 * n * h * w * cin * cout * 9
            desc = f"N{n} {h}x{w} {cin}->{cout}"
        elif name == "fea_conv3x3_wgrad":
            n, h, w, cin, cout = args[7:12]
            fl = 2.0 * n * h * w * cin * cout * 9
            desc = f"N{n} {h}x{w} {cin}->{cout}"
        elif name in ("fea_upconv2x2_fwd", "fea_upconv2x2_bwd_data"):
            n, h, w, cin, cout = args[-7:-2] if name == "fea_upconv2x2_fwd" else args[-7:-2]
            fl = 2.0 * n * h * w * cin * cout * 4
            desc = f"N{n} {h}x{w} {cin}->{cout}"
        elif name == "fea_upconv2x2_wgrad":
            n, h, w, cin, cout = args[7:12]
            fl = 2.0 * n * h * w * cin * cout * 4
            desc = f"N{n} {h}x{w} {cin}->{cout}"
        else:
            desc = " ".join(str(v) for v in ints[-6:])
        rows.append((name, desc, ms, fl))
    tot = sum(r[2] for r in rows)
    print(f"# {per_step} calls/step, sum of call times {tot:.2f} ms")
    for name, desc, ms, fl in rows:
        tf = f"{fl / ms / 1e9:8.0f} TF/s" if fl else " " * 13
        print(f"{name:28s} {desc:32s} {ms:8.3f} ms {tf}")
    agg = {}
    for name, desc, ms, fl in rows:
        d = agg.setdefault(name, [0.0, 0.0, 0])
        d[0] += ms; d[1] += fl; d[2] += 1
    print("# by entry point")
    for name, (ms, fl, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        tf = f"{fl / ms / 1e9:8.0f} TF/s" if fl else ""
        print(f"{name:28s} x{cnt:3d} {ms:8.3f} ms {tf}")


if __name__ == "__main__":
    main()
