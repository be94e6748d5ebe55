#!/usr/bin/env python3
"""Micro-benchmark of single conv layers through the C ABI on random data.

    python scripts/bench_conv.py [--op conv|wgrad] [--shapes all|deep|c64|<n,h,w,ci,co>] [--iters 20] [--entry fea_conv3x3]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fea_amd import _lib as L  # noqa: E402

LAYERS = {  # name: (N, H, W, Cin, Cout)
    "c64": (64, 512, 512, 64, 64),
    "c128_64": (64, 512, 512, 128, 64),
    "c64_128": (64, 256, 256, 64, 128),
    "c128": (64, 256, 256, 128, 128),
    "c256_128": (64, 256, 256, 256, 128),
    "c256": (64, 128, 128, 256, 256),
    "c512_256": (64, 128, 128, 512, 256),
    "c512": (64, 64, 64, 512, 512),
    "c1024_512": (64, 64, 64, 1024, 512),
    "c1024": (64, 32, 32, 1024, 1024),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--op", default="conv")
    ap.add_argument("--shapes", default="all")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--entry", default=None)
    ap.add_argument("--stats", type=int, default=1)
    a = ap.parse_args()
    if a.shapes == "all":
        shapes = list(LAYERS.items())
    elif a.shapes in LAYERS:
        shapes = [(a.shapes, LAYERS[a.shapes])]
    elif "," in a.shapes and a.shapes.split(",")[0] in LAYERS:
        shapes = [(s, LAYERS[s]) for s in a.shapes.split(",")]
    else:
        shapes = [("custom", tuple(int(v) for v in a.shapes.split(",")))]
    dt = torch.bfloat16
    code = L.dt_code(dt)
    S = L.stream_ptr()
    for name, (n, h, w, ci, co) in shapes:
        g = torch.Generator(device="cuda").manual_seed(0)
        x = torch.randn(n, h, w, ci, device="cuda", generator=g).to(dt)
        wt = (torch.randn(co, ci, 3, 3, device="cuda", generator=g) / (3 * ci ** 0.5))
        fl = 2.0 * n * h * w * ci * co * 9
        if a.op == "conv":
            entry = a.entry or "fea_conv3x3"
            wp = torch.empty(9 * co * ci, dtype=dt, device="cuda")
            L.call("fea_pack_conv3x3_frag" if entry == "fea_conv3x3p" else "fea_pack_conv3x3", wt, wp, co, ci, 0, code, S)
            y = torch.empty(n, h, w, co, dtype=dt, device="cuda")
            bias = torch.zeros(co, device="cuda")
            nt = L.call("fea_conv_num_tiles", n, h, w)
            stats = torch.empty(nt * co * 2, device="cuda") if a.stats else None
            fn = lambda: L.call(entry, x, ci, wp, bias, y, co, stats, n, h, w, ci, co, code, S)  # noqa: E731
        elif a.op == "bnred":
            entry = a.entry or "fea_conv3x3_dgrad_bnred"
            wp = torch.empty(9 * co * ci, dtype=dt, device="cuda")
            L.call("fea_pack_conv3x3_frag" if entry.startswith("fea_conv3x3p") else "fea_pack_conv3x3", wt, wp, co, ci, 0, code, S)
            y = torch.empty(n, h, w, co, dtype=dt, device="cuda")
            yb = torch.randn(n, h, w, co, device="cuda", generator=g).to(dt)
            cf = [torch.randn(co, device="cuda") for _ in range(4)]
            nt = L.call("fea_conv_num_tiles", n, h, w)
            part = torch.empty(nt * co * 2, device="cuda")
            fn = lambda: L.call(entry, x, ci, wp, y, co, yb, co, *cf, part, n, h, w, ci, co, code, S)  # noqa: E731
        else:
            entry = a.entry or "fea_conv3x3_wgrad"
            dy = torch.randn(n, h, w, co, device="cuda", generator=g).to(dt)
            dW = torch.empty(co, ci, 3, 3, device="cuda")
            wsb = L.load().fea_wgrad_ws_bytes(0, n, h, w, co, ci, code)
            ws = torch.empty(wsb // 4, device="cuda")
            fn = lambda: L.call(entry, dy, co, x, ci, dW, ws, wsb, n, h, w, ci, co, code, S)  # noqa: E731
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        print(f"{entry:22s} {name:10s} N{n} {h}x{w} {ci}->{co}: {ms:7.3f} ms  {fl / ms / 1e9:7.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
