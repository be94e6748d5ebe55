#!/usr/bin/env python3
"""Co-run experiment, round 3: a matrix-bound conv of the DEEP levels (little HBM traffic of its own) beside an HBM-bound
BatchNorm pass of level 0, on two streams.  Prints alone / serial / concurrent times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fea_amd import _lib as L  # noqa: E402


def main():
    dt = torch.bfloat16
    code = L.dt_code(dt)
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    # deep conv: 512 -> 512 @ 64x64, batch 64
    n, h, w, c = 64, 64, 64, 512
    x = torch.randn(n, h, w, c, device=dev, generator=g).to(dt)
    dy = torch.randn(n, h, w, c, device=dev, generator=g).to(dt)
    wp = torch.empty(9 * c * c, dtype=dt, device=dev)
    L.call("fea_pack_conv3x3", torch.randn(c, c, 3, 3, device=dev) / 60, wp, c, c, 0, code, L.stream_ptr())
    yo = torch.empty_like(x)
    wsb = L.load().fea_wgrad_ws_bytes(0, n, h, w, c, c, code)
    ws = torch.empty(max(wsb, 16) // 4, device=dev)
    dW = torch.empty(c, c, 3, 3, device=dev)
    # level-0 BatchNorm passes: 64 channels @ 512x512
    n0, h0, w0, c0 = 64, 512, 512, 64
    y = torch.randn(n0, h0, w0, c0, device=dev, generator=g).to(dt)
    da = torch.randn(n0, h0, w0, c0, device=dev, generator=g).to(dt)
    out = torch.empty_like(y)
    coef = [torch.randn(c0, device=dev) for _ in range(4)]
    bncoef = torch.randn(2 * c0, device=dev)
    nblk = L.call("fea_bn_bwd_blocks", n0, h0, w0, c0)
    part = torch.empty(nblk * c0 * 2, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def conv(s):
        L.call("fea_conv3x3", x, c, wp, None, yo, c, None, n, h, w, c, c, code, s.cuda_stream)

    def wgrad(s):
        L.call("fea_conv3x3_wgrad", dy, c, x, c, dW, ws, wsb, n, h, w, c, c, code, s.cuda_stream)

    def apply(s):
        L.call("fea_bn_relu_bwd_apply", da, c0, None, y, *coef, bncoef, out, part, n0, h0, w0, c0, code, s.cuda_stream)

    def bnfwd(s):
        L.call("fea_bn_relu_pool", y, coef[0], coef[1], out, c0, None, n0, h0, w0, c0, code, s.cuda_stream)

    def copy(s):
        with torch.cuda.stream(s):
            out.copy_(y)

    def timed(fa, fb, reps=10):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ev = torch.cuda.Event()
        ev.record()
        s1.wait_event(ev); s2.wait_event(ev)
        for _ in range(reps):
            if fa: fa(s1)
            if fb: fb(s2)
        d1, d2 = torch.cuda.Event(), torch.cuda.Event()
        d1.record(s1); d2.record(s2)
        torch.cuda.current_stream().wait_event(d1)
        torch.cuda.current_stream().wait_event(d2)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for sname, sf in (("bn_bwd_apply@512", apply), ("bn_relu_fwd@512", bnfwd), ("torch copy 2 GiB", copy)):
        for name, f in (("conv 512->512@64", conv), ("wgrad 512->512@64", wgrad)):
            timed(f, sf, 3)
            a, b, ab = timed(f, None), timed(None, sf), timed(f, sf)
            print(f"{name} + {sname}: alone {a:.3f} / {b:.3f} ms   serial {a + b:.3f}   concurrent {ab:.3f}   "
                  f"hidden {(a + b - ab) / min(a, b) * 100:.0f} % of the shorter kernel", flush=True)


if __name__ == "__main__":
    main()
