#!/usr/bin/env python3
"""Do an MFMA-bound kernel (conv3x3 weight gradient / conv) and an HBM-bound kernel (BatchNorm
backward apply) overlap when launched on two streams?  Prints serial vs concurrent time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fea_amd import _lib as L  # noqa: E402


def main():
    dt = torch.bfloat16
    code = L.dt_code(dt)
    n, h, w, c = 64, 256, 256, 128
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    dy = torch.randn(n, h, w, c, device=dev, generator=g).to(dt)
    x = torch.randn(n, h, w, c, device=dev, generator=g).to(dt)
    da = torch.randn(n, h, w, c, device=dev, generator=g).to(dt)
    y = torch.randn(n, h, w, c, device=dev, generator=g).to(dt)
    out = torch.empty_like(y)
    coef = [torch.randn(c, device=dev) for _ in range(4)]
    bncoef = torch.randn(2 * c, device=dev)
    nblk = L.call("fea_bn_bwd_blocks", n, h, w, c)
    part = torch.empty(nblk * c * 2, device=dev)
    wsb = L.load().fea_wgrad_ws_bytes(0, n, h, w, c, c, code)
    ws = torch.empty(wsb // 4, device=dev)
    dW = torch.empty(c, c, 3, 3, device=dev)
    wp = torch.empty(9 * c * c, dtype=dt, device=dev)
    L.call("fea_pack_conv3x3", torch.randn(c, c, 3, 3, device=dev), wp, c, c, 0, code, L.stream_ptr())
    yo = torch.empty_like(y)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def wgrad(s):
        L.call("fea_conv3x3_wgrad", dy, c, x, c, dW, ws, wsb, n, h, w, c, c, code, s.cuda_stream)

    def conv(s):
        L.call("fea_conv3x3", x, c, wp, None, yo, c, None, n, h, w, c, c, code, s.cuda_stream)

    def apply(s):
        L.call("fea_bn_relu_bwd_apply", da, c, None, y, *coef, bncoef, out, part, n, h, w, c, code, s.cuda_stream)

    sc, sh = torch.randn(c, device=dev), torch.randn(c, device=dev)

    def bnfwd(s):       # 28 VGPRs: several waves fit beside the MFMA kernels
        L.call("fea_bn_relu_pool", y, sc, sh, out, c, None, n, h, w, c, code, s.cuda_stream)

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

    for sname, sf in (("bn_bwd_apply", apply), ("bn_relu_fwd", bnfwd)):
        for name, f in (("wgrad", wgrad), ("conv", conv)):
            timed(f, sf, 3)
            a, b, ab = timed(f, None), timed(None, sf), timed(f, sf)
            print(f"{name} + {sname}: alone {a:.3f} ms / {b:.3f} ms   serial {a + b:.3f}   concurrent {ab:.3f}   "
                  f"hidden {(a + b - ab) / b * 100:.0f} % of the streaming kernel")


if __name__ == "__main__":
    main()
