#!/usr/bin/env python3
"""What happens to the persistent kernels (one workgroup per CU, static tile ranges) when some CUs are taken by another
stream's kernels -- the stand-in for RCCL's channel kernels during the overlapped gradient all-reduce.

K workgroups of scripts/hold_cus.hip (512 threads, 80 KB of LDS each: one per CU, spinning for ~5 ms) on a second stream
hold K CUs: a persistent workgroup (90-152 KB of LDS) cannot be placed beside one and waits for a peer to finish.  Prints the kernel time alone and with 8 / 16 / 32 CUs held, and the ratio a perfectly
work-conserving scheduler would give (256 / (256 - K)).
"""
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
    n, h, w, c = 64, 512, 512, 64
    x = torch.randn(n, h, w, c, device=dev, generator=g).to(dt)
    dy = torch.randn(n, h, w, c, device=dev, generator=g).to(dt)
    y = torch.empty_like(x)
    wt = torch.randn(c, c, 3, 3, device=dev) / 24
    wpf = torch.empty(9 * c * c, dtype=dt, device=dev)
    L.call("fea_pack_conv3x3_frag", wt, wpf, c, c, 0, code, L.stream_ptr())
    bias = torch.zeros(c, device=dev)
    dW = torch.empty(c, c, 3, 3, device=dev)
    bufs = {}

    def sized_for_split():
        """stats rows and the weight-gradient workspace depend on fea_set_persistent_split"""
        rows = L.call("fea_conv3x3p_rows", n, h, w, c, c, code)
        wsb = L.load().fea_wgrad_ws_bytes(0, n, h, w, c, c, code)
        bufs["stats"] = torch.empty(rows * c * 2, device=dev)
        bufs["wsb"], bufs["ws"] = wsb, torch.empty(max(wsb, 16) // 4, device=dev)
    # a deep layer through the one-tile-per-workgroup kernel, for comparison (dynamic by construction)
    n2, h2, w2, c2 = 64, 64, 64, 512
    x2 = torch.randn(n2, h2, w2, c2, device=dev, generator=g).to(dt)
    y2 = torch.empty_like(x2)
    wp2 = torch.empty(9 * c2 * c2, dtype=dt, device=dev)
    L.call("fea_pack_conv3x3", torch.randn(c2, c2, 3, 3, device=dev) / 60, wp2, c2, c2, 0, code, L.stream_ptr())

    kernels = {
        "c3p 64->64 @512^2 (persistent)": lambda: L.call("fea_conv3x3p", x, c, wpf, bias, y, c, bufs["stats"], n, h, w, c, c, code, L.stream_ptr()),
        "wgrad 64->64 @512^2 (persistent)": lambda: L.call("fea_conv3x3_wgrad", dy, c, x, c, dW, bufs["ws"], bufs["wsb"], n, h, w, c, c, code, L.stream_ptr()),
        "igemm 512->512 @64^2 (one tile per workgroup)": lambda: L.call("fea_conv3x3", x2, c2, wp2, None, y2, c2, None, n2, h2, w2, c2, c2, code, L.stream_ptr()),
    }
    import ctypes
    hold = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhold_cus.so"))
    hold.hold_cus.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
    sink = torch.zeros(1, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream()
    HOLD = 500_000                # s_memtime ticks (100 MHz): 5 ms, longer than any of the kernels

    def timed(fn, k, reps=5):
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            fn()                                    # no idle-clock ramp inside the timed launch
            if k:
                hold.hold_cus(k, HOLD, sink.data_ptr(), side.cuda_stream)
                torch.cuda._sleep(200_000)          # let the holders take their CUs first
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        return ts[len(ts) // 2]

    for split in (1, 2, 4, 8):
        L.call("fea_set_persistent_split", split)
        sized_for_split()
        print(f"-- fea_set_persistent_split({split})")
        for name, fn in kernels.items():
            if split > 1 and "persistent" not in name:
                continue
            for _ in range(3):
                fn()
            base = timed(fn, 0)
            line = f"{name}: alone {base:.3f} ms"
            for k in (8, 16, 32):
                t = timed(fn, k)
                line += f" | {k} CUs held {t:.3f} ms = {t / base:.2f}x (work-conserving {256 / (256 - k):.2f}x)"
            print(line, flush=True)


if __name__ == "__main__":
    main()
