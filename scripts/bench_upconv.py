"""Times the transposed-convolution kernels at the four decoder levels of the bench shape (batch 64, 512^2), persistent GEMM
(upgemm.hip) against the implicit-GEMM path (FEA_AMD_UPGEMM=0).  python scripts/bench_upconv.py [fwd|bwd] """
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fea_amd import _lib as L  # noqa: E402

DEV = torch.device("cuda:0")
BF = torch.bfloat16
what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
N = int(os.environ.get("N", 64))
code = L.dt_code(BF)
S = torch.cuda.current_stream().cuda_stream


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (h, ci, co) in ((32, 1024, 512), (64, 512, 256), (128, 256, 128), (256, 128, 64)):
    w = h
    g = torch.Generator().manual_seed(1)
    wt = (torch.randn(ci, co, 2, 2, generator=g) / ci ** 0.5).to(DEV)
    flops = 2.0 * N * h * w * ci * co * 4
    if what == "fwd":
        x = torch.randn(N, h, w, ci, generator=g).to(BF).to(DEV)
        bias = torch.randn(co, generator=g).to(DEV)
        wp = torch.empty(4 * ci * co, dtype=BF, device=DEV)
        L.call("fea_pack_upconv2x2", wt, wp, ci, co, 0, code, S)
        cat = torch.empty((N, 2 * h, 2 * w, 2 * co), dtype=BF, device=DEV)
        res = {}
        for on in ("0", "1"):
            os.environ["FEA_AMD_UPGEMM"] = on
            res[on] = timeit(lambda: L.call("fea_upconv2x2_fwd", x, ci, wp, bias, cat, 2 * co, N, h, w, ci, co, code, S))
        if os.environ.get("UPG_STAMPS"):
            dbg = torch.zeros(4096, dtype=torch.int64, device=DEV)
            os.environ["UPG_DBG_PTR"] = hex(dbg.data_ptr())
            L.call("fea_upconv2x2_fwd", x, ci, wp, bias, cat, 2 * co, N, h, w, ci, co, code, S)
            torch.cuda.synchronize()
            cyc = dbg[dbg > 0].double()
            print(f"   workgroup shader cycles: mean {cyc.mean().item():.0f} max {cyc.max().item():.0f} -> clock {cyc.max().item() / res['1'] * 1e-6:.2f} GHz "
                  f"(if the longest workgroup spans the launch)")
        print(f"fwd N{N} {h}x{w} {ci}->{co}: igemm {res['0']:.3f} ms ({flops / res['0'] * 1e-9:.0f} TF/s)   persistent {res['1']:.3f} ms "
              f"({flops / res['1'] * 1e-9:.0f} TF/s)", flush=True)
    elif what == "wgrad":
        x = torch.randn(N, h, w, ci, generator=g).to(BF).to(DEV)
        dy = torch.randn(N, 2 * h, 2 * w, 2 * co, generator=g).to(BF).to(DEV)
        dW = torch.empty((ci, co, 2, 2), device=DEV)
        res = {}
        for on in ("0", "1"):
            os.environ["FEA_AMD_WU256"] = on
            nbytes = L.load().fea_wgrad_ws_bytes(2, N, h, w, ci, co, code)
            ws = torch.empty(max(nbytes, 16) // 4, device=DEV)
            res[on] = timeit(lambda: L.call("fea_upconv2x2_wgrad", x, ci, dy, 2 * co, dW, ws, nbytes, N, h, w, ci, co, code, S))
        print(f"wgrad N{N} {h}x{w} {ci}->{co}: 128-ch {res['0']:.3f} ms ({flops / res['0'] * 1e-9:.0f} TF/s)   256-ch {res['1']:.3f} ms "
              f"({flops / res['1'] * 1e-9:.0f} TF/s)", flush=True)
    else:
        dy = torch.randn(N, 2 * h, 2 * w, 2 * co, generator=g).to(BF).to(DEV)
        y = torch.randn(N, h, w, ci, generator=g).to(BF).to(DEV)
        coef = [torch.randn(ci, generator=g).to(DEV) for _ in range(3)] + [(torch.rand(ci, generator=g) + 0.5).to(DEV)]
        wp = torch.empty(4 * ci * co, dtype=BF, device=DEV)
        L.call("fea_pack_upconv2x2", wt, wp, ci, co, 1, code, S)
        dx = torch.empty((N, h, w, ci), dtype=BF, device=DEV)
        part = torch.empty((max(4096, N * h * w // 256), ci, 2), device=DEV)
        res = {}
        for on in ("0", "1"):
            os.environ["FEA_AMD_UPGEMM"] = on
            res[on] = timeit(lambda: L.call("fea_upconv2x2_bwd_data_bnred", dy, 2 * co, wp, dx, ci, y, ci, *coef, part, N, h, w, ci, co, code, S))
        print(f"bwd N{N} {h}x{w} {ci}->{co}: igemm {res['0']:.3f} ms ({flops / res['0'] * 1e-9:.0f} TF/s)   persistent {res['1']:.3f} ms "
              f"({flops / res['1'] * 1e-9:.0f} TF/s)", flush=True)
