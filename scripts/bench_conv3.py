"""Times fea_conv3x3 (deep implicit-GEMM kernel) on the deep layers of the bench shape.  python scripts/bench_conv3.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fea_amd import _lib as L  # noqa: E402

DEV = torch.device("cuda:0")
BF = torch.bfloat16
N = int(os.environ.get("N", 64))
code = L.dt_code(BF)
S = torch.cuda.current_stream().cuda_stream


def timeit(fn, reps=10):
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


for (h, ci, co) in ((32, 1024, 1024), (64, 1024, 512), (64, 512, 512), (128, 512, 256), (128, 256, 256), (256, 256, 128)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, h, h, ci, generator=g).to(BF).to(DEV)
    wt = (torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)).to(DEV)
    wp = torch.empty(9 * co * ci, dtype=BF, device=DEV)
    L.call("fea_pack_conv3x3", wt, wp, co, ci, 0, code, S)
    bias = torch.randn(co, generator=g).to(DEV)
    y = torch.empty((N, h, h, co), dtype=BF, device=DEV)
    stats = torch.empty((L.call("fea_conv_num_tiles", N, h, h), co, 2), device=DEV)
    flops = 2.0 * N * h * h * ci * co * 9
    t = timeit(lambda: L.call("fea_conv3x3", x, ci, wp, bias, y, co, stats, N, h, h, ci, co, code, S))
    print(f"conv3x3 N{N} {h}x{h} {ci}->{co}: {t:.3f} ms ({flops / t * 1e-9:.0f} TF/s)", flush=True)

# first layer: fea_conv3x3_first (fp32 NCHW input, 3 channels) at the bench shape
if os.environ.get("FIRST", "1") == "1":
    h = 512
    g = torch.Generator().manual_seed(2)
    x = torch.randn(N, 3, h, h, generator=g).to(DEV)
    wt = (torch.randn(64, 3, 3, 3, generator=g) / 5).to(DEV)
    wp = torch.empty(64 * 64, dtype=BF, device=DEV)
    L.call("fea_pack_conv3x3_first", wt, wp, 64, 3, code, S)
    bias = torch.randn(64, generator=g).to(DEV)
    y = torch.empty((N, h, h, 64), dtype=BF, device=DEV)
    stats = torch.empty((L.call("fea_conv_num_tiles", N, h, h), 64, 2), device=DEV)
    t = timeit(lambda: L.call("fea_conv3x3_first", x, wp, bias, y, 64, stats, N, 3, h, h, 64, code, S))
    t3 = timeit(lambda: L.call("fea_conv3x3_first_bias_relu", x, wp, bias, y, 64, N, 3, h, h, 64, code, S))
    print(f"conv3x3_first_bias_relu (no BN sums): {t3:.3f} ms", flush=True)
    print(f"conv3x3_first N{N} {h}x{h} 3->64: {t:.3f} ms ({(y.numel() * 2 + x.numel() * 4) / t * 1e-9:.2f} TB/s)", flush=True)
