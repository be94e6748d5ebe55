#!/usr/bin/env python3
"""Phase cycle stamps of the persistent first-layer kernel (needs a library built with -DFC_STAMPS, FEA_AMD_LIB=...)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fea_amd import _lib as L
N, h = 64, 512
code = L.dt_code(torch.bfloat16); S = L.stream_ptr()
x = torch.randn(N, 3, h, h, device="cuda")
wt = torch.randn(64, 3, 3, 3, device="cuda") / 5
wp = torch.empty(64 * 64, dtype=torch.bfloat16, device="cuda")
L.call("fea_pack_conv3x3_first", wt, wp, 64, 3, code, S)
bias = torch.randn(64, device="cuda")
y = torch.empty((N, h, h, 64), dtype=torch.bfloat16, device="cuda")
rows = L.call("fea_conv3x3_first_rows", N, 3, h, h, 64, code)
stats = torch.empty((rows, 64, 2), device="cuda")
nwg = rows // 4
dbg = torch.zeros(nwg * 4 * 6, dtype=torch.int64, device="cuda")
os.environ["FC_DBG_PTR"] = str(dbg.data_ptr())
for _ in range(3):
    L.call("fea_conv3x3_first", x, wp, bias, y, 64, stats, N, 3, h, h, 64, code, S)
torch.cuda.synchronize()
d = dbg.view(nwg, 4, 6).double()
ntile = N * (h // 8) * (h // 32) / nwg
print(f"{nwg} workgroups, {ntile:.1f} tiles each, {d.sum(2).mean().item() / ntile:.0f} cycles (s_memtime ticks) per tile")
for k, nm in enumerate(["halo wait + LDS write", "barrier", "next halo issue", "gather + MFMA + stores"]):
    print(f"  {nm:24s} {d[:, :, k].mean().item() / ntile:8.0f} per tile  (min wave {d[:, :, k].min().item() / ntile:8.0f}, max {d[:, :, k].max().item() / ntile:8.0f})")
