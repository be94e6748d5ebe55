#!/usr/bin/env python3
"""Phase cycle stamps of the persistent conv kernel (needs a library built with -DC3P_STAMPS, FEA_AMD_LIB=...)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fea_amd import _lib as L
n, h, w, ci, co = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "64,512,512,64,64").split(","))
dt = torch.bfloat16; code = L.dt_code(dt); S = L.stream_ptr()
x = torch.randn(n, h, w, ci, device="cuda").to(dt)
wt = torch.randn(co, ci, 3, 3, device="cuda") / (3 * ci ** 0.5)
wp = torch.empty(9 * co * ci, dtype=dt, device="cuda")
L.call("fea_pack_conv3x3_frag", wt, wp, co, ci, 0, code, S)
y = torch.empty(n, h, w, co, dtype=dt, device="cuda")
rows = L.call("fea_conv3x3p_rows", n, h, w, ci, co, code)
stats = torch.empty(rows * co * 2, device="cuda")
dbg = torch.zeros(512 * 4 * 5, dtype=torch.int64, device="cuda")
os.environ["C3P_DBG_PTR"] = str(dbg.data_ptr())
for _ in range(3):
    L.call("fea_conv3x3p", x, ci, wp, None, y, co, stats, n, h, w, ci, co, code, S)
torch.cuda.synchronize()
d = dbg.view(-1, 4, 5).double()
nwg = rows // 2 * (co // 64)
d = d[:nwg]
ntile = n * (h // 8) * (w // 32) / (rows // 2)
names = ["barrier wait", "dma issue + mfma", "vmcnt(0)", "epilogue", "loop overhead"]
tot = d.sum(2).mean().item()
print(f"{nwg} workgroups, {ntile:.0f} tiles each, {tot / ntile:.0f} cycles per tile")
for k, nm in enumerate(names):
    print(f"  {nm:18s} {d[:, :, k].mean().item() / ntile:8.0f} cycles/tile  (wave 0: {d[:, 0, k].mean().item() / ntile:8.0f}, wave 3: {d[:, 3, k].mean().item() / ntile:8.0f})")
