#!/usr/bin/env python3
"""Phase cycle stamps of the bf16 weight-gradient kernel (needs a library built with -DWG2_STAMPS, FEA_AMD_LIB=...).
usage: wgrad2_stamps.py N,H,W,Cin,Cout"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fea_amd import _lib as L
n, h, w, ci, co = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "64,64,64,512,512").split(","))
dt = torch.bfloat16; code = L.dt_code(dt); S = L.stream_ptr()
x = torch.randn(n, h, w, ci, device="cuda").to(dt)
dy = torch.randn(n, h, w, co, device="cuda").to(dt)
dw = torch.empty(co, ci, 3, 3, device="cuda")
wsb = L.call("fea_wgrad_ws_bytes", 0, n, h, w, co, ci, code)
ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
dbg = torch.zeros(4096 * 8 * 6, dtype=torch.int64, device="cuda")
os.environ["WG2_DBG_PTR"] = str(dbg.data_ptr())
for _ in range(3):
    L.call("fea_conv3x3_wgrad", dy, co, x, ci, dw, ws, wsb, n, h, w, ci, co, code, S)
torch.cuda.synchronize()
d = dbg.view(-1, 8, 6).double()
used = d.sum((1, 2)) > 0
d = d[used]
tw = 32 if w >= 32 else 16
ntiles = n * ((h + 256 // tw - 1) // (256 // tw)) * ((w + tw - 1) // tw)
nwg = d.shape[0]
blocks = (ci // 64) * (co // 64)
per = ntiles / (nwg / blocks)
names = ["vmcnt(0)", "barrier", "dma issue", "loop overhead", "k-steps 0-7 (+ ring fill)", "k-steps 8-15"]
tot = d.sum(2).mean().item()
print(f"{nwg} workgroups, {per:.1f} tiles each, {tot / per:.0f} cycles per tile (MFMA alone: 5-tap waves 2560, 4-tap waves 2048; per SIMD 4608)")
for k, nm in enumerate(names):
    print(f"  {nm:18s} {d[:, :, k].mean().item() / per:8.0f} cycles/tile  (5-tap waves: {d[:, :4, k].mean().item() / per:8.0f}, 4-tap waves: {d[:, 4:, k].mean().item() / per:8.0f})")
