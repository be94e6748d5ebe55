#!/usr/bin/env python3
"""Phase cycle stamps of the implicit-GEMM conv kernel, bf16 128-column form (needs a library built with -DIG_STAMPS,
FEA_AMD_LIB=...).  usage: igemm_stamps.py N,H,W,Cin,Cout"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fea_amd import _lib as L
n, h, w, ci, co = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "64,64,64,512,512").split(","))
dt = torch.bfloat16; code = L.dt_code(dt); S = L.stream_ptr()
x = torch.randn(n, h, w, ci, device="cuda").to(dt)
wt = torch.randn(co, ci, 3, 3, device="cuda") / (3 * ci ** 0.5)
wp = torch.empty(9 * co * ci, dtype=dt, device="cuda")
L.call("fea_pack_conv3x3", wt, wp, co, ci, 0, code, S)
y = torch.empty(n, h, w, co, dtype=dt, device="cuda")
nt = L.call("fea_conv_num_tiles", n, h, w)
stats = torch.empty(nt * co * 2, device="cuda")
dbg = torch.zeros(4 * 6, dtype=torch.int64, device="cuda")
L.call("fea_conv3x3", x, ci, wp, None, y, co, stats, n, h, w, ci, co, code, S)       # warm-up without stamps
torch.cuda.synchronize()
os.environ["IG_DBG_PTR"] = str(dbg.data_ptr())
L.call("fea_conv3x3", x, ci, wp, None, y, co, stats, n, h, w, ci, co, code, S)
torch.cuda.synchronize()
d = dbg.view(4, 6).double()
nwg = nt * (co // 128)
nch = ci // 64
names = ["barrier: previous chunk consumed", "stage issue (W tap 0 + A halo)", "barrier + DMA drain", "nine taps (MFMA)", "per-tap barriers", "loop overhead"]
per = d / (nwg * nch)
print(f"{nwg} workgroups x {nch} chunks; cycles per chunk and wave: {per.sum(1).mean().item():.0f} (MFMA alone: 9216; two workgroups share a SIMD)")
for k, nm in enumerate(names):
    print(f"  {nm:34s} {per[:, k].mean().item():8.0f}")
