#!/usr/bin/env python3
"""What a plain streaming kernel reaches on this device (torch elementwise ops on 2 GiB bf16 tensors): the practical HBM
roof the BatchNorm passes are measured against."""
import torch
n = 1 << 30
a = torch.randn(n, device="cuda").to(torch.bfloat16)
b = torch.randn(n, device="cuda").to(torch.bfloat16)
c = torch.empty_like(a)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, fn, nbytes in (("copy 1R+1W", lambda: c.copy_(a), 2 * 2 * n), ("add 2R+1W", lambda: torch.add(a, b, out=c), 3 * 2 * n),
                         ("relu 1R+1W", lambda: torch.relu(a, out=c) if False else torch.clamp_min(a, 0, out=c), 2 * 2 * n),
                         ("sum 1R", lambda: a.sum(), 2 * n)):
    ms = t(fn)
    print(f"{name:12s} {ms:7.3f} ms  {nbytes / ms / 1e9:6.2f} TB/s")
