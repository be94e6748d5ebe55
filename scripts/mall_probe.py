#!/usr/bin/env python3
"""Does the Infinity Cache (256 MB, memory side) serve a consumer that reads what the kernel before it wrote?

    python scripts/mall_probe.py

For tensor sizes from 16 MB to 2 GB (bf16): producer `b = a * 2` then consumer `c = b + 1`, timed with HIP events over
many repetitions of the PAIR; effective rate of the consumer = 2 x bytes / time.  A rate well above the ~5.5 TB/s of a
2 GB stream at sizes below the cache says that a batch-chunked schedule (producer chunk -> consumer chunk) would turn
HBM passes into cache hits; a flat curve says it would not."""
import torch

dev = "cuda"
print("size_MB  producer_us  consumer_us  consumer_TB/s  (consumer alone on a cold tensor: TB/s)")
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * (1 << 20) // 2
    a = torch.randn(n, device=dev, dtype=torch.bfloat16)
    b = torch.empty_like(a)
    c = torch.empty_like(a)
    # something large in between for the cold case
    junk = torch.empty(1 << 30, device=dev, dtype=torch.bfloat16)
    reps = max(5, min(200, 8192 // mb))
    for _ in range(3):
        torch.mul(a, 2, out=b); torch.add(b, 1, out=c)
    torch.cuda.synchronize()
    tp = tc = 0.0
    for _ in range(reps):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(); torch.mul(a, 2, out=b); e1.record(); torch.add(b, 1, out=c); e2.record()
        torch.cuda.synchronize()
        tp += e0.elapsed_time(e1); tc += e1.elapsed_time(e2)
    tp, tc = tp / reps * 1e3, tc / reps * 1e3
    cold = 0.0
    for _ in range(5):
        junk.zero_()
        e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
        e1.record(); torch.add(b, 1, out=c); e2.record(); torch.cuda.synchronize()
        cold += e1.elapsed_time(e2)
    cold = cold / 5 * 1e3
    by = 2.0 * n * 2
    print(f"{mb:6d}  {tp:10.1f}  {tc:10.1f}  {by / tc / 1e6:10.2f}      {by / cold / 1e6:10.2f}", flush=True)
    del a, b, c, junk
