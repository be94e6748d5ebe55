#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one pass each, as
MI355X_MICROARCH.md prescribes) -> profiles/r3_pmc_hbm_traffic.json, which bench.py reads for `roofline.traffic`.

    python scripts/pmc_traffic.py FETCH.csv WRITE.csv "train batch 64 512x512 bf16" STEPS [out.json]

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB and, on gfx950, FETCH_SIZE reports half of the
bytes of a 16-byte-per-lane streaming read.  The JSON carries a digest of the kernel sources: bench.py reports
`traffic: null` when the sources have changed since the passes were taken.
"""
import collections
import csv
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fea_amd", "csrc")


def source_digest() -> str:
    h = hashlib.sha256()
    for fn in sorted(os.listdir(CSRC)):
        if fn.endswith((".hip", ".h")):
            with open(os.path.join(CSRC, fn), "rb") as f:
                h.update(fn.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def short_name(kernel_name: str) -> str:
    """Stable short key for the conv kernel instantiations; other kernels keep their (truncated) name."""
    m = re.search(r"igemm_kernelIDF16bLi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", kernel_name)
    if m:
        return "igemm_kernel<bf16,%s,%s,%s,%s>" % m.groups()
    m = re.search(r"igemm_kernel<.*?,\s*(\d+),\s*(\d+),\s*(\d+)>", kernel_name)
    if m and "int, E" in kernel_name:
        return "igemm_kernel<bf16,?,%s,%s,%s>" % m.groups()
    return kernel_name.split("(")[0][:80]


def per_kernel(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = short_name(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


def main():
    fetch_csv, write_csv, workload, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(ROOT, "profiles", "r3_pmc_hbm_traffic.json")
    ft, fc = per_kernel(fetch_csv, "FETCH_SIZE")
    wt, wc = per_kernel(write_csv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(ft) | set(wt), key=lambda k: -(2 * ft.get(k, 0) + wt.get(k, 0))):
        n = max(fc.get(k, 0), wc.get(k, 0), 1)
        rd, wr = 2 * ft.get(k, 0.0) * 1024, wt.get(k, 0.0) * 1024
        kernels[k] = {"launches_per_step": n / steps, "hbm_bytes_per_launch": (rd + wr) / n,
                      "read_bytes_per_launch": rd / n, "write_bytes_per_launch": wr / n,
                      "hbm_bytes_per_step": (rd + wr) / steps}
    rec = {"workload": workload, "steps_profiled": steps, "source_digest": source_digest(),
           "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes, separate --pmc passes",
           "total_bytes_per_step": sum(v["hbm_bytes_per_step"] for v in kernels.values()), "kernels": kernels}
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(f"wrote {out}: {rec['total_bytes_per_step'] / 1e9:.1f} GB per step over {len(kernels)} kernels")


if __name__ == "__main__":
    main()
