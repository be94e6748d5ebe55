#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if pat and pat not in r["Kernel_Name"]:
        continue
    k = f'{r["Kernel_Name"][:70]} grid={r["Grid_Size"]} wg={r["Workgroup_Size"]} vgpr={r["VGPR_Count"]}+{r["Accum_VGPR_Count"]} lds={r["LDS_Block_Size"]}'
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v) / len(v):.4g}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        mf = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(d["SQ_VALU_MFMA_BUSY_CYCLES"])
        ga = sum(d["GRBM_GUI_ACTIVE"]) / len(d["GRBM_GUI_ACTIVE"])
        print(f"   -> MFMA busy / (1024 SIMDs x cycles) = {mf / (1024 * ga / 8):.3f}")
