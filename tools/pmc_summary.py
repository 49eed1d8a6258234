#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per (kernel, grid) mean counter values over dispatches."""
import csv
import sys
from collections import defaultdict

rows = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name", "")
            short = name.split("(")[0].replace("void ", "")[:60]
            key = (short, r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", r.get("Arch_VGPR_Count", "")))
            rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(rows.items()):
    n = max(len(v) for v in cs.values())
    print(f"{key[0]} grid={key[1]} lds={key[2]} vgpr={key[3]} dispatches={n}")
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v) / len(v):16.1f}")
