#!/usr/bin/env python3
"""Where a step's wall time goes BETWEEN kernels: from a rocprofv3 --kernel-trace CSV of bench.py, over the last `steps` timed
steps, per hardware queue: busy time, idle gaps between consecutive dispatches (the dependent-launch boundary), and the union of
busy intervals over all queues against the window's wall time.

    python tools/trace_gaps.py <kernel_trace.csv> <steps> <conv launches per step: roofline.launches_per_step of bench.py>

The window is found as tools/prof_summary.py finds it, by counting conv-family launches from the end of the trace: the bench's three
serial profile passes come last and are left out, the `steps` steps in front of them are the window.
"""
import csv
import sys
from collections import defaultdict

path, steps, per_step = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
CONV = ("conv_mfma_kernel", "conv1x1_persist_kernel", "conv1x1_wres_kernel", "conv3x3_patch_kernel", "stem_fused_kernel", "bneck_fused_kernel", "bneck_fused_wide_kernel", "c3k2_fused_kernel", "dwpw_fused_kernel", "conv_first_mfma_kernel", "chain1x1_kernel")
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
conv = [r for r in rows if any(k in r["Kernel_Name"] for k in CONV)]
w0 = int(conv[-(steps + 3) * per_step]["Start_Timestamp"])
w1 = int(conv[-3 * per_step]["Start_Timestamp"])
win = [r for r in rows if w0 <= int(r["Start_Timestamp"]) < w1]
qkey = "Queue_Id" if "Queue_Id" in win[0] else "Stream_Id"
t0, t1 = int(win[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in win)
wall = t1 - t0
by_q = defaultdict(list)
for r in win:
    by_q[r[qkey]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("<")[0].split("(")[0].replace("void ", "")))
print(f"# {path}: {steps} timed steps ({len(win) / steps:.1f} dispatches per step), window {wall / 1e6:.3f} ms = {wall / steps / 1e6:.4f} ms per step")
ivs = sorted((s, e) for r in by_q.values() for s, e, _ in r)
busy, cs, ce = 0, ivs[0][0], ivs[0][1]
for s, e in ivs[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"union of busy intervals over all queues: {busy / steps / 1e6:.4f} ms per step; no kernel running on any queue: {(wall - busy) / steps / 1e6:.4f} ms per step")
for q, lst in sorted(by_q.items(), key=lambda kv: -len(kv[1])):
    lst.sort()
    b = sum(e - s for s, e, _ in lst)
    gaps = [(lst[i + 1][0] - lst[i][1], lst[i][2], lst[i + 1][2]) for i in range(len(lst) - 1)]
    small = [g for g in gaps if 0 <= g[0] < 20000]
    print(f"queue {q}: {len(lst) / steps:.1f} dispatches/step, busy {b / steps / 1e6:.4f} ms/step, "
          f"{len(small) / steps:.1f} gaps < 20 us per step summing {sum(g[0] for g in small) / steps / 1e6:.4f} ms/step "
          f"(median {sorted(g[0] for g in small)[len(small) // 2] / 1e3 if small else 0:.2f} us)")
    if len(lst) / steps > 20:
        after = defaultdict(lambda: [0, 0])
        for g, a, _ in small:
            after[a][0] += 1
            after[a][1] += g
        for a, (c, d) in sorted(after.items(), key=lambda kv: -kv[1][1])[:8]:
            print(f"    after {a:36s} {c / steps:6.1f} gaps/step, avg {d / c / 1e3:6.2f} us")
