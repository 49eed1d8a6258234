#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of bench.py: per-step time of every kernel family over the LAST `steps`
steps (the timed region; earlier dispatches include warm-up and the autotuner's trial launches).

    python tools/prof_summary.py <kernel_trace.csv> <steps> <conv launches per step: roofline.launches_per_step of bench.py>

"conv family" = every kernel that does dense-conv MFMA work -- conv_mfma_kernel (implicit GEMM), conv1x1_persist_kernel, conv1x1_wres_kernel,
conv3x3_patch_kernel and the fused conv kernels (stem, Bottleneck, C3k2 block, DWConv+1x1, first conv) -- the set bench.py's
`roofline` object prices since round 2 (round 1 priced the first three only).
"""
import csv
import sys
from collections import defaultdict

path, steps = sys.argv[1], int(sys.argv[2])
per_step = int(sys.argv[3])
CONV = ("conv_mfma_kernel", "conv1x1_persist_kernel", "conv1x1_wres_kernel", "conv3x3_patch_kernel", "stem_fused_kernel", "bneck_fused_kernel", "bneck_fused_wide_kernel", "c3k2_fused_kernel", "dwpw_fused_kernel", "conv_first_mfma_kernel", "chain1x1_kernel")
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
conv = [r for r in rows if any(k in r["Kernel_Name"] for k in CONV)]
tail = conv[-steps * per_step:]
t0 = int(tail[0]["Start_Timestamp"])
fam = defaultdict(lambda: [0, 0])
for r in rows:
    if int(r["Start_Timestamp"]) < t0:
        continue
    n = r["Kernel_Name"]
    key = n.split("<")[0].split("(")[0].replace("void ", "")
    if key.startswith("_Z"):
        for k in CONV + ("dwconv3x3", "sppf_pool", "attention", "decode", "nms_filter",
                  "nms_sort", "nms_greedy", "raw_nchw"):
            if k in key:
                key = k if k.endswith("_kernel") else k + "_kernel"
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    fam[key][0] += 1
    fam[key][1] += d
print(f"# {path}: last {steps} steps ({per_step} conv_mfma launches per step)")
print(f"{'kernel family':40s} {'launches/step':>14s} {'avg us':>10s} {'ms/step':>10s}")
tot = 0
for k, (c, d) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:40s} {c / steps:14.1f} {d / c / 1e3:10.2f} {d / steps / 1e6:10.4f}")
    tot += d
print(f"{'TOTAL device time':40s} {'':14s} {'':10s} {tot / steps / 1e6:10.4f}")
c = sum(fam[k][0] for k in CONV if k in fam)
d = sum(fam[k][1] for k in CONV if k in fam)
print(f"conv family ({' + '.join(k for k in CONV if k in fam)}): {c / steps:.0f} launches/step, average launch {d / c / 1e3:.2f} us, {d / steps / 1e6:.4f} ms/step")
