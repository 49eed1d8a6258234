#!/bin/bash
# L2 hit / miss counts of a producer -> consumer pair of 1x1 conv launches (tools/l2_boundary.py) -> gpurun_out/<tag>_l2_boundary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_l2b -- python3 $R/tools/l2_boundary.py > $R/gpurun_out/pmc_l2b.out 2> $R/gpurun_out/pmc_l2b.err; echo "rc=$?"
python3 - $R <<'PY' | tee $R/gpurun_out/${TAG}_l2_boundary.txt
import csv, glob, os, sys
R = sys.argv[1]
print(open(f"{R}/gpurun_out/pmc_l2b.out").read().strip())
f = sorted(glob.glob(f"{R}/gpurun_out/pmc_l2b/*/*counter_collection.csv"), key=os.path.getmtime)[-1]
per = {}
for r in csv.DictReader(open(f)):
    d = per.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "t0": int(r["Start_Timestamp"])})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
seq = [d for d in sorted(per.values(), key=lambda d: d["t0"]) if "conv" in d["name"]][-6:]
for i, d in enumerate(seq):
    h, m = d.get("TCC_HIT_sum", 0), d.get("TCC_MISS_sum", 0)
    print(f"{'producer' if i % 2 == 0 else 'consumer'}: hits {h / 1e3:8.1f} k  misses {m / 1e3:8.1f} k  hit rate {100 * h / max(h + m, 1):5.1f} %   {d['name'][:60]}")
PY
