#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 5 90 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_clk -- python3 $R/tools/one_conv.py 2 > /dev/null 2>&1
python3 - $(find $R/gpurun_out/pmc_clk -name "*counter_collection.csv") <<'PY'
import csv, sys
from collections import defaultdict
rows = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "conv_mfma" not in n: continue
        key = (n.split("(")[0].replace("void ", ""), int(r["Grid_Size"]) // 256)
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        rows[key]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key, cs in rows.items():
    a = {c: sum(v) / len(v) for c, v in cs.items()}
    print(key, {k: round(v, 1) for k, v in a.items()}, "eff clock GHz", round(a.get("GRBM_GUI_ACTIVE", 0) / 8 / a["dur_us"] / 1e3, 3))
PY
