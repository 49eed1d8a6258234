#!/bin/bash
# memory-side counters for tools/one_conv.py:  tools/pmc_run2.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum --output-format csv -d $R/gpurun_out/pmc_${T}_c -- python3 $R/tools/one_conv.py 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $R/gpurun_out/pmc_${T}_d -- python3 $R/tools/one_conv.py 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_BUSY_avr TCC_TAG_STALL_sum TCC_READ_sum GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_${T}_e -- python3 $R/tools/one_conv.py 2 > /dev/null 2>&1
python3 - $(find $R/gpurun_out/pmc_${T}_c $R/gpurun_out/pmc_${T}_d $R/gpurun_out/pmc_${T}_e -name "*counter_collection.csv") <<'PY'
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
    print(key, " ".join(f"{c}={v:.4g}" for c, v in sorted(a.items())))
PY
