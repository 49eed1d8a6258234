#!/bin/bash
# HBM traffic of the conv kernels via FETCH_SIZE / WRITE_SIZE (separate passes, as the guide prescribes).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 5 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/tools/one_conv.py 2 > /dev/null 2>&1; echo fetch rc=$?
timeout -k 5 100 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/tools/one_conv.py 2 > /dev/null 2>&1; echo write rc=$?
python3 - $(find $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write -name "*counter_collection.csv") <<'PY'
import csv, sys
from collections import defaultdict
rows = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "conv_mfma" not in n: continue
        key = (n.split("(")[0].replace("void ", ""), int(r["Grid_Size"]))
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in rows.items():
    print(key, {c: round(sum(v) / len(v), 1) for c, v in cs.items()})
PY
