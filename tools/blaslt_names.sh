#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/blaslt -- python3 $R/tools/blaslt_names.py > /dev/null 2>&1; echo rc=$?
python3 - $R <<'PY'
import csv, glob, os, sys
f = sorted(glob.glob(f"{sys.argv[1]}/gpurun_out/blaslt/*/*kernel_trace.csv"), key=os.path.getmtime)[-1]
seen = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "Cijk" in n or "gemm" in n.lower():
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        k = (n, r["Grid_Size_X"], r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"], r.get("Accum_VGPR_Count"))
        seen.setdefault(k, []).append(d)
for k, v in seen.items():
    print(f"{min(v) / 1e3:7.1f} us  grid {k[1]} wg {k[2]} lds {k[3]} vgpr {k[4]} agpr {k[5]}  {k[0][:400]}")
PY
