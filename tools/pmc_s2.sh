#!/bin/bash
# FETCH_SIZE / WRITE_SIZE and duration of the stride-2 convs under K-order experiments (BSY_CONV_DBG 0 / 64)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for d in 0 64; do
  for c in FETCH_SIZE; do
    BSY_CONV_DBG=$d timeout -k 5 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_s2_${d}_$c -- python3 $R/tools/one_s2.py > /dev/null 2>&1; echo "dbg $d $c rc=$?"
  done
done
python3 - $R <<'PY'
import csv, glob, sys, collections
R = sys.argv[1]
for d in (0, 64):
    f = glob.glob(f"{R}/gpurun_out/pmc_s2_{d}_FETCH_SIZE/*/*counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if "conv_mfma_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # 4 launches per (shape, cfg): report the last of each group
    for i in range(0, len(rows), 4):
        g = rows[i:i + 4]
        r = g[-1]
        print(f"dbg {d} launch-group {i // 4}: grid {r['Grid_Size']} fetch*2 {2 * float(r['Counter_Value']) * 1024 / 1e6:8.1f} MB  dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us")
PY
