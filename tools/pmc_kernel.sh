#!/bin/bash
# SQ counters of named kernels in one bench forward: where a kernel's wave cycles go (waiting / issuing VALU / LDS / MFMA).
# usage: tools/pmc_kernel.sh <kernel name substring> [more substrings ...]   -> gpurun_out/pmc_kernel.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_kernel -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1; echo "rc=$?"
python3 - $R "$@" <<'PY' | tee $R/gpurun_out/pmc_kernel.txt
import csv, glob, os, sys
R, names = sys.argv[1], sys.argv[2:]
f = sorted(glob.glob(f"{R}/gpurun_out/pmc_kernel/*/*counter_collection.csv"), key=os.path.getmtime)[-1]
disp = {}
for r in csv.DictReader(open(f)):
    d = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "grid": r.get("Grid_Size"), "wg": r.get("Workgroup_Size"), "vgpr": r.get("VGPR_Count"), "lds": r.get("LDS_Block_Size")})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for nm in names:
    ds = [d for d in disp.values() if nm in d["name"]]
    if not ds:
        print(nm, "not found"); continue
    d = ds[-1]
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{nm}: {d['dur'] / 1e3:.1f} us, grid {d['grid']} wg {d['wg']} vgpr {d['vgpr']} lds {d['lds']}")
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
        print(f"   {k:22s} {100 * d.get(k, 0) / wc:6.1f} % of wave cycles")
    print(f"   VALU insts {d.get('SQ_INSTS_VALU', 0):.3e}  LDS insts {d.get('SQ_INSTS_LDS', 0):.3e}  MFMA busy cycles {d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.3e}  wave cycles (quad) {wc:.3e}")
PY
