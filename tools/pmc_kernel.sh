#!/bin/bash
# SQ counters of named kernels in one bench forward: where a kernel's wave cycles go (waiting / issuing VALU / LDS / MFMA).
# usage: [BSY_BENCH_ARGS="--precision fp32"] tools/pmc_kernel.sh <kernel name substring> [more substrings ...]   -> gpurun_out/pmc_kernel.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_kernel -- python3 $R/bench.py $BSY_BENCH_ARGS --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1; echo "rc=$?"
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
    # one line per distinct instantiation (template arguments) of the last forward's dispatches: aggregate of its launches
    groups = {}
    for d in ds[-400:]:
        groups.setdefault(d["name"], []).append(d)
    for name, g in sorted(groups.items(), key=lambda kv: -sum(x["dur"] for x in kv[1])):
        wc = sum(x.get("SQ_WAVE_CYCLES", 0) for x in g) or 1
        short = name[name.find(nm):][:110]
        print(f"{short}: {len(g)} launches, {sum(x['dur'] for x in g) / len(g) / 1e3:.1f} us avg, wg {g[0]['wg']} vgpr {g[0]['vgpr']} lds {g[0]['lds']}")
        line = "   "
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            line += f"{k[3:]} {100 * sum(x.get(k, 0) for x in g) / wc:5.1f} %  "
        busy = sum(x.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for x in g)
        print(line + f"MFMA busy / (4 x wave quad-cycles) {100 * busy / (4 * wc):5.1f} %")
PY
