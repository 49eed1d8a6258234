#!/bin/bash
# MFMA utilisation of one bench step per kernel family (north_star: "MFMA utilisation against gfx950 peaks"):
#   rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES on bench.py
# (counters only with --kernel-trace, their own run).  SQ_VALU_MFMA_BUSY_CYCLES = sum over the SIMDs of the cycles their
# matrix pipe is busy (32 per v_mfma_f32_32x32x16_f16, MI355X_MICROARCH.md); GRBM_GUI_ACTIVE = sum over the 8 XCDs of the
# cycles the dispatch is active  ->  utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs).
# usage: tools/pmc_bench_mfma.sh [conv launches per forward]   -> gpurun_out/mfma_util.txt   (the launch count is read from the bench
# line of the profiled run, roofline.launches_per_step, unless given)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_bench_mfma -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_bench_mfma.json 2> /dev/null; echo "rc=$?"
python3 - $R ${1:-0} <<'PY' | tee $R/gpurun_out/mfma_util.txt
import csv, glob, json, os, sys
R, NCONV = sys.argv[1], int(sys.argv[2])
if NCONV <= 0:
    NCONV = int(json.loads(open(f"{R}/gpurun_out/pmc_bench_mfma.json").read().strip().splitlines()[-1])["roofline"]["launches_per_step"])
f = sorted(glob.glob(f"{R}/gpurun_out/pmc_bench_mfma/*/*counter_collection.csv"), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
CONV = ("conv_mfma_kernel", "conv1x1_persist_kernel", "conv1x1_wres_kernel", "conv3x3_patch_kernel", "stem_fused_kernel", "bneck_fused_kernel", "bneck_fused_wide_kernel", "c3k2_fused_kernel", "dwpw_fused_kernel", "conv_first_mfma_kernel", "chain1x1_kernel")
disp = {}
for r in rows:
    d = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "t": int(r["Start_Timestamp"]), "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ds = sorted(disp.values(), key=lambda d: d["t"])
conv = [d for d in ds if any(k in d["name"] for k in CONV)]
t0 = conv[-NCONV]["t"]  # the last forward (bench's final serial profile pass)
fam = {}
for d in ds:
    if d["t"] < t0:
        continue
    n = d["name"]
    key = "conv family" if any(k in n for k in CONV) else n.split("(")[0].split("<")[0][-40:]
    a = fam.setdefault(key, {"n": 0, "busy": 0.0, "gui": 0.0, "mfma": 0.0, "ns": 0})
    a["n"] += 1; a["busy"] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0); a["gui"] += d.get("GRBM_GUI_ACTIVE", 0)
    a["mfma"] += d.get("SQ_INSTS_MFMA", 0); a["ns"] += d["dur"]
print("# last forward of bench.py under rocprofv3 --pmc (YOLO11s 640x640 fp16, batch 64)")
print(f"{'kernel family':42s} {'launches':>8s} {'ms':>8s} {'MFMA insts':>12s} {'MFMA busy / SIMD cycles':>24s} {'eff. clock GHz':>15s}")
for k, a in sorted(fam.items(), key=lambda kv: -kv[1]["ns"]):
    if a["gui"] <= 0:
        continue
    util = a["busy"] / (a["gui"] / 8 * 1024)
    clk = a["gui"] / 8 / a["ns"]
    print(f"{k:42s} {a['n']:8d} {a['ns'] / 1e6:8.3f} {a['mfma']:12.0f} {100 * util:23.1f}% {clk:15.2f}")
PY
