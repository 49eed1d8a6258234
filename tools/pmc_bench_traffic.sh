#!/bin/bash
# HBM traffic of one bench step per kernel family: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
# (MI355X_MICROARCH.md "HBM": FETCH_SIZE reports exactly half of a wide coalesced read stream on gfx950 -> doubled;
# WRITE_SIZE is exact; both in KiB).  Writes gpurun_out/traffic.json; copy it to profiles/ to have bench.py report it.
# usage: tools/pmc_bench_traffic.sh [family launches per forward]   ("conv_mfma_kernel" below = the whole dense-conv family incl. the
# fused conv kernels).  The launch count that windows "the last forward" is read from the bench line of the profiled run itself
# (roofline.launches_per_step); the optional argument only overrides it.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_bench_$c -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_bench_$c.json 2> /dev/null; echo "$c rc=$?"
done
python3 - $R ${1:-0} <<'PY'
import csv, glob, json, os, sys
R = sys.argv[1]
NCONV = int(sys.argv[2])  # 0: take roofline.launches_per_step of the profiled bench run
if NCONV <= 0:
    NCONV = int(json.loads(open(f"{R}/gpurun_out/pmc_bench_FETCH_SIZE.json").read().strip().splitlines()[-1])["roofline"]["launches_per_step"])
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = sorted(glob.glob(f"{R}/gpurun_out/pmc_bench_{c}/*/*counter_collection.csv"), key=os.path.getmtime)[-1]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    CONV = ("conv_mfma_kernel", "conv1x1_persist_kernel", "conv1x1_wres_kernel", "conv3x3_patch_kernel", "stem_fused_kernel", "bneck_fused_kernel", "bneck_fused_wide_kernel", "c3k2_fused_kernel", "dwpw_fused_kernel", "conv_first_mfma_kernel", "chain1x1_kernel")  # kernels behind the plan's OP_CONV ops
    conv = [r for r in rows if any(k in r["Kernel_Name"] for k in CONV)]
    last = conv[-NCONV:]  # the last forward (bench's final profile pass)
    t0 = int(last[0]["Start_Timestamp"])
    fam = {}
    for r in rows:
        if int(r["Start_Timestamp"]) < t0: continue
        n = r["Kernel_Name"]
        key = "conv_mfma_kernel" if any(k in n for k in CONV) else n.split("(")[0].split("<")[0][-40:]
        fam[key] = fam.get(key, 0.0) + float(r["Counter_Value"]) * 1024.0
    out[c] = fam
conv_bytes = 2.0 * out["FETCH_SIZE"]["conv_mfma_kernel"] + out["WRITE_SIZE"]["conv_mfma_kernel"]
res = {"conv_mfma_hbm_bytes_per_step": conv_bytes, "conv_launches_per_step": NCONV,
       "conv_mfma_hbm_bytes_per_launch_avg": conv_bytes / NCONV,
       "fetch_bytes_raw": out["FETCH_SIZE"]["conv_mfma_kernel"], "write_bytes": out["WRITE_SIZE"]["conv_mfma_kernel"],
       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on bench.py, last forward; bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 correction)",
       "all_families_fetch_x2_plus_write": {k: 2 * out["FETCH_SIZE"].get(k, 0) + out["WRITE_SIZE"].get(k, 0) for k in set(out["FETCH_SIZE"]) | set(out["WRITE_SIZE"])}}
json.dump(res, open(f"{R}/gpurun_out/traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "all_families_fetch_x2_plus_write"}))
print({k: round(v / 1e6, 1) for k, v in res["all_families_fetch_x2_plus_write"].items()})
PY
