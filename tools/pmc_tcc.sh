#!/bin/bash
# L2 (TCC) hit rate and memory-side read requests of every conv launch of one bench forward: where the staged operand bytes
# of a conv tile are served from (the XCD's L2, or beyond it: Infinity Cache / HBM).  Counters in their own passes
# (--kernel-trace only beside them).  usage: tools/pmc_tcc.sh [tag]   -> gpurun_out/<tag>_tcc.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/${TAG}_counters_list.txt 2>&1
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  d=$R/gpurun_out/pmc_tcc_$(echo $set | cut -d' ' -f1)
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $d.json 2> $d.err; echo "$set rc=$?"
done
python3 - $R $TAG <<'PY' | tee $R/gpurun_out/${TAG}_tcc.txt
import csv, glob, json, os, sys
R, TAG = sys.argv[1], sys.argv[2]
CONV = ("conv_mfma_kernel", "conv1x1_persist_kernel", "conv1x1_wres_kernel", "conv3x3_patch_kernel", "stem_fused_kernel", "bneck_fused_kernel",
        "bneck_fused_wide_kernel", "c3k2_fused_kernel", "dwpw_fused_kernel", "conv_first_mfma_kernel", "chain1x1_kernel")
disp = {}
n_fam = None
for first in ("TCC_HIT_sum", "TCC_EA0_RDREQ_sum"):
    fs = sorted(glob.glob(f"{R}/gpurun_out/pmc_tcc_{first}/*/*counter_collection.csv"), key=os.path.getmtime)
    if not fs:
        print(first, "pass: no counter file"); continue
    try:
        n_fam = int(json.loads(open(f"{R}/gpurun_out/pmc_tcc_{first}.json").read().strip().splitlines()[-1])["roofline"]["launches_per_step"])
    except Exception as e:
        print("no bench line for", first, e)
    rows = list(csv.DictReader(open(fs[-1])))
    per = {}
    for r in rows:
        d = per.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "t0": int(r["Start_Timestamp"]), "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    seq = sorted(per.values(), key=lambda d: d["t0"])
    conv = [d for d in seq if any(k in d["name"] for k in CONV)]
    last = conv[-(n_fam or 68):]          # the last serial profile pass of bench.py = one forward
    disp[first] = last
if "TCC_HIT_sum" in disp:
    a = disp["TCC_HIT_sum"]
    b = disp.get("TCC_EA0_RDREQ_sum", [None] * len(a))
    print(f"# {TAG}: L2 (TCC) counters of the {len(a)} dense-conv launches of one forward, YOLO11s 64 x 640 x 640 fp16, in launch order")
    print(f"# {'kernel instantiation':70s} {'us':>7s} {'hit %':>6s} {'hits M':>8s} {'miss M':>8s} {'EA rdreq M':>10s} {'req M':>8s}")
    H = M = 0
    for i, d in enumerate(a):
        nm = d["name"]
        k = next(k for k in CONV if k in nm)
        short = nm[nm.find(k):][:70]
        h, m = d.get("TCC_HIT_sum", 0), d.get("TCC_MISS_sum", 0)
        H += h; M += m
        e = b[i] if i < len(b) and b[i] is not None else {}
        print(f"  {short:70s} {d['dur'] / 1e3:7.1f} {100 * h / max(h + m, 1):6.1f} {h / 1e6:8.2f} {m / 1e6:8.2f} {e.get('TCC_EA0_RDREQ_sum', 0) / 1e6:10.2f} {e.get('TCC_REQ_sum', 0) / 1e6:8.2f}")
    print(f"# family: hit rate {100 * H / max(H + M, 1):.1f} %  ({H / 1e6:.1f} M hits, {M / 1e6:.1f} M misses; one request = one 128-B line)")
PY
