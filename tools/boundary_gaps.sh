#!/bin/bash
# What the dependent-launch boundaries of one forward cost: rocprofv3 --kernel-trace (no counters) of bench.py with every op on ONE
# stream (BSY_LANES=0, --serial-nms), gaps between consecutive kernels of the timed steps.  -> gpurun_out/<tag>_boundary_gaps.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
export BSY_LANES=0
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gaps -- python3 $R/bench.py --steps 20 --warmup 5 --serial-nms --no-cpu-baseline > $R/gpurun_out/gaps.json 2> $R/gpurun_out/gaps.err; echo "rc=$?"
python3 - $R <<'PY' | tee $R/gpurun_out/${TAG}_boundary_gaps.txt
import csv, glob, json, os, sys
R = sys.argv[1]
f = sorted(glob.glob(f"{R}/gpurun_out/gaps/*/*kernel_trace.csv"), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
stem = [i for i, r in enumerate(rows) if "stem_fused" in r["Kernel_Name"]]
# the 20 timed steps = the 20 forwards before the last 1 (parity) + 3 (profile passes); each starts at a stem kernel
starts = stem[-(20 + 4 + 1):-(4 + 1)] if len(stem) >= 26 else stem[5:25]
tot_d = tot_g = n = 0
gaps = []
for s0, s1 in zip(starts[:-1], starts[1:]):
    seq = rows[s0:s1]
    for a, b in zip(seq[:-1], seq[1:]):
        tot_d += int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
        g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
        gaps.append(g); tot_g += g; n += 1
steps = len(starts) - 1
gaps.sort()
b = json.loads(open(f"{R}/gpurun_out/gaps.json").read().strip().splitlines()[-1])
print(f"# serial schedule (BSY_LANES=0, NMS on the forward's stream), {steps} timed steps under rocprofv3 --kernel-trace: {b['ms_per_step']:.3f} ms per step by the bench's clock")
print(f"kernels per step {n / steps:.1f}; kernel time {tot_d / steps / 1e6:.3f} ms per step; gaps between consecutive kernels {tot_g / steps / 1e6:.3f} ms per step")
print(f"gap per boundary: median {gaps[len(gaps) // 2] / 1e3:.2f} us, p10 {gaps[len(gaps) // 10] / 1e3:.2f} us, p90 {gaps[9 * len(gaps) // 10] / 1e3:.2f} us, mean {tot_g / n / 1e3:.2f} us")
PY
