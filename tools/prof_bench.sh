#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py + per-family summary (tools/prof_summary.py); outputs under gpurun_out/prof_<tag>/.
# usage: tools/prof_bench.sh <tag> [steps=20]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}; STEPS=${2:-20}
mkdir -p $R/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps $STEPS --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_$TAG/bench_under_rocprof.json 2> $R/gpurun_out/prof_$TAG/bench.err
echo "rocprof rc=$?"
NCONV=$(python3 -c "import json,sys; print(json.loads(open('$R/gpurun_out/prof_$TAG/bench_under_rocprof.json').read().strip().splitlines()[-1])['roofline']['launches_per_step'])")
CSV=$(ls -t $R/gpurun_out/prof_$TAG/*/*kernel_trace.csv | head -1)
python3 $R/tools/prof_summary.py $CSV $STEPS $NCONV > $R/gpurun_out/prof_$TAG/kernel_summary.txt
# bench.py ends with 3 SERIAL passes (bsy_plan_profile: every op on the caller's stream, no side lanes): the kernel
# durations of those passes are the ones comparable with bench.py's roofline.avg_launch_ms
python3 $R/tools/prof_summary.py $CSV 3 $NCONV > $R/gpurun_out/prof_$TAG/kernel_summary_serial_passes.txt
cp $(ls -t $R/gpurun_out/prof_$TAG/*/*kernel_stats.csv | head -1) $R/gpurun_out/prof_$TAG/kernel_stats.csv
cat $R/gpurun_out/prof_$TAG/kernel_summary.txt
tail -1 $R/gpurun_out/prof_$TAG/kernel_summary_serial_passes.txt
python3 -c "
import json
d=json.loads(open('$R/gpurun_out/prof_$TAG/bench_under_rocprof.json').read().strip().splitlines()[-1])
r=d['roofline']; print('bench (under rocprof): ms/step', d['ms_per_step'], 'conv avg launch us', r['avg_launch_ms']*1e3, 'family ms/step', r['family_ms_per_step'], 'achieved', r['achieved'], 'frac', r['frac'])"
