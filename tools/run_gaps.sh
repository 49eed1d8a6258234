set -e
tools/prof_bench.sh r04i 20
CSV=$(ls -t gpurun_out/prof_r04i/*/*kernel_trace.csv | head -1)
NCONV=$(python3 -c "import json; print(json.loads(open('gpurun_out/prof_r04i/bench_under_rocprof.json').read().strip().splitlines()[-1])['roofline']['launches_per_step'])")
python3 tools/trace_gaps.py $CSV 20 $NCONV > gpurun_out/r04i_trace_gaps.txt
cat gpurun_out/r04i_trace_gaps.txt
timeout -k 10 200 python tools/gpu_explore.py time 64 > gpurun_out/r04i_per_op.txt 2>&1
head -3 gpurun_out/r04i_per_op.txt
