#!/bin/bash
# one gpurun call: the profiles of the bench only (no tests): rocprofv3 kernel summaries, PMC traffic + MFMA utilisation, bench lines,
# per-op tables.  usage: tools/profile_run.sh <tag>
tag=$1
mkdir -p gpurun_out
tools/prof_bench.sh $tag 20 || exit $?
tools/pmc_bench_traffic.sh || exit $?
tools/pmc_bench_mfma.sh || exit $?
timeout -k 10 300 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit $?
cut -c1-400 gpurun_out/${tag}_bench.json
timeout -k 10 200 python tools/gpu_explore.py time 64 > gpurun_out/${tag}_per_op.txt 2>&1 || exit $?
timeout -k 10 200 python tools/gpu_explore.py time 8 > gpurun_out/${tag}_per_op_b8.txt 2>&1 || exit $?
head -3 gpurun_out/${tag}_per_op_b8.txt
