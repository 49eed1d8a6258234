#!/bin/bash
# round-4 first call: baseline bench at HEAD, per-op tables, TCC counters, one-rank RCCL rehearsal
tag=${1:-r04_a}
mkdir -p gpurun_out
timeout -k 10 300 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit $?
cut -c1-300 gpurun_out/${tag}_bench.json
timeout -k 10 200 python tools/gpu_explore.py time 64 > gpurun_out/${tag}_per_op.txt 2>&1 || exit $?
head -3 gpurun_out/${tag}_per_op.txt
timeout -k 10 200 python tools/gpu_explore.py time 8 > gpurun_out/${tag}_per_op_b8.txt 2>&1 || exit $?
head -3 gpurun_out/${tag}_per_op_b8.txt
BSY_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline > gpurun_out/${tag}_nccl1_rehearsal.json 2> gpurun_out/${tag}_nccl1.err; echo "nccl 1-rank rc=$?"
cut -c1-300 gpurun_out/${tag}_nccl1_rehearsal.json
tail -3 gpurun_out/${tag}_nccl1.err
tools/pmc_tcc.sh $tag > gpurun_out/${tag}_tcc.log 2>&1; echo "tcc rc=$?"
tail -5 gpurun_out/${tag}_tcc.log
