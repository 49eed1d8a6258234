#!/bin/bash
# one gpurun call: the whole GPU suite, a 2-rank gloo rehearsal of bench.py's self-launch, a ONE-rank RCCL rehearsal, then the profiles of
# the bench, the per-op tables, the fp32x / fp32-mode bench lines and the BS-YOLO11s bench line (the files copied to profiles/<tag>_*)
# usage: tools/full_run.sh <tag>
tag=$1; nfam=${2:-0}  # 0: the PMC scripts read the family's launch count from their own bench run
mkdir -p gpurun_out
# the library must be the one the sources build (objects and their signatures travel with the snapshot: a no-op when up to date, loud when a source does not compile)
python -m bs_yolo_amd.build > gpurun_out/${tag}_build.log 2>&1 || { echo "BUILD FAILED"; tail -20 gpurun_out/${tag}_build.log; exit 1; }
tail -1 gpurun_out/${tag}_build.log | cut -c1-120
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/${tag}_tests.log 2>&1; rc=$?
tail -4 gpurun_out/${tag}_tests.log
if [ $rc -gt 1 ]; then echo "tests rc=$rc: stopping"; exit $rc; fi
BSY_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --batch 16 --scaling strong --no-cpu-baseline > gpurun_out/${tag}_gloo2_rehearsal.json 2> gpurun_out/${tag}_gloo2.err || { echo "gloo rehearsal failed"; tail -5 gpurun_out/${tag}_gloo2.err; }
cut -c1-300 gpurun_out/${tag}_gloo2_rehearsal.json
BSY_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline > gpurun_out/${tag}_nccl1_rehearsal.json 2> gpurun_out/${tag}_nccl1.err || { echo "RCCL one-rank rehearsal failed"; tail -5 gpurun_out/${tag}_nccl1.err; }
cut -c1-300 gpurun_out/${tag}_nccl1_rehearsal.json
tools/prof_bench.sh $tag 20 || exit $?
tools/pmc_bench_traffic.sh $nfam || exit $?
tools/pmc_bench_mfma.sh $nfam || exit $?
tools/pmc_tcc.sh $tag > gpurun_out/${tag}_tcc.log 2>&1 || exit $?
timeout -k 10 300 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit $?
cut -c1-400 gpurun_out/${tag}_bench.json
timeout -k 10 200 python tools/gpu_explore.py time 64 > gpurun_out/${tag}_per_op.txt 2>&1 || exit $?
timeout -k 10 200 python tools/gpu_explore.py time 8 > gpurun_out/${tag}_per_op_b8.txt 2>&1 || exit $?
timeout -k 10 200 python tools/gpu_explore.py time 64 yolo11 fp32x > gpurun_out/${tag}_per_op_fp32x.txt 2>&1 || exit $?
timeout -k 10 300 python bench.py --precision fp32x --steps 100 > gpurun_out/${tag}_bench_fp32x.json 2>> gpurun_out/${tag}_bench.err || exit $?
cut -c1-300 gpurun_out/${tag}_bench_fp32x.json
timeout -k 10 300 python bench.py --precision fp32 --steps 50 --no-cpu-baseline > gpurun_out/${tag}_bench_fp32.json 2>> gpurun_out/${tag}_bench.err || exit $?
cut -c1-300 gpurun_out/${tag}_bench_fp32.json
timeout -k 10 300 python bench.py --family bsyolo11 --no-cpu-baseline > gpurun_out/${tag}_bench_bsyolo11s_640.json 2>> gpurun_out/${tag}_bench.err || exit $?
cut -c1-300 gpurun_out/${tag}_bench_bsyolo11s_640.json
exit $rc
