#!/bin/bash
# one gpurun call: the whole GPU suite, a 2-rank gloo rehearsal of bench.py's self-launch, then the profiles of the bench
tag=$1; nfam=${2:-73}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/${tag}_tests.log 2>&1; rc=$?
tail -4 gpurun_out/${tag}_tests.log
if [ $rc -gt 1 ]; then echo "tests rc=$rc: stopping"; exit $rc; fi
BSY_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --batch 16 --scaling strong --no-cpu-baseline > gpurun_out/${tag}_gloo2.json 2> gpurun_out/${tag}_gloo2.err || { echo "gloo rehearsal failed"; tail -5 gpurun_out/${tag}_gloo2.err; }
cut -c1-300 gpurun_out/${tag}_gloo2.json
tools/prof_bench.sh $tag 20 || exit $?
tools/pmc_bench_traffic.sh $nfam || exit $?
tools/pmc_bench_mfma.sh $nfam || exit $?
timeout -k 10 300 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit $?
cut -c1-400 gpurun_out/${tag}_bench.json
exit $rc
