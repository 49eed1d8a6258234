#!/bin/bash
# one gpurun call: GPU tests, then (unless the tests were killed) per-op timing, the bench line and a 2-rank gloo rehearsal
# usage: tools/quick_run.sh <tag> [pytest -k expression]
tag=$1; kexpr=$2
mkdir -p gpurun_out
if [ -n "$kexpr" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$kexpr" > gpurun_out/${tag}_tests.log 2>&1
else
  timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/${tag}_tests.log 2>&1
fi
rc=$?
tail -5 gpurun_out/${tag}_tests.log
if [ $rc -gt 1 ]; then echo "tests rc=$rc: stopping"; exit $rc; fi
timeout -k 10 300 python tools/gpu_explore.py time > gpurun_out/${tag}_time.log 2>&1 || exit $?
head -4 gpurun_out/${tag}_time.log
timeout -k 10 300 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit $?
cat gpurun_out/${tag}_bench.json | cut -c1-600
exit $rc
