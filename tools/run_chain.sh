#!/bin/bash
# chain1x1 kernel: its parity tests, then the per-op table with and without it (same box)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -k "chain" > gpurun_out/chain_tests.log 2>&1; rc=$?
tail -15 gpurun_out/chain_tests.log
echo "chain tests rc=$rc"
timeout -k 10 200 python tools/gpu_explore.py time 64 > gpurun_out/chain_per_op.txt 2>&1 || exit $?
BSY_FUSE_CHAIN=0 timeout -k 10 200 python tools/gpu_explore.py time 64 > gpurun_out/nochain_per_op.txt 2>&1 || exit $?
head -3 gpurun_out/chain_per_op.txt; head -3 gpurun_out/nochain_per_op.txt
grep "kind 21" gpurun_out/chain_per_op.txt
