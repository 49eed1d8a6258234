#!/bin/bash
# A/B of an environment switch on ONE box: tools/ab_time.sh <tag> "<ENV=..>" -> gpurun_out/<tag>_{a,b}.log (a = default, b = with env)
tag=$1; envs=$2
timeout -k 10 200 python tools/gpu_explore.py time > gpurun_out/${tag}_a.log 2>&1 || exit $?
env $envs timeout -k 10 200 python tools/gpu_explore.py time > gpurun_out/${tag}_b.log 2>&1 || exit $?
timeout -k 10 200 python tools/gpu_explore.py time > gpurun_out/${tag}_a2.log 2>&1 || exit $?
head -3 gpurun_out/${tag}_a.log gpurun_out/${tag}_b.log gpurun_out/${tag}_a2.log | grep -v amdgpu
