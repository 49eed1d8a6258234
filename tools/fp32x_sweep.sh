#!/bin/bash
# fp32x tile / K-step sweep: per-op table of the YOLO11s 64 x 640^2 forward for every (tile, BK) override
mkdir -p gpurun_out
for t in 0 1 2; do for bk in 16 32; do
  [ $t = 0 ] && [ $bk = 16 ] && continue
  BSY_CONV32X_TILE=$t BSY_CONV32X_BK=$bk timeout -k 10 200 python tools/gpu_explore.py time 64 yolo11 fp32x > gpurun_out/fp32x_t${t}_bk${bk}.txt 2>&1 || exit $?
  head -3 gpurun_out/fp32x_t${t}_bk${bk}.txt | tail -2
done; done
timeout -k 10 200 python tools/gpu_explore.py time 64 yolo11 fp32x > gpurun_out/fp32x_auto.txt 2>&1 || exit $?
head -3 gpurun_out/fp32x_auto.txt | tail -2
