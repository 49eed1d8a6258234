#!/bin/bash
# BASELINE configs at HEAD (one GPU): 1280 x 1280 lines, config 3's per-GPU shard, configs 4 / 5, fp32x on the m scale
tag=${1:-r04_f}
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --imgsz 1280 --no-cpu-baseline --steps 60 > gpurun_out/${tag}_bench_yolo11s_1280.json 2> gpurun_out/${tag}_cfg.err || exit $?
cut -c1-220 gpurun_out/${tag}_bench_yolo11s_1280.json
timeout -k 10 300 python bench.py --scale m --imgsz 1280 --batch 32 --no-cpu-baseline --steps 40 > gpurun_out/${tag}_bench_yolo11m_1280_bs32.json 2>> gpurun_out/${tag}_cfg.err || exit $?
cut -c1-220 gpurun_out/${tag}_bench_yolo11m_1280_bs32.json
timeout -k 10 300 python bench.py --scale m --imgsz 1280 --batch 32 --precision fp32x --no-cpu-baseline --steps 20 > gpurun_out/${tag}_bench_yolo11m_1280_bs32_fp32x.json 2>> gpurun_out/${tag}_cfg.err || exit $?
cut -c1-260 gpurun_out/${tag}_bench_yolo11m_1280_bs32_fp32x.json
(timeout -k 10 300 python tools/config4_time.py; timeout -k 10 300 python tools/sahi_time.py) > gpurun_out/${tag}_configs_4_5.txt 2>&1 || exit $?
tail -12 gpurun_out/${tag}_configs_4_5.txt
timeout -k 10 300 python bench.py --family bsyolo11 --precision fp32x --no-cpu-baseline --steps 50 > gpurun_out/${tag}_bench_bsyolo11s_fp32x.json 2>> gpurun_out/${tag}_cfg.err || exit $?
cut -c1-260 gpurun_out/${tag}_bench_bsyolo11s_fp32x.json
