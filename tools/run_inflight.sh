#!/bin/bash
# same-box A/B: one forward in flight vs two (bench.py --inflight 2)
mkdir -p gpurun_out
for n in 1 2 1 2; do
  timeout -k 10 300 python bench.py --steps 200 --no-cpu-baseline --inflight $n > gpurun_out/inflight_$n.json 2> gpurun_out/inflight_$n.err || { tail -5 gpurun_out/inflight_$n.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/inflight_$n.json').read().strip().splitlines()[-1]); print('inflight $n:', d['value'], 'img/s', d['ms_per_step'], 'ms/step; frac', d['roofline']['frac'])"
done
timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline --inflight 2 --family bsyolo11 > gpurun_out/inflight_2_bsyolo.json 2>> gpurun_out/inflight_2.err && cut -c1-200 gpurun_out/inflight_2_bsyolo.json
timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline --inflight 2 --precision fp32x > gpurun_out/inflight_2_fp32x.json 2>> gpurun_out/inflight_2.err && cut -c1-200 gpurun_out/inflight_2_fp32x.json
