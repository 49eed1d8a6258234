#!/usr/bin/env python3
"""Experiment: B=64 as one plan vs two B=32 (or four B=16) plans on concurrent streams (tail overlap)."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bs_yolo_amd.engine import YoloEngine
from bs_yolo_amd.graphs import stock_cfg
from bs_yolo_amd.plan import Plan
from bs_yolo_amd.weights import synth_state_dict

cfg = stock_cfg("yolo11", "s")
sd = synth_state_dict(Plan(cfg, 1, 64, 64), 0)
B = 64
x = torch.rand(B, 3, 640, 640, device="cuda:0").half()


def run(nsplit, iters=20):
    engs = [YoloEngine(cfg, sd) for _ in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    xs = [c.contiguous() for c in x.chunk(nsplit)]
    def step():
        cur = torch.cuda.current_stream()
        for e, s, xc in zip(engs, streams, xs):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                e(xc, want_raw=False)
        for s in streams:
            cur.wait_stream(s)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"split {nsplit}: {dt * 1e3:.3f} ms/64 images  {B / dt:.0f} img/s", flush=True)
    for e in engs:
        e.close()


for n in (1, 2, 4, 1, 2):
    run(n)
