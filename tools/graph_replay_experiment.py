#!/usr/bin/env python3
"""Experiment: replay the forward as a HIP graph (torch.cuda.CUDAGraph capture of bsy_plan_run) vs eager launches."""
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
eng = YoloEngine(cfg, sd)
x = torch.rand(64, 3, 640, 640, device="cuda:0").half()
for _ in range(3):
    y, _ = eng(x, want_raw=False)
torch.cuda.synchronize()
def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print(f"eager : {bench(lambda: eng(x, want_raw=False)):.3f} ms", flush=True)
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng(x, want_raw=False)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g, capture_error_mode="relaxed"):
        yg, _ = eng(x, want_raw=False)
    torch.cuda.synchronize()
    print(f"graph : {bench(g.replay):.3f} ms", flush=True)
    print("equal:", torch.equal(yg, y))
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
print(f"eager : {bench(lambda: eng(x, want_raw=False)):.3f} ms", flush=True)
