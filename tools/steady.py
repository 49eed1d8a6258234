#!/usr/bin/env python3
"""Experiment: steady-state TF/s of every conv config on a large GEMM-like problem (quantization negligible)."""
import os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bs_yolo_amd import ops as O
dev = "cuda:0"
for (cin, cout, k, hw) in ((1024, 1024, 1, 64), (256, 256, 3, 64), (512, 512, 1, 64)):
    w = torch.randn(cout, cin, k, k) * (2.0 / (cin * k * k)) ** 0.5
    wp, bp = O.pack_conv_weight(w, torch.zeros(cout), dev)
    x = (torch.randn(64, hw, hw, cin, device=dev) * 0.5).half()
    line = f"{k}x{k} {cin}->{cout} M={64 * hw * hw}:"
    for t in (2, 3, 4, 7, 8):
        for v in (1, 2, 3):
            cfg = (t << 4) | v
            os.environ["BSY_CONV_CFG"] = str(cfg)
            try:
                out = O.conv2d_nhwc(x, wp, bp, cout, k, 1, True)
            except Exception:
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                O.conv2d_nhwc(x, wp, bp, cout, k, 1, True, out=out)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            fl = 2 * 64 * hw * hw * cout * cin * k * k
            line += f" {cfg:#x}:{fl / ms / 1e9:.0f}"
    print(line, flush=True)
