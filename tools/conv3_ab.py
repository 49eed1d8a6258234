#!/usr/bin/env python3
"""Time 3x3 conv shapes of yolo11s (B=64) under the current BSY_CONV_DBG / BSY_CONV_CFG environment."""
import os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bs_yolo_amd import ops as O

SHAPES = [(64, 320, 320, 32, 64, 2), (64, 160, 160, 128, 128, 2), (64, 80, 80, 256, 256, 2), (64, 40, 40, 256, 512, 2),
          (64, 80, 80, 64, 64, 1), (64, 80, 80, 128, 64, 1), (64, 40, 40, 64, 64, 1), (64, 40, 40, 128, 128, 1), (64, 20, 20, 128, 128, 1),
          (64, 160, 160, 32, 32, 1), (64, 80, 80, 32, 64, 1), (64, 80, 80, 64, 32, 1), (64, 40, 40, 256, 64, 1), (64, 40, 40, 64, 128, 1)]
cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [-1]
if len(sys.argv) > 2:
    SHAPES = [s for s in SHAPES if s[5] == int(sys.argv[2])]
dev = "cuda:0"
for (B, H, W, cin, cout, s) in SHAPES:
    x = (torch.randn(B, H, W, cin, device=dev) * 0.5).half()
    w = torch.randn(cout, cin, 3, 3) * (2.0 / (cin * 9)) ** 0.5
    wp, bp = O.pack_conv_weight(w, torch.zeros(cout), dev)
    line = f"3x3s{s} {cin:4d}->{cout:4d} @{H // s}x{W // s}:"
    for cfg in cfgs:
        if cfg < 0:
            os.environ.pop("BSY_CONV_CFG", None)
        else:
            os.environ["BSY_CONV_CFG"] = str(cfg)
        try:
            out = O.conv2d_nhwc(x, wp, bp, cout, 3, s, True)
        except Exception:
            line += f" {cfg}:n/a"
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            O.conv2d_nhwc(x, wp, bp, cout, 3, s, True, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        fl = 2 * B * out.shape[1] * out.shape[2] * cout * cin * 9
        line += f" {cfg}:{ms * 1e3:.0f}us/{fl / ms / 1e9:.0f}TF"
    print(line, flush=True)
