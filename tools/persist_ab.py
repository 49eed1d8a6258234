#!/usr/bin/env python3
"""A/B the persistent 1x1 conv configs (tile ids 8/9) against the one-tile-per-workgroup configs, with a numerics check."""
import os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bs_yolo_amd import ops as O

SHAPES = [(64, 160, 160, 64, 64), (64, 160, 160, 96, 128), (64, 80, 80, 128, 128), (64, 80, 80, 192, 256),
          (64, 80, 80, 256, 128), (64, 40, 40, 256, 256), (64, 40, 40, 384, 256), (64, 40, 40, 768, 256),
          (64, 80, 80, 128, 64), (64, 80, 80, 64, 80), (3, 37, 41, 96, 72)]
CFGS = [None] + [(t << 4) | v for t in (2, 3, 6, 8, 9) for v in (1, 2, 3)]
dev = "cuda:0"
for (B, H, W, cin, cout) in SHAPES:
    x = (torch.randn(B, H, W, cin, device=dev) * 0.5).half()
    w = torch.randn(cout, cin, 1, 1) * (2.0 / cin) ** 0.5
    bias = torch.randn(cout) * 0.1
    wp, bp = O.pack_conv_weight(w, bias, dev)
    ref = torch.nn.functional.silu(x.float().reshape(-1, cin) @ w.reshape(cout, cin).half().float().t().to(dev) + bias.to(dev))
    line = f"{cin:4d}->{cout:4d} @{H}x{W}:"
    for cfg in CFGS:
        if cfg is None:
            os.environ.pop("BSY_CONV_CFG", None)
        else:
            os.environ["BSY_CONV_CFG"] = str(cfg)
        try:
            out = O.conv2d_nhwc(x, wp, bp, cout, 1, 1, True)
        except Exception as e:
            line += f" {cfg}:n/a"
            continue
        err = (out.float().reshape(-1, cout) - ref).abs().max().item()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            O.conv2d_nhwc(x, wp, bp, cout, 1, 1, True, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        line += f" {'H' if cfg is None else cfg}:{us:.0f}" + ("" if err < 2e-2 else f"(ERR {err:.3g})")
    print(line, flush=True)
