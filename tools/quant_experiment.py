#!/usr/bin/env python3
"""Experiment: does kernel time move in whole 'rounds' of workgroups?  3x3 s2 256->256, cfg forced, M varied."""
import os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bs_yolo_amd import ops as O
dev = "cuda:0"
cin = cout = 256
w = torch.randn(cout, cin, 3, 3) * (2.0 / (cin * 9)) ** 0.5
wp, bp = O.pack_conv_weight(w, torch.zeros(cout), dev)
for cfg in (0x23, 0x21, 0x43):
    os.environ["BSY_CONV_CFG"] = str(cfg)
    for (oh, ow) in ((32, 24), (32, 32), (32, 40), (32, 44), (32, 48), (40, 40), (32, 52), (32, 56), (32, 60), (32, 64), (32, 72)):
        x = (torch.randn(64, 2 * oh, 2 * ow, cin, device=dev) * 0.5).half()
        out = O.conv2d_nhwc(x, wp, bp, cout, 3, 2, True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            O.conv2d_nhwc(x, wp, bp, cout, 3, 2, True, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        M = 64 * oh * ow
        fl = 2 * M * cout * cin * 9
        print(f"cfg {cfg:#x} M={M:7d} tiles128={M // 128 * 2:5d}: {ms * 1e3:6.1f} us {fl / ms / 1e9:6.0f} TF/s", flush=True)
