#!/usr/bin/env python3
"""Ablations of one HBM-bound 1x1 layer (model.4.cv2 of YOLO11s: 192 -> 256 @ 80 x 80, B = 64) under BSY_CONV_DBG (set in the environment:
1 = no DMA, 4 = no epilogue) for a list of configurations.  usage: BSY_CONV_DBG=<n> python tools/abl_1x1.py [cin cout H]"""
import os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bs_yolo_amd import ops as O
dev = "cuda:0"
cin, cout, H = (int(a) for a in (sys.argv[1:4] or (192, 256, 80)))
x = (torch.randn(64, H, H, cin, device=dev) * 0.5).half()
w = torch.randn(cout, cin, 1, 1) * (2.0 / cin) ** 0.5
wp, bp = O.pack_conv_weight(w, torch.zeros(cout), dev)
mb = 64 * H * H * (cin + cout) * 2 / 1e6
print("BSY_CONV_DBG =", os.environ.get("BSY_CONV_DBG", "0"), f"{cin}->{cout} @{H}: {mb:.0f} MB")
for cfg in (0x41, 0x82, 0xe1, 0xe2):
    os.environ["BSY_CONV_CFG"] = str(cfg)
    try:
        out = O.conv2d_nhwc(x, wp, bp, cout, 1, 1, True)
    except Exception:
        continue
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        O.conv2d_nhwc(x, wp, bp, cout, 1, 1, True, out=out)
        e0.record()
        for _ in range(10):
            O.conv2d_nhwc(x, wp, bp, cout, 1, 1, True, out=out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print(f"   cfg 0x{cfg:02x}  {best * 1e3:8.1f} us  {mb / best / 1e3:6.2f} TB/s (of the full layer's bytes)", flush=True)
