#!/usr/bin/env python3
"""The 1x1 layers of YOLO11s (B = 64, 640 x 640) whose Cout is a multiple of 256: every configuration, best-of-3 bursts of 10 launches.
usage: python tools/one_1x1.py"""
import os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bs_yolo_amd import ops as O
dev = "cuda:0"
LAYERS = (("model.4.cv2", 80, 192, 256), ("model.6.cv1", 40, 256, 256), ("model.6.cv2", 40, 384, 256), ("model.13.cv2", 40, 384, 256),
          ("model.8.cv1", 20, 512, 512), ("model.8.cv2", 20, 768, 512))
for name, H, cin, cout in LAYERS:
    x = (torch.randn(64, H, H, cin, device=dev) * 0.5).half()
    w = torch.randn(cout, cin, 1, 1) * (2.0 / cin) ** 0.5
    wp, bp = O.pack_conv_weight(w, torch.zeros(cout), dev)
    res = {}
    for tile in (2, 4, 5, 7, 8):
        for var in (1, 2, 3):
            cfg = tile << 4 | var
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
            res[cfg] = best
    mb = 64 * H * H * (cin + cout) * 2 / 1e6
    print(name, f"{mb:.0f} MB")
    for cfg, t in sorted(res.items(), key=lambda kv: kv[1]):
        print(f"   cfg 0x{cfg:02x}  {t * 1e3:8.1f} us  {mb / t / 1e3:7.2f} TB/s")
