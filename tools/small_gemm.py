#!/usr/bin/env python3
"""The 20 x 20 1x1 layers of YOLO11s (B = 64: M = 25 600 pixels): every configuration, best-of-3 bursts of 20 back-to-back launches.
usage: python tools/small_gemm.py"""
import os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bs_yolo_amd import ops as O
dev = "cuda:0"
LAYERS = (("model.8.cv2", 20, 768, 512), ("model.9.cv2", 20, 1024, 512), ("model.8.cv1", 20, 512, 512), ("model.8.m.0.cv1", 20, 256, 256))
for name, H, cin, cout in LAYERS:
    x = (torch.randn(64, H, H, cin, device=dev) * 0.5).half()
    w = torch.randn(cout, cin, 1, 1) * (2.0 / cin) ** 0.5
    wp, bp = O.pack_conv_weight(w, torch.zeros(cout), dev)
    res = {}
    for tile in (1, 2, 3, 4, 5, 6, 7):
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
                for _ in range(20):
                    O.conv2d_nhwc(x, wp, bp, cout, 1, 1, True, out=out)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20)
            res[cfg] = best
    gf = 2.0 * 64 * H * H * cin * cout / 1e9
    print(name, f"{gf:.1f} GFLOP")
    for cfg, t in sorted(res.items(), key=lambda kv: kv[1])[:6]:
        print(f"   cfg 0x{cfg:02x}  {t * 1e3:8.1f} us  {gf / t:7.1f} TF/s", flush=True)
