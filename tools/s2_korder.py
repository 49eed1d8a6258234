#!/usr/bin/env python3
"""Stride-2 3x3 layers of YOLO11s (B = 64, 640 x 640): every implicit-GEMM configuration with the tap-major and the chunk-major K walk
(variant + 4), best-of-3 bursts of 10 launches each.  usage: python tools/s2_korder.py"""
import os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bs_yolo_amd import ops as O
dev = "cuda:0"
LAYERS = (("model.3", 160, 128, 128), ("model.5", 80, 256, 256), ("model.7", 40, 256, 512), ("model.17", 80, 128, 128), ("model.20", 40, 256, 256))
for name, H, cin, cout in LAYERS:
    x = (torch.randn(64, H, H, cin, device=dev) * 0.5).half()
    w = torch.randn(cout, cin, 3, 3) * (2.0 / (cin * 9)) ** 0.5
    wp, bp = O.pack_conv_weight(w, torch.zeros(cout), dev)
    res = {}
    for tile in (1, 2, 3, 4, 5, 7):
        for var in (1, 2, 3, 5, 6, 7):
            cfg = tile << 4 | var
            os.environ["BSY_CONV_CFG"] = str(cfg)
            try:
                out = O.conv2d_nhwc(x, wp, bp, cout, 3, 2, True)
            except Exception:
                continue
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                O.conv2d_nhwc(x, wp, bp, cout, 3, 2, True, out=out)
                e0.record()
                for _ in range(10):
                    O.conv2d_nhwc(x, wp, bp, cout, 3, 2, True, out=out)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            res[cfg] = best
    gf = 2.0 * 64 * (H // 2) ** 2 * 9 * cin * cout / 1e9
    print(name, f"{gf:.1f} GFLOP")
    for cfg, t in sorted(res.items(), key=lambda kv: kv[1])[:8]:
        print(f"   cfg 0x{cfg:02x}  {t * 1e3:8.1f} us  {gf / t:7.1f} TF/s")
    for cfg in sorted(res):
        if cfg & 4 and (cfg & ~4) in res:
            print(f"   0x{cfg & ~4:02x} {res[cfg & ~4] * 1e3:7.1f} us -> 0x{cfg:02x} {res[cfg] * 1e3:7.1f} us")
