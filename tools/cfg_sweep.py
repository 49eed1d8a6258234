#!/usr/bin/env python3
"""Time every valid conv configuration (tile << 4 | variant) on a few layer shapes; prints the five fastest per shape."""
import os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bs_yolo_amd import ops as O

SHAPES = [(64, 80, 80, 256, 256, 3, 2), (64, 40, 40, 256, 512, 3, 2), (64, 20, 20, 512, 512, 1, 1), (64, 20, 20, 768, 512, 1, 1), (64, 20, 20, 1024, 512, 1, 1), (64, 20, 20, 256, 512, 1, 1),
          (64, 40, 40, 384, 256, 1, 1), (64, 40, 40, 768, 256, 1, 1), (64, 40, 40, 256, 256, 1, 1), (64, 80, 80, 192, 256, 1, 1),
          (64, 80, 80, 128, 128, 1, 1), (64, 160, 160, 128, 128, 3, 2), (64, 80, 80, 128, 128, 3, 2), (64, 80, 80, 512, 128, 1, 1), (64, 160, 160, 96, 128, 1, 1)]
dev = "cuda:0"
if len(sys.argv) > 1:  # cfg_sweep.py <index> : one shape only
    SHAPES = [SHAPES[int(sys.argv[1])]]
for (B, H, W, cin, cout, k, s) in SHAPES:
    x = (torch.randn(B, H, W, cin, device=dev) * 0.5).half()
    w = torch.randn(cout, cin, k, k) * (2.0 / (cin * k * k)) ** 0.5
    wp, bp = O.pack_conv_weight(w, torch.zeros(cout), dev)
    res = []
    for cfg in [-1] + [(t << 4) | v for t in range(16) for v in range(8)]:
        if cfg < 0:
            os.environ.pop("BSY_CONV_CFG", None)
        else:
            os.environ["BSY_CONV_CFG"] = str(cfg)
        try:
            out = O.conv2d_nhwc(x, wp, bp, cout, k, s, True)
        except Exception:
            continue
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                O.conv2d_nhwc(x, wp, bp, cout, k, s, True, out=out)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        res.append((min(ts), max(ts), cfg))
    res.sort()
    fl = 2 * B * (H // s) * (W // s) * cout * cin * k * k
    print(f"{k}x{k}s{s} {cin}->{cout} @{H // s}: " + "  ".join(f"{c if c < 0 else hex(c)}:{a:.0f}-{b:.0f}us" for a, b, c in res[:6]) +
          f"   best {fl / res[0][0] / 1e6:.0f} TF/s", flush=True)
