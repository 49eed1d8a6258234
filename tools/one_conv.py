#!/usr/bin/env python3
"""Run selected conv shapes a few times (for PMC collection / micro-timing).  usage: one_conv.py [iters]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bs_yolo_amd import ops as O

SHAPES = [  # (B, H, W, cin, cout, k, s)
    (64, 160, 160, 64, 64, 1, 1),     # model.2.cv1  (memory-bound)
    (64, 160, 160, 96, 128, 1, 1),    # model.2.cv2
    (64, 160, 160, 128, 128, 3, 2),   # model.3      (compute-bound)
    (64, 80, 80, 256, 256, 3, 2),     # model.5
    (64, 20, 20, 128, 128, 3, 1),     # 20x20 bottleneck conv
    (64, 20, 20, 512, 512, 1, 1),     # model.8.cv1
]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = "cuda:0"
for (B, H, W, cin, cout, k, s) in SHAPES:
    x = (torch.randn(B, H, W, cin, device=dev) * 0.5).half()
    w = torch.randn(cout, cin, k, k) * (2.0 / (cin * k * k)) ** 0.5
    wp, bp = O.pack_conv_weight(w, torch.zeros(cout), dev)
    out = None
    for _ in range(2):
        out = O.conv2d_nhwc(x, wp, bp, cout, k, s, True, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        O.conv2d_nhwc(x, wp, bp, cout, k, s, True, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2 * B * out.shape[1] * out.shape[2] * cout * cin * k * k
    by = x.numel() * 2 + out.numel() * 2
    print(f"{k}x{k}s{s} {cin:4d}->{cout:4d} @{out.shape[1]}x{out.shape[2]}: {ms:.4f} ms {fl / ms / 1e9:7.1f} TF/s {by / ms / 1e6:7.1f} GB/s", flush=True)
