#!/usr/bin/env python3
"""A known-good reference on the same hardware (guide rule 10: never infer a platform ceiling from your own attempts): the vendor
libraries on the SHAPES of this forward's dense conv layers -- hipBLASLt / rocBLAS through torch.matmul for the 1x1 layers (a plain
GEMM: pixels x K times K x Cout, no bias, no activation) and MIOpen through torch.nn.functional.conv2d (channels_last fp16) for the
3x3 layers -- next to this library's own kernels on the same box (plain Conv+bias+SiLU launches through bsy_conv2d, untuned heuristic
configuration unless BSY_CONV_CFG says otherwise).  Not part of the product: nothing here is linked or called by it."""
import os
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bs_yolo_amd import ops as O

dev = "cuda:0"
B = 64
LAYERS = [  # name, H (input), Cin, Cout, k, s          (YOLO11s @ 640 x 640, batch 64)
    ("model.3", 160, 128, 128, 3, 2), ("model.4.cv1", 80, 128, 128, 1, 1), ("model.4.cv2", 80, 192, 256, 1, 1), ("model.5", 80, 256, 256, 3, 2),
    ("model.6.cv1", 40, 256, 256, 1, 1), ("model.6.cv2", 40, 384, 256, 1, 1), ("model.7", 40, 256, 512, 3, 2), ("model.8.cv1", 20, 512, 512, 1, 1),
    ("model.8.m.0.m.0.cv1", 20, 128, 128, 3, 1), ("model.8.cv2", 20, 768, 512, 1, 1), ("model.9.cv2", 20, 1024, 512, 1, 1), ("model.13.cv1", 40, 768, 256, 1, 1),
    ("model.16.cv1", 80, 512, 128, 1, 1), ("model.17", 80, 128, 128, 3, 2), ("model.23.cv2.0.0", 80, 128, 64, 3, 1), ("model.6.m.0.m.0.cv1", 40, 64, 64, 3, 1),
]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e-3)
    return sorted(ts)[2]


print(f"{'layer':22s} {'shape':26s} {'GFLOP':>6s} | {'this library':>20s} | {'vendor library':>34s}")
for name, H, cin, cout, k, s in LAYERS:
    OH = H // s
    fl = 2.0 * B * OH * OH * cout * cin * k * k
    x = (torch.randn(B, H, H, cin, device=dev) * 0.5).half()
    w = torch.randn(cout, cin, k, k) * (2.0 / (cin * k * k)) ** 0.5
    wp, bp = O.pack_conv_weight(w, torch.zeros(cout), dev)
    out = O.conv2d_nhwc(x, wp, bp, cout, k, s, True)
    t_ours = timeit(lambda: O.conv2d_nhwc(x, wp, bp, cout, k, s, True, out=out))
    if k == 1:
        a = x.view(-1, cin)
        wt = w.view(cout, cin).t().contiguous().half().to(dev)
        t_lib = timeit(lambda: torch.matmul(a, wt))
        what = "torch.matmul (hipBLASLt / rocBLAS), no bias / act"
    else:
        xc = x.permute(0, 3, 1, 2)  # NCHW view of the NHWC data = channels_last
        wc = w.half().to(dev).contiguous(memory_format=torch.channels_last)
        t_lib = timeit(lambda: F.conv2d(xc, wc, None, s, k // 2))
        what = "F.conv2d channels_last (MIOpen), no bias / act"
    print(f"{name:22s} {k}x{k}s{s} {cin:4d}->{cout:4d} @{OH:3d}x{OH:<3d} {fl / 1e9:6.1f} | {t_ours * 1e6:7.1f} us {fl / t_ours / 1e12:5.0f} TFLOP/s | {t_lib * 1e6:7.1f} us {fl / t_lib / 1e12:5.0f} TFLOP/s  {what}")
