#!/usr/bin/env python3
"""Bytes in flight vs staging rate at a CONSTANT tile: the implicit-GEMM kernel's ring variants of the 256 x 256
and 256 x 128 tiles -- BK 32 x 2 stages (one K-step = 32 / 24 KiB in flight per workgroup), BK 32 x 3 stages (two K-steps in flight),
BK 64 x 2 stages (one 64 / 48-KiB K-step in flight) -- on the large-K layers of YOLO11s at 64 images.  Prints time, TFLOP/s and the
bytes staged into LDS per clock and CU (2.4 GHz, the CUs the grid occupies).  Results are bit-identical across variants (one K walk)."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bs_yolo_amd import ops as O

SHAPES = [("model.5", 64, 80, 80, 256, 256, 3, 2), ("model.7", 64, 40, 40, 256, 512, 3, 2), ("model.8.cv2", 64, 20, 20, 768, 512, 1, 1),
          ("model.9.cv2", 64, 20, 20, 1024, 512, 1, 1), ("model.13.cv1", 64, 40, 40, 768, 256, 1, 1), ("model.3", 64, 160, 160, 128, 128, 3, 2)]
VARS = {2: "BK32 x 2 stages", 1: "BK32 x 3 stages", 3: "BK64 x 2 stages"}
TILES = {7: (256, 256, 8, 1), 4: (256, 128, 8, 2), 2: (128, 128, 4, 3)}  # tile id -> (pixels, couts, waves, workgroups per CU by LDS / registers)
dev = "cuda:0"
for (name, B, H, W, cin, cout, k, s) in SHAPES:
    x = (torch.randn(B, H, W, cin, device=dev) * 0.5).half()
    w = torch.randn(cout, cin, k, k) * (2.0 / (cin * k * k)) ** 0.5
    wp, bp = O.pack_conv_weight(w, torch.zeros(cout), dev)
    M = B * (H // s) * (W // s)
    K = cin * k * k
    fl = 2.0 * M * cout * K
    print(f"{name}: {k}x{k}s{s} {cin}->{cout} @{H // s}x{W // s}, {fl / 1e9:.1f} GFLOP")
    ref = None
    for tile, (tm, tn, waves, wgcu) in TILES.items():
        if cout % tn:
            continue
        for var, vname in VARS.items():
            os.environ["BSY_CONV_CFG"] = str((tile << 4) | var)
            try:
                out = O.conv2d_nhwc(x, wp, bp, cout, k, s, True)
            except Exception as e:
                continue
            if ref is None:
                ref = out.clone()
            same = torch.equal(out, ref)
            ts = []
            for _ in range(5):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    O.conv2d_nhwc(x, wp, bp, cout, k, s, True, out=out)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10 * 1e-3)
            t = sorted(ts)[len(ts) // 2]
            ntiles = -(-M // tm) * (cout // tn)
            staged = ntiles * K * (tm + tn) * 2.0           # bytes DMA'd into LDS by the whole launch
            bk = 64 if var == 3 else 32
            inflight = (tm + tn) * bk * 2 * (2 if var == 1 else 1) * min(wgcu, max(1, -(-ntiles // 256)))
            cus = min(256, ntiles)
            print(f"   tile {tm}x{tn} {vname:16s}: {t * 1e6:7.1f} us  {fl / t / 1e12:6.0f} TFLOP/s  staged {staged / 1e9:5.2f} GB = {staged / t / cus / 2.4e9:5.1f} B/clk/CU"
                  f"  (~{inflight // 1024} KiB in flight per CU, {ntiles} workgroups){'' if same else '  RESULT DIFFERS'}")
os.environ.pop("BSY_CONV_CFG", None)
