#!/usr/bin/env python3
"""How much would L2-resident activations buy a staging-bound layer?  The same FLOPs, tile shape (256 x 256, BK 64) and tile count,
once with the pixel operand coming from beyond L2 (64 images of 20 x 20, 512 couts: every pixel tile is read by 2 cout tiles -- the
product's model.8.cv1 / model.10.cv1 shape) and once with it L2-resident for 15 of its 16 reads (8 images, 4096 couts: 16 cout tiles
per pixel tile, the XCD-contiguous tile order keeps them on one XCD).  If the second form were much faster, a persistent stage kernel
with L2-resident hand-offs would be worth building; it is not (profiles/r04_l2_resident_gemm.txt)."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bs_yolo_amd import ops as O

dev = "cuda:0"
os.environ["BSY_CONV_CFG"] = str((7 << 4) | 3)
for K in (512, 768, 1024):
    for (B, cout, what) in ((64, 512, "pixels from beyond L2 (2 cout tiles per pixel tile)"), (8, 4096, "pixels L2-resident (16 cout tiles per pixel tile)")):
        x = (torch.randn(B, 20, 20, K, device=dev) * 0.5).half()
        w = torch.randn(cout, K, 1, 1) * (2.0 / K) ** 0.5
        wp, bp = O.pack_conv_weight(w, torch.zeros(cout), dev)
        out = O.conv2d_nhwc(x, wp, bp, cout, 1, 1, True)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                O.conv2d_nhwc(x, wp, bp, cout, 1, 1, True, out=out)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e-3)
        t = sorted(ts)[2]
        M = B * 400
        fl = 2.0 * M * cout * K
        tiles = -(-M // 256) * (cout // 256)
        print(f"K {K:4d}  {B:2d} images x {cout:4d} couts: {tiles} tiles of 256 x 256, {fl / 1e9:5.1f} GFLOP: {t * 1e6:6.1f} us = {fl / t / 1e12:5.0f} TFLOP/s   {what}")
