#!/usr/bin/env python3
"""One stride-2 3x3 conv (model.3 / model.5 of YOLO11s, B = 64) under fixed configs, for PMC traffic experiments."""
import os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bs_yolo_amd import ops as O
dev = "cuda:0"
for (H, cin, cout) in ((160, 128, 128), (80, 256, 256)):
    x = (torch.randn(64, H, H, cin, device=dev) * 0.5).half()
    w = torch.randn(cout, cin, 3, 3) * (2.0 / (cin * 9)) ** 0.5
    wp, bp = O.pack_conv_weight(w, torch.zeros(cout), dev)
    for cfg in (0x23, 0x41, 0x21):
        os.environ["BSY_CONV_CFG"] = str(cfg)
        out = O.conv2d_nhwc(x, wp, bp, cout, 3, 2, True)
        for _ in range(3):
            O.conv2d_nhwc(x, wp, bp, cout, 3, 2, True, out=out)
        torch.cuda.synchronize()
