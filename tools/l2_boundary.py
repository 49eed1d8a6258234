#!/usr/bin/env python3
"""Does a consumer kernel find its producer's output in the XCD's L2?  Two launches of the engine's own 1x1 conv kernel on the same
stream: A writes a 13-MB map (25 600 pixels x 256 channels, 1.6 MB per XCD under the library's XCD-contiguous tile order), B reads
it with the SAME pixel -> XCD mapping.  Run under `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum` (tools/pmc_l2_boundary.sh): if B's
misses equal its input lines, the L2 contents do not survive the kernel boundary (the runtime's agent-scope acquire / release between
dependent dispatches invalidates the non-coherent per-XCD L2s) and every layer's input comes from the Infinity Cache at best."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bs_yolo_amd import ops as O

dev = "cuda:0"
B, H, W, C = 64, 20, 20, 256
x = (torch.randn(B, H, W, C, device=dev) * 0.5).half()
w = torch.randn(C, C, 1, 1) * (2.0 / C) ** 0.5
wp, bp = O.pack_conv_weight(w, torch.zeros(C), dev)
a = O.conv2d_nhwc(x, wp, bp, C, 1, 1, True)
b = O.conv2d_nhwc(a, wp, bp, C, 1, 1, True)
torch.cuda.synchronize()
for _ in range(3):
    O.conv2d_nhwc(x, wp, bp, C, 1, 1, True, out=a)   # producer: writes `a`
    O.conv2d_nhwc(a, wp, bp, C, 1, 1, True, out=b)   # consumer: reads `a` right behind it
torch.cuda.synchronize()
print("input lines of the consumer (128 B):", B * H * W * C * 2 // 128)
