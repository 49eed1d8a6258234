#!/usr/bin/env python3
"""Time the fused bottleneck (32 -> 16 -> 32 @160x160, B = 64) against the two conv launches."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bs_yolo_amd import ops as O
dev = "cuda:0"
B, H, W, c, ch, ld = 64, 160, 160, 32, 16, 96
buf = (torch.randn(B, H, W, ld, device=dev) * 0.5).half()
w1, b1 = torch.randn(ch, c, 3, 3) * 0.08, torch.randn(ch) * 0.1
w2, b2 = torch.randn(c, ch, 3, 3) * 0.1, torch.randn(c) * 0.1
w1p, b1p = O.pack_conv_weight(w1, b1, dev)
w2p, b2p = O.pack_conv_weight(w2, b2, dev)
x, y = buf[..., 32:64], buf[..., 64:96]
xc = x.contiguous()
mid = torch.empty(B, H, W, ch, dtype=torch.float16, device=dev)
out = torch.empty(B, H, W, c, dtype=torch.float16, device=dev)
from bs_yolo_amd import lib as L
st = torch.cuda.current_stream().cuda_stream
def fused():
    L.check(L.lib.bsy_bottleneck_fused(x.data_ptr(), ld, B, H, W, c, ch, w1p.data_ptr(), b1p.data_ptr(), w2p.data_ptr(),
                                       b2p.data_ptr(), y.data_ptr(), ld, 1, st))
def two():
    O.conv2d_nhwc(xc, w1p, b1p, ch, 3, 1, True, out=mid)
    O.conv2d_nhwc(mid, w2p, b2p, c, 3, 1, True, res=xc, out=out)
import time
for name, fn in (("fused", fused), ("two launches", two), ("fused", fused)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:13s}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us", flush=True)
