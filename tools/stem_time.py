#!/usr/bin/env python3
"""Time the fused stem (layers 0 + 1) against the two separate launches, B = 64, 640 x 640."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bs_yolo_amd import ops as O
from bs_yolo_amd import lib as L
dev = "cuda:0"
for dt in (torch.float16, torch.float32):
    x = torch.rand(64, 3, 640, 640, device=dev).to(dt)
    w0, b0 = torch.randn(32, 3, 3, 3) * 0.3, torch.randn(32) * 0.1
    w1, b1 = torch.randn(64, 32, 3, 3) * 0.08, torch.randn(64) * 0.1
    w0p, b0p = O.pack_conv_weight(w0, b0, dev)
    w1p, b1p = O.pack_conv_weight(w1, b1, dev)
    mid = torch.empty(64, 320, 320, 32, dtype=torch.float16, device=dev)
    out = torch.empty(64, 160, 160, 64, dtype=torch.float16, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    def fused():
        L.check(L.lib.bsy_stem_fused(x.data_ptr(), L.dtype_code(dt), 64, 640, 640, w0p.data_ptr(), b0p.data_ptr(), 32,
                                     w1p.data_ptr(), b1p.data_ptr(), 64, out.data_ptr(), 64, 1, s))
    def two():
        L.check(L.lib.bsy_conv_first(x.data_ptr(), L.dtype_code(dt), 64, 640, 640, w0p.data_ptr(), b0p.data_ptr(),
                                     mid.data_ptr(), 32, 32, 3, 2, 1, s))
        O.conv2d_nhwc(mid, w1p, b1p, 64, 3, 2, True, out=out)
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
        ms = e0.elapsed_time(e1) / 20
        byts = x.numel() * x.element_size() + out.numel() * 2
        print(f"{dt} {name:13s}: {ms * 1e3:7.1f} us  ({byts / ms / 1e6:.0f} GB/s of image-in + layer-1-out bytes)", flush=True)
