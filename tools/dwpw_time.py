#!/usr/bin/env python3
"""Time the fused DWConv 3x3 + 1x1 units of YOLO11s's class branch (B = 64): (H, C, Cout) of cv3.0.0 / cv3.0.1 / cv3.1.0 / cv3.1.1 / cv3.2.1."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bs_yolo_amd import ops as O
from bs_yolo_amd import lib as L
dev = "cuda:0"
for H, c, c2 in ((80, 128, 128), (40, 256, 128), (40, 128, 128), (20, 128, 128)):
    x = (torch.randn(64, H, H, c, device=dev) * 0.5).half()
    wd, bd = torch.randn(c, 1, 3, 3) * 0.4, torch.randn(c) * 0.2
    w, b = torch.randn(c2, c, 1, 1) * (2.0 / c) ** 0.5, torch.randn(c2) * 0.2
    wdp = wd.float().view(c, 9).t().contiguous().to(dev)
    bdp = bd.float().to(dev)
    wp, bp = O.pack_conv_weight(w, b, dev)
    out = torch.empty((64, H, H, c2), dtype=torch.float16, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        L.check(L.lib.bsy_dwpw_fused(x.data_ptr(), c, 64, H, H, c, wdp.data_ptr(), bdp.data_ptr(), wp.data_ptr(), bp.data_ptr(), out.data_ptr(), c2, c2, 1, st))
    best = 1e9
    for _ in range(3):
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print(f"dwpw {H}x{H} {c}->{c2}: {best * 1e3:7.1f} us", flush=True)
