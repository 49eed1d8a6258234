"""Does a depthwise kernel (packed-f32 FMAs, no LDS) return wrong values when MFMA kernels share the CUs from another
stream?  (the early-head experiment.)  SCDown's 3x3 stride-2 depthwise conv on stream A, compared bit
for bit with its result when run alone, while stream B runs patch convs / fused DWConv+1x1 back to back.
Usage: python tools/concurrency_stress.py [iters=300]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bs_yolo_amd import ops as O  # noqa: E402

dev = "cuda:0"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
g = torch.Generator().manual_seed(0)
C = 128
x = (torch.randn(1, 64, 64, C, generator=g) * 30).half().to(dev)           # NHWC, like model.24.cv1's output
wd = (torch.randn(C, 1, 3, 3, generator=g) * 0.5).to(dev)
bd = torch.randn(C, generator=g).to(dev)
ref = O.dwconv_nhwc(x, wd, bd, stride=2, act=False).clone()
torch.cuda.synchronize()
# the other stream's work: the P4 head's kernels (3x3 patch convs 128 -> 64 -> 64 and a fused DWConv + 1x1) on a 64 x 64 map
xh = (torch.randn(1, 64, 64, 128, generator=g)).half().to(dev)
w1 = torch.randn(64, 128, 3, 3, generator=g) * 0.03
w2 = torch.randn(64, 64, 3, 3, generator=g) * 0.04
wp1, bp1 = O.pack_conv_weight(w1, torch.zeros(64), dev)
wp2, bp2 = O.pack_conv_weight(w2, torch.zeros(64), dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
for mode in ("alone", "beside MFMA kernels"):
    bad = 0
    for i in range(iters):
        if mode != "alone":
            with torch.cuda.stream(sB):
                for _ in range(3):
                    t = O.conv2d_nhwc(xh, wp1, bp1, 64, 3, 1, True)
                    t = O.conv2d_nhwc(t, wp2, bp2, 64, 3, 1, True)
        with torch.cuda.stream(sA):
            out = O.dwconv_nhwc(x, wd, bd, stride=2, act=False)
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
            if bad <= 3:
                idx = (out != ref).nonzero()
                print("  mismatch at iter", i, "n", idx.shape[0], "first (n,y,x,c):", idx[:6].tolist())
    print(f"depthwise 3x3 s2 {mode}: {bad} of {iters} runs differ from the reference result")
