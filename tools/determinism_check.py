"""Bit-reproducibility of the product schedule at the headline size: the same batch through the engine N times (head
branches on their six side streams), every prediction tensor compared bit for bit with the first.
Usage: python tools/determinism_check.py [n=30]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bs_yolo_amd.engine import YoloEngine  # noqa: E402
from bs_yolo_amd.graphs import stock_cfg  # noqa: E402
from bs_yolo_amd.plan import Plan  # noqa: E402
from bs_yolo_amd.weights import synth_state_dict  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for fam, sc, nc, B, S in (("yolo11", "s", 80, 64, 640), ("bsyolo11", "s", 12, 64, 640), ("yolo11", "m", 80, 8, 1280), ("bsyolo11", "n", 12, 4, 1024)):
    cfg = stock_cfg(fam, sc, nc)
    eng = YoloEngine(cfg, synth_state_dict(Plan(cfg, 1, 64, 64), seed=0))
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(3)).half().cuda()
    y0, r0 = eng(x, want_raw=True)
    y0 = y0.clone(); r0 = [r.clone() for r in r0]
    bad = 0
    for _ in range(n):
        y, r = eng(x, want_raw=True)
        bad += int(not torch.equal(y, y0)) + sum(int(not torch.equal(a, b)) for a, b in zip(r, r0))
    torch.cuda.synchronize()
    print(f"{fam}{sc} B={B} {S}x{S}: {n} reruns, {bad} differing tensors")
    eng.close()
