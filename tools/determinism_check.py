"""The product schedule (head branches on their six side streams) against a SERIAL schedule of the same plan (every op on
one stream, BSY_LANES=0): the same batch N times, every prediction tensor compared bit for bit with the serial result.
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
    # reference: the same plan with every op on ONE stream (BSY_LANES=0), same kernel configurations
    import os
    os.environ["BSY_LANES"] = "0"
    ser = YoloEngine(cfg, synth_state_dict(Plan(cfg, 1, 64, 64), seed=0), autotune=False)
    y0, r0 = ser(x, want_raw=True)
    y0 = y0.clone(); r0 = [r.clone() for r in r0]
    os.environ["BSY_LANES"] = "1"
    eng.close()
    eng = YoloEngine(cfg, synth_state_dict(Plan(cfg, 1, 64, 64), seed=0), autotune=False)
    assert max(o.get("lane", 0) for o in eng.plan_for(B, S, S, torch.float16, torch.float16)[0].ops) > 0
    assert max(o.get("lane", 0) for o in ser.plan_for(B, S, S, torch.float16, torch.float16)[0].ops) == 0
    bad = 0
    for _ in range(n):
        y, r = eng(x, want_raw=True)
        bad += int(not torch.equal(y, y0)) + sum(int(not torch.equal(a, b)) for a, b in zip(r, r0))
    torch.cuda.synchronize()
    print(f"{fam}{sc} B={B} {S}x{S}: {n} concurrent-lane forwards vs the serial schedule: {bad} differing tensors")
    ser.close()
    eng.close()
