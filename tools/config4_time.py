"""BASELINE config 4 on one GPU: YOLOv8l-seg, 640 x 640 fp16, batch 32 -- engine forward (Segment head + prototype convs)
+ batched NMS with 32 mask coefficients + process_mask for every image.  Usage: python tools/config4_time.py"""
import math
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bs_yolo_amd import masks as HM, nms as HN  # noqa: E402
from bs_yolo_amd.engine import YoloEngine  # noqa: E402
from bs_yolo_amd.graphs import stock_cfg  # noqa: E402
from bs_yolo_amd.plan import Plan  # noqa: E402
from bs_yolo_amd.weights import synth_state_dict  # noqa: E402

DEV = "cuda:0"


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    B = 32
    cfg = stock_cfg("yolov8", "l", 80, "segment")
    sd = synth_state_dict(Plan(cfg, 1, 64, 64), seed=0)
    sd = {k: (v * 0.8 if k.endswith("bn.weight") else v) for k, v in sd.items()}  # keep the deep graph's activations O(1)
    eng = YoloEngine(cfg, sd)
    x = torch.rand(B, 3, 640, 640, generator=torch.Generator().manual_seed(4)).half().to(DEV)
    # calibrate the class head from raw logits so that ~1.5 % of the anchors pass conf 0.25 (as bench.py does)
    _, (raws, _, _) = eng(x[:8], want_raw=True)
    bias0 = float(next(v for k, v in sd.items() if ".cv3." in k and k.endswith(".2.bias")).flatten()[0])
    lc = torch.cat([r[:, 64:64 + 80].float().flatten(2) for r in raws], 2) - bias0
    gain = 1.0 / max(float(lc.std()), 1e-6)
    q = float(torch.quantile((lc * gain).amax(1).flatten().cpu(), 1.0 - 0.015))
    for k in list(sd):
        if ".cv3." in k and k.endswith(".2.weight"):
            sd[k] = sd[k] * gain
        elif ".cv3." in k and k.endswith(".2.bias"):
            sd[k] = torch.full_like(sd[k], math.log(0.25 / 0.75) - q)
    eng.close()
    eng = YoloEngine(cfg, sd)
    plan = eng.plan_for(B, 640, 640, torch.float16, torch.float16)[0]

    def forward():
        return eng(x, want_raw=False)

    def full():
        y, (_, _, proto) = forward()
        det, counts = HN.nms_batched(y, 0.25, 0.7, max_det=300, nc=80)
        n = counts.tolist()
        return [HM.process_mask(proto[b], det[b, :n[b], 6:], det[b, :n[b], :4], (640, 640), upsample=True) for b in range(B)], n

    f = timed(forward)
    t = timed(full, 5)
    masks, n = full()
    print(f"YOLOv8l-seg bs {B} 640x640 fp16: forward {f:.3f} ms ({plan.flops / f / 1e9:.0f} TFLOP/s), forward + NMS + "
          f"process_mask(upsample) {t:.3f} ms = {B / t * 1e3:.0f} img/s; {sum(n) / B:.1f} detections / image, "
          f"masks {tuple(masks[0].shape)}")


if __name__ == "__main__":
    main()
