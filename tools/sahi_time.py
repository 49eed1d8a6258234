"""Timing of the sliced-inference path (BASELINE config 5 on ONE GPU): tile extraction, YOLO11x forward over the 70
tiles, batched NMS, cross-tile merge.  Usage: python tools/sahi_time.py [scale]   (run on the GPU box)"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from bs_yolo_amd import nms as HN, sahi as HS  # noqa: E402
from bs_yolo_amd.engine import YoloEngine  # noqa: E402
from bs_yolo_amd.graphs import stock_cfg  # noqa: E402
from bs_yolo_amd.plan import Plan  # noqa: E402
from bs_yolo_amd.weights import synth_state_dict  # noqa: E402

DEV = "cuda:0"


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    scale = sys.argv[1] if len(sys.argv) > 1 else "x"
    P = synth_state_dict(Plan(stock_cfg("yolo11", scale), 1, 64, 64), seed=5)
    P = {n: (v * 0.8 if n.endswith("bn.weight") else v) for n, v in P.items()}  # keep the deep graph's activations O(1)
    img = torch.randint(0, 256, (4000, 6000, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(DEV)
    bb = HS.get_slice_bboxes(4000, 6000, 640, 640, 0, 0)
    ms = timed(lambda: HS.slice_image(img, bb, device=DEV))
    by = 70 * 640 * 640 * 3 * (1 + 2)
    print(f"slice_tiles 70 x 640x640: {ms:.4f} ms  {by / ms / 1e6:.0f} GB/s (tile bytes read + fp16 planes written)")
    tiles = HS.slice_image(img, bb, device=DEV)
    eng = YoloEngine(stock_cfg("yolo11", scale), P, autotune=True)
    y, raws = eng(tiles, want_raw=True)
    top = torch.cat([r[:, 64:].flatten(2) for r in raws], 2).amax(1).float().flatten()
    thr = float(torch.quantile(top[torch.randperm(top.numel(), device=top.device)[:1_000_000]], 0.99))
    fwd = timed(lambda: eng(tiles, want_raw=False), 10)
    gf = eng.plan_for(70, 640, 640, torch.float16, torch.float16)[0].flops / 1e9
    print(f"YOLO11{scale} forward, 70 tiles: {fwd:.3f} ms  ({gf / fwd:.0f} TFLOP/s)")
    y[:, 4:] = torch.sigmoid(torch.logit(y[:, 4:].float().clamp(1e-6, 1 - 1e-6)) - thr + float(np.log(0.25 / 0.75))).half()
    nms = timed(lambda: HN.nms_batched(y, 0.25, 0.7, max_det=300, in_place=False))
    det, cnt = HN.nms_batched(y, 0.25, 0.7, max_det=300, in_place=False)
    sh = torch.tensor([[b[0], b[1]] for b in bb], dtype=torch.float32, device=DEV)
    mg = timed(lambda: HS.postprocess(det, cnt, sh, full_shape=(4000, 6000)))
    out, n = HS.postprocess(det, cnt, sh, full_shape=(4000, 6000))
    print(f"NMS 70 tiles: {nms:.3f} ms   cross-tile GREEDYNMM: {mg:.3f} ms  ({int(cnt.sum())} tile detections -> {int(n)})")
    t0 = time.perf_counter()
    for _ in range(5):
        o, _ = HS.get_sliced_prediction(img, eng, 640, 640, 0, 0, perform_standard_pred=False)
    torch.cuda.synchronize()
    print(f"get_sliced_prediction end to end (host clock, incl. the count read-back): {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms / image")


if __name__ == "__main__":
    main()
