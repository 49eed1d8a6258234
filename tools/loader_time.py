"""Throughput of the host input pipeline (bs_yolo_amd.loaders) from image files, alone and in front of YOLO11s:
python tools/loader_time.py [n_images=1024] [workers=16]   (run on the GPU box; writes its JPEGs under /tmp)"""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bs_yolo_amd import loaders as HL  # noqa: E402
from bs_yolo_amd.engine import YoloEngine  # noqa: E402
from bs_yolo_amd.graphs import stock_cfg  # noqa: E402
from bs_yolo_amd.plan import Plan  # noqa: E402
from bs_yolo_amd.weights import synth_state_dict  # noqa: E402


def main():
    from PIL import Image
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    d = Path(tempfile.mkdtemp(prefix="bsy_loader_"))
    rng = np.random.default_rng(0)
    base = rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)
    for i in range(n):  # smooth-ish content (upscaled noise): JPEG sizes comparable to photographs (~60-90 KB at 640 x 480)
        im = Image.fromarray(np.roll(base, i, 1)).resize((640, 480), Image.BICUBIC)
        im.save(d / f"{i:05d}.jpg", quality=90)
    size = sum(f.stat().st_size for f in d.iterdir()) / n
    cfg = stock_cfg("yolo11", "s")
    eng = YoloEngine(cfg, synth_state_dict(Plan(cfg, 1, 64, 64), seed=0))
    files = sorted(str(f) for f in d.iterdir()) * 4  # 4 passes over the files per measurement
    for mode in ("thread", "process"):
        loader = HL.LoadImagesPinned(files, batch=64, imgsz=640, workers=workers, depth=3, decode=mode)
        sum(b.im.shape[0] for b in loader)  # warm-up: worker start, arenas, page cache
        t0 = time.perf_counter()
        k = sum(b.im.shape[0] for b in loader)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"loader only, decode={mode}: {k / dt:.0f} img/s ({n} JPEGs 640x480, {size / 1e3:.0f} KB avg, {workers} workers)")
        list(HL.predict_stream(eng, loader))  # warm-up: plan, autotune
        t0 = time.perf_counter()
        k = 0
        for batch, det, counts in HL.predict_stream(eng, loader):
            k += len(batch.paths)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"files -> detections (YOLO11s, rect 480x640 batches of 64), decode={mode}: {k / dt:.0f} img/s")
        loader.close()


if __name__ == "__main__":
    main()
