#!/usr/bin/env python3
"""fp32x mode on the GPU box: per-layer error of conv32x against an fp64 reference and against the exact fp32 kernel, kernel times,
and the whole-graph parity / throughput of the three precisions on the golden graphs and at the benchmark size."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bs_yolo_amd import ops as O  # noqa: E402
from bs_yolo_amd.engine import YoloEngine  # noqa: E402
from bs_yolo_amd.graphs import stock_cfg  # noqa: E402
from bs_yolo_amd.plan import Plan  # noqa: E402
from bs_yolo_amd.weights import synth_state_dict  # noqa: E402

DEV = "cuda:0"


def layer_cases():
    for (B, H, W, cin, cout, k, s) in [(8, 80, 80, 128, 128, 3, 2), (8, 40, 40, 256, 256, 3, 2), (8, 20, 20, 768, 512, 1, 1), (8, 80, 80, 192, 256, 1, 1),
                                       (8, 40, 40, 64, 64, 3, 1), (8, 160, 160, 32, 16, 3, 1), (8, 20, 20, 128, 80, 1, 1)]:
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, H, W, cin, generator=g).abs() * 0.7  # post-SiLU-like magnitudes, some tiny
        x[..., ::3] *= 0.01
        w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
        b = torch.randn(cout, generator=g) * 0.3
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), s, k // 2).permute(0, 2, 3, 1)
        xd = x.to(DEV)
        ye = O.conv2d_nhwc_f32(xd, w, b, k, s, False, impl=2)[..., :cout]
        yx = O.conv2d_nhwc_f32x(xd, w, b, k, s, False)[..., :cout]
        torch.cuda.synchronize()
        rng = float(ref.abs().max())
        ee = float((ye.cpu().double() - ref).abs().max()) / rng
        ex = float((yx.cpu().double() - ref).abs().max()) / rng
        # timing
        def t(fn, n=20):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n
        # weights are re-packed per call in the ops wrappers: time through events around the C call instead
        fl = 2 * B * (H // s) * (W // s) * cout * cin * k * k
        print(f"conv {k}x{k}s{s} {cin}->{cout} @{H}x{W} B{B}: exact err {ee:.2e} of range, fp32x err {ex:.2e} of range ({fl / 1e9:.1f} GFLOP)")


def graphs():
    from oracle import yolo_ref as R
    for tag in ("yolo11n_detect", "yolo11s_detect", "yolo11m_detect", "bsyolo11n_detect", "yolov8n_segment"):
        z = np.load(ROOT / "tests" / "golden" / f"graph_{tag}.npz")
        meta = json.loads(str(z["meta"]))
        m = R.Model(meta["family"], meta["scale"], meta["nc"], meta["task"])
        P = R.synth_params(m, meta["seed"])
        cfg = stock_cfg(meta["family"], meta["scale"], meta["nc"], meta["task"])
        for prec in ("fp32", "fp32x"):
            eng = YoloEngine(cfg, P, precision=prec)
            si, es, eb = 0, 0.0, 0.0
            while f"x{si}" in z:
                x = torch.from_numpy(z[f"x{si}"])
                y = eng(x.to(DEV))[0].cpu().numpy()
                yr = z[f"y{si}"]
                nc = meta["nc"]
                es = max(es, float(np.abs(y[:, 4:4 + nc] - yr[:, 4:4 + nc]).max()))
                eb = max(eb, float(np.abs(y[:, :4] - yr[:, :4]).max()) / max(x.shape[2], x.shape[3]))
                si += 1
            print(f"{tag:20s} {prec:6s}: score max err {es:.2e}   box max err / imgsz {eb:.2e}")
            eng.close()


def bench_size():
    cfg = stock_cfg("yolo11", "s")
    sd = synth_state_dict(Plan(cfg, 1, 64, 64), seed=0)
    x = torch.rand(64, 3, 640, 640, generator=torch.Generator().manual_seed(1234)).to(DEV)
    ys = {}
    for prec in ("fp32", "fp32x", "fp16"):
        eng = YoloEngine(cfg, sd, precision=prec)
        xi = x.half() if prec == "fp16" else x
        for _ in range(3):
            y = eng(xi, want_raw=False)[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            y = eng(xi, want_raw=False)[0]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        ys[prec] = y.float()
        print(f"YOLO11s 64 x 640^2 {prec:6s}: {dt * 1e3:7.3f} ms / forward = {64 / dt:8.0f} img/s")
        if prec == "fp32x":
            ops, plan = eng.profile(xi)
            ops, plan = eng.profile(xi)
            tot = sum(t for _, _, t in ops)
            conv = sum(t for (nm, kind, t) in ops if kind in (1, 7))
            print(f"   per-op sum {tot:.3f} ms, conv kinds {conv:.3f} ms; slowest ops:")
            for nm, kind, t in sorted(ops, key=lambda r: -r[2])[:14]:
                print(f"      {t:7.4f} ms kind {kind} {nm}")
        eng.close()
    for a in ("fp32x", "fp16"):
        ds, db = (ys[a][:, 4:] - ys["fp32"][:, 4:]).abs(), (ys[a][:, :4] - ys["fp32"][:, :4]).abs()
        print(f"{a} vs fp32 mode at the benchmark size: score max {float(ds.max()):.2e} mean {float(ds.mean()):.2e}; box max {float(db.max()):.2e} px mean {float(db.mean()):.2e}")


if __name__ == "__main__":
    what = sys.argv[1:] or ["layers", "graphs", "bench"]
    if "layers" in what:
        layer_cases()
    if "graphs" in what:
        graphs()
    if "bench" in what:
        bench_size()
