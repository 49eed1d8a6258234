#!/usr/bin/env python3
"""GPU-box exploration: per-layer error of the engine vs the oracle, per-op timings, quick throughput number."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bs_yolo_amd import lib as L  # noqa: E402
from bs_yolo_amd.engine import YoloEngine  # noqa: E402
from bs_yolo_amd.graphs import stock_cfg  # noqa: E402
from oracle import yolo_ref as R  # noqa: E402

DEV = "cuda:0"


def per_layer(tag="yolo11n_detect"):
    z = np.load(ROOT / "tests" / "golden" / f"graph_{tag}.npz")
    meta = json.loads(str(z["meta"]))
    m = R.Model(meta["family"], meta["scale"], meta["nc"], meta["task"])
    P = R.synth_params(m, meta["seed"])
    eng = YoloEngine(stock_cfg(meta["family"], meta["scale"], meta["nc"], meta["task"]), P)
    for dt in (torch.float32, torch.float16):
        x = torch.from_numpy(z["x0"])
        y, raws = eng(x.to(dt).to(DEV))
        torch.cuda.synchronize()
        plan, h = eng.plan_for(x.shape[0], x.shape[2], x.shape[3], dt, dt)
        print(f"== {tag} input {dt}")
        if "layer0_0" in z:
            for i, t in enumerate(plan.layer_out):
                if t is None:
                    continue
                ref = z[f"layer0_{i}"]
                if isinstance(t, list):
                    got = torch.cat([eng.read_view(plan, h, v) for v in t], 1).numpy()
                else:
                    got = eng.read_view(plan, h, t).numpy()
                err = np.abs(got - ref)
                print(f"  layer {i:2d} max|ref| {np.abs(ref).max():8.3f} max err {err.max():.3e} rel-to-max {err.max() / np.abs(ref).max():.2e}")
        si = 1
        while f"x{si}" in z:
            xx = torch.from_numpy(z[f"x{si}"])
            y2, _ = eng(xx.to(dt).to(DEV))
            torch.cuda.synchronize()
            e = np.abs(y2.float().cpu().numpy() - z[f"y{si}"])
            print(f"  input {si} {tuple(xx.shape)}: box max err {e[:, :4].max():.4f} score max err {e[:, 4:].max():.2e} (mean {e[:, 4:].mean():.2e})")
            R.FP16_EMULATION = True
            with torch.inference_mode():
                yq, _ = m.forward(P, xx)
            R.FP16_EMULATION = False
            e = np.abs(y2.float().cpu().numpy() - yq.numpy())
            print(f"     vs fp16-emulating oracle: box max err {e[:, :4].max():.4f} score max err {e[:, 4:].max():.2e} (mean {e[:, 4:].mean():.2e})")
            si += 1
        R.FP16_EMULATION = True
        with torch.inference_mode():
            yq, _ = m.forward(P, x)
        R.FP16_EMULATION = False
        e = np.abs(y.float().cpu().numpy() - yq.numpy())
        print(f"  input 0 vs fp16-emulating oracle: box max err {e[:, :4].max():.4f} score max err {e[:, 4:].max():.2e} (mean {e[:, 4:].mean():.2e})")
        yr = z["y0"]
        yy = y.float().cpu().numpy()
        print("  y box max err", np.abs(yy[:, :4] - yr[:, :4]).max(), " score max err", np.abs(yy[:, 4:] - yr[:, 4:]).max(),
              " max score", yr[:, 4:].max())
        for l in range(3):
            rr = z[f"raw0_{l}"]
            print(f"  raw{l} max err {np.abs(raws[l].float().cpu().numpy() - rr).max():.3e} (max |ref| {np.abs(rr).max():.2f})")
    eng.close()


def timing(scale="s", B=64, S=640, dt=torch.float16, family="yolo11", precision="fp16"):
    from bs_yolo_amd.plan import Plan
    from bs_yolo_amd.weights import synth_state_dict
    cfg = stock_cfg(family, scale, 12 if family == "bsyolo11" else 80)
    sd = synth_state_dict(Plan(cfg, 1, 64, 64), 0)
    eng = YoloEngine(cfg, sd, precision=precision)
    if precision in ("fp32", "fp32x"):
        dt = torch.float32
    x = torch.rand(B, 3, S, S, device=DEV).to(dt)
    for _ in range(3):
        y, _ = eng(x, want_raw=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        y, _ = eng(x, want_raw=False)
    torch.cuda.synchronize()
    dtm = (time.perf_counter() - t0) / n
    plan, _ = eng.plan_for(B, S, S, dt, dt)
    print(f"{family}{scale} B={B} {S}x{S}: {dtm * 1e3:.3f} ms/batch  {B / dtm:.0f} img/s  {plan.flops / dtm / 1e12:.1f} TFLOP/s")
    ops, plan = eng.profile(x)
    ops2, _ = eng.profile(x)
    tot = sum(t for _, _, t in ops2)
    print(f"sum of per-op times {tot:.3f} ms")
    rows = []
    bykind = {}
    for (name, kind, t), o in zip(ops2, plan.ops):
        fl = by = 0
        if kind in (L.OP_CONV, L.OP_DWCONV):
            cin = o["src0"].C + (o["src1"].C if o.get("src1") else 0)
            fl = 2 * B * o["OH"] * o["OW"] * o.get("cout", o["dst"].C) * cin * o["ksize"] ** 2 if kind == L.OP_CONV else 0
            ins = B * o["H"] * o["W"] * o["src0"].C * 2 // (4 if o["src0"].up else 1)
            if o.get("src1"):
                ins += B * o["H"] * o["W"] * o["src1"].C * 2 // (4 if o["src1"].up else 1)
            mode = o.get("out_f32", 0)
            outs = B * o["OH"] * o["OW"] * ({2: o.get("cout", 0), 3: 4}.get(mode, o["dst"].C)) * (4 if mode == 1 else 2)
            if o.get("res"):
                ins += outs
            by = ins + outs
        if kind == L.OP_CHAIN:
            fl = o["mfma_flops"]
        bykind[kind] = bykind.get(kind, 0.0) + t
        rows.append((t, name, kind, fl, by, o))
    print("time by kind (ms):", {k: round(v, 3) for k, v in bykind.items()})
    tun = dict(eng.tuning(B, S, S, dt))
    import collections
    print("tuned configs:", dict(collections.Counter(hex(v) for v in tun.values())))
    for t, name, kind, fl, by, o in rows:
        shape = ""
        if kind == L.OP_CONV:
            cin = o["src0"].C + (o["src1"].C if o.get("src1") else 0)
            shape = f"{o['ksize']}x{o['ksize']}s{o['stride']} {cin:4d}->{o.get('cout', o['dst'].C):4d} @{o['OH']}x{o['OW']}"
        if kind == L.OP_CHAIN:
            cin = o["src0"].C + (o["src1"].C if o.get("src1") else 0)
            shape = f"chain {cin}->{o['heads']}|{(o['box'][1].C if o['box'][1] else 0) + o['mid_c']}->{o['dst'].C} @{o['OH']}x{o['OW']}"
        print(f"  {t:8.4f} ms kind {kind} {name:26s} {shape:30s} {fl / t / 1e9 if t > 0 else 0:7.1f} TF/s {by / t / 1e6 if t > 0 else 0:8.1f} GB/s cfg {hex(tun.get(name, -1)) if name in tun else ''}")
    eng.close()


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("layers", "all"):
        for tag in (sys.argv[2:] or ["yolo11n_detect", "yolo11s_detect"]):
            per_layer(tag)
    if what in ("time", "all"):
        timing(os.environ.get("BSY_EXPLORE_SCALE", "s"), int(sys.argv[2]) if len(sys.argv) > 2 else 64, 640, family=sys.argv[3] if len(sys.argv) > 3 else "yolo11",
               precision=sys.argv[4] if len(sys.argv) > 4 else "fp16")
