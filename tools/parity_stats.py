"""End-to-end error statistics of the fp16-storage engine (max / 99.9th percentile / mean of |dscore| and |dbox|): against the
reference's own fp32 outputs (golden fixtures), against the oracle run with fp16 storage emulation, and -- at the benchmark's
full size -- against the engine's own fp32 correctness mode.  Prints one line per case; the bounds of
tests/test_gpu_parity.py::test_engine_matches_reference_golden are set from these.  Usage (GPU box): python tools/parity_stats.py"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bs_yolo_amd.engine import YoloEngine  # noqa: E402
from bs_yolo_amd.graphs import stock_cfg  # noqa: E402
from bs_yolo_amd.plan import Plan  # noqa: E402
from bs_yolo_amd.weights import synth_state_dict  # noqa: E402
from oracle import yolo_ref as R  # noqa: E402


def st(e):
    e = np.asarray(e).ravel()
    return f"max {e.max():.2e} p99.9 {np.quantile(e, 0.999):.2e} mean {e.mean():.2e}"


for tag in ["yolo11n_detect", "yolo11s_detect", "yolo11m_detect", "yolo11n_segment", "yolov8n_segment", "bsyolo11n_detect", "bsyolo11s_detect"]:
    z = np.load(ROOT / "tests" / "golden" / f"graph_{tag}.npz")
    meta = json.loads(str(z["meta"]))
    m = R.Model(meta["family"], meta["scale"], meta["nc"], meta["task"])
    P = R.synth_params(m, meta["seed"])
    cfg = stock_cfg(meta["family"], meta["scale"], meta["nc"], meta["task"])
    for dt in (torch.float16, torch.float32):
        eng = YoloEngine(cfg, P)
        si, nc = 0, meta["nc"]
        while f"x{si}" in z:
            x = torch.from_numpy(z[f"x{si}"])
            y = eng(x.to(dt).to("cuda:0"))[0].float().cpu().numpy()
            yr = z[f"y{si}"]
            R.FP16_EMULATION = True
            with torch.inference_mode():
                yq = m.forward(P, x)[0].numpy()
            R.FP16_EMULATION = False
            print(f"{tag} in={str(dt)[6:]} x{si} {tuple(x.shape)}: vs ref score {st(np.abs(y[:, 4:4+nc]-yr[:, 4:4+nc]))} | box {st(np.abs(y[:, :4]-yr[:, :4]))}")
            print(f"{'':40s} vs emu score {st(np.abs(y[:, 4:4+nc]-yq[:, 4:4+nc]))} | box {st(np.abs(y[:, :4]-yq[:, :4]))}")
            si += 1
        eng.close()

# the benchmark configuration itself: YOLO11s, 64 x 640 x 640, bench weights -- fp16 product path vs fp32 correctness mode
cfg = stock_cfg("yolo11", "s")
sd = synth_state_dict(Plan(cfg, 1, 64, 64), seed=0)
x = torch.rand(64, 3, 640, 640, generator=torch.Generator().manual_seed(1234)).half().to("cuda:0")
e16, e32 = YoloEngine(cfg, sd), YoloEngine(cfg, sd, precision="fp32")
y16 = e16(x)[0].float()
y32 = e32(x.float())[0]
torch.cuda.synchronize()
d = (y16 - y32).abs().cpu().numpy()
print(f"bench config 64x640x640 fp16 engine vs fp32 mode: score {st(d[:, 4:])} | box {st(d[:, :4])}; anchors above conf 0.25: fp16 {int((y16[:, 4:].amax(1) > 0.25).sum())} fp32 {int((y32[:, 4:].amax(1) > 0.25).sum())}")
