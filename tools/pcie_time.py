"""PCIe-inclusive rate of the headline workload: the 64-image fp16 batch (157 MB) starts in PINNED HOST memory every step
(the reference's LoadTensor source, data/loaders.py:516-584, hands over host tensors), uploaded on a copy stream one batch
ahead of the forward + NMS.  bench.py's `value` keeps inputs resident in HBM; this is the figure DESIGN.md quotes beside it.
Usage: python tools/pcie_time.py [steps=40]"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bs_yolo_amd import nms as HN  # noqa: E402
from bs_yolo_amd.engine import YoloEngine  # noqa: E402
from bs_yolo_amd.graphs import stock_cfg  # noqa: E402
from bs_yolo_amd.plan import Plan  # noqa: E402
from bs_yolo_amd.weights import synth_state_dict  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda:0")
    cfg = stock_cfg("yolo11", "s")
    eng = YoloEngine(cfg, synth_state_dict(Plan(cfg, 1, 64, 64), seed=0))
    host = [torch.rand(64, 3, 640, 640, generator=torch.Generator().manual_seed(i)).half().pin_memory() for i in range(2)]
    devb = [torch.empty_like(h, device=dev) for h in host]
    copy = torch.cuda.Stream(dev)
    up = [torch.cuda.Event() for _ in range(2)]
    used = [torch.cuda.Event() for _ in range(2)]

    def upload(i):
        with torch.cuda.stream(copy):
            copy.wait_event(used[i])  # the forward that read this device buffer is done
            devb[i].copy_(host[i], non_blocking=True)
            up[i].record(copy)

    def run(n, overlap):
        cur = torch.cuda.current_stream(dev)
        for i in range(2):
            used[i].record(cur)
        upload(0)
        for k in range(n):
            i = k & 1
            if overlap and k + 1 < n:
                upload(i ^ 1)
            cur.wait_event(up[i])
            y, _ = eng(devb[i], want_raw=False)
            HN.nms_batched(y, 0.25, 0.7, max_det=300)
            used[i].record(cur)
            if not overlap and k + 1 < n:
                upload(i ^ 1)
        torch.cuda.synchronize()

    run(5, True)
    for overlap in (False, True):
        t0 = time.perf_counter()
        run(steps, overlap)
        dt = time.perf_counter() - t0
        print(f"host-resident input, upload {'overlapped with' if overlap else 'serialised before'} the forward: "
              f"{dt / steps * 1e3:.3f} ms / step = {64 * steps / dt:.0f} img/s")
    x = devb[0]
    t0 = time.perf_counter()
    for _ in range(steps):
        y, _ = eng(x, want_raw=False)
        HN.nms_batched(y, 0.25, 0.7, max_det=300)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"HBM-resident input (bench.py's value): {dt / steps * 1e3:.3f} ms / step = {64 * steps / dt:.0f} img/s")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        devb[0].copy_(host[0], non_blocking=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"upload alone: {ms:.3f} ms per 157 MB batch = {host[0].numel() * 2 / ms / 1e6:.1f} GB/s")


if __name__ == "__main__":
    main()
