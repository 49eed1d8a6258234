import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from bs_yolo_amd import val as HV
from oracle import val_ref as V
rng = np.random.default_rng(6)
n, m, ncls = 500_000, 60_000, 80
conf = rng.uniform(0.001, 1.0, n).astype(np.float32)
pred_cls = rng.integers(0, ncls, n).astype(np.float32)
target_cls = rng.integers(0, ncls, m).astype(np.float32)
tp = np.logical_and.accumulate(np.stack([rng.random(n) < 0.1 * conf * (1.0 - 0.07 * j) for j in range(10)], 1), 1)
d = [torch.from_numpy(a).cuda() for a in (tp, conf, pred_cls)]
HV.ap_per_class(*d, target_cls); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): HV.ap_per_class(*d, target_cls)
torch.cuda.synchronize(); t1 = time.perf_counter()
print("device ap_per_class, 500k detections x 10 thresholds, 80 classes: %.1f ms (incl. downloads + host max-F1)" % ((t1 - t0) / 3 * 1e3))
