"""CPU restatement of the validator's detection-to-label matching.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Follows /root/reference/ultralytics:
  box_iou            utils/metrics.py:52-70
  match_predictions  engine/validator.py:222-258 (the default, non-scipy branch), as called by
                     DetectionValidator._process_batch (models/yolo/detect/val.py:209-228):
                         iou = box_iou(gt_bboxes, detections[:, :4]); match_predictions(detections[:, 5], gt_cls, iou)
Parity unpinned for EXACT IoU ties between two labels of one detection: the reference orders matches with
numpy's default (unstable) argsort reversed, so which of two equal entries comes first is an implementation accident.
"""
from __future__ import annotations

import numpy as np

IOUV = np.linspace(0.5, 0.95, 10).astype(np.float32)  # validator: torch.linspace(0.5, 0.95, 10) (detect/val.py:36)


def box_iou(box1: np.ndarray, box2: np.ndarray, eps: float = 1e-7) -> np.ndarray:
    """(N,4) x (M,4) xyxy -> (N,M), fp32 like the reference (`.float()`)."""
    box1, box2 = np.asarray(box1, np.float32), np.asarray(box2, np.float32)
    a1, a2 = box1[:, None, :2], box1[:, None, 2:]
    b1, b2 = box2[None, :, :2], box2[None, :, 2:]
    inter = np.clip(np.minimum(a2, b2) - np.maximum(a1, b1), 0, None).prod(2)
    return inter / ((a2 - a1).prod(2) + (b2 - b1).prod(2) - inter + np.float32(eps))


def match_predictions(pred_classes: np.ndarray, true_classes: np.ndarray, iou: np.ndarray, iouv: np.ndarray = IOUV) -> np.ndarray:
    """pred_classes (N,), true_classes (M,), iou (M,N) -> correct (N, len(iouv)) bool."""
    correct = np.zeros((pred_classes.shape[0], iouv.shape[0]), dtype=bool)
    correct_class = true_classes[:, None] == pred_classes
    iou = iou * correct_class
    for i, threshold in enumerate(iouv.tolist()):
        matches = np.nonzero(iou >= threshold)
        matches = np.array(matches).T
        if matches.shape[0]:
            if matches.shape[0] > 1:
                matches = matches[iou[matches[:, 0], matches[:, 1]].argsort()[::-1]]
                matches = matches[np.unique(matches[:, 1], return_index=True)[1]]
                matches = matches[np.unique(matches[:, 0], return_index=True)[1]]
            correct[matches[:, 1].astype(int), i] = True
    return correct


def process_batch(detections: np.ndarray, gt_bboxes: np.ndarray, gt_cls: np.ndarray, iouv: np.ndarray = IOUV) -> np.ndarray:
    """detect/val.py:209-228."""
    return match_predictions(detections[:, 5], gt_cls, box_iou(gt_bboxes, detections[:, :4]), iouv)


def _interp(x, xp, fp, left=None, right=None):
    """numpy.interp restated (what ap_per_class / compute_ap rely on): j = last index with xp[j] <= x; x below xp[0] ->
    left (default fp[0]); above xp[-1] -> right (default fp[-1]); on a knot or on the last knot -> fp[j]; else
    slope * (x - xp[j]) + fp[j].  xp non-decreasing (duplicates allowed)."""
    xp = np.asarray(xp, dtype=np.float64); fp = np.asarray(fp, dtype=np.float64)
    out = np.empty(len(x), dtype=np.float64)
    lo = fp[0] if left is None else left
    hi = fp[-1] if right is None else right
    for i, v in enumerate(np.asarray(x, dtype=np.float64)):
        if v < xp[0]:
            out[i] = lo
        elif v > xp[-1]:
            out[i] = hi
        else:
            j = int(np.searchsorted(xp, v, side="right")) - 1
            if j == len(xp) - 1 or xp[j] == v:
                out[i] = fp[j]
            else:
                out[i] = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]) * (v - xp[j]) + fp[j]
    return out


def compute_ap(recall, precision):
    """utils/metrics.py:588-617: sentinels (0, 1) / (1, 0), precision envelope (running max from the right), 101-point
    interpolation integrated by the trapezoid rule -> (ap, mpre, mrec)."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]
    x = np.linspace(0, 1, 101)
    y = _interp(x, mrec, mpre)
    ap = float(np.sum((x[1:] - x[:-1]) * (y[1:] + y[:-1]) / 2.0))
    return ap, mpre, mrec


def ap_per_class(tp, conf, pred_cls, target_cls, eps=1e-16):
    """utils/metrics.py:620-706 without the plots -> the same 12-tuple.  Detections are ranked by confidence (ties: lower
    index first; the reference's np.argsort(-conf) leaves them unpinned)."""
    tp = np.asarray(tp); conf = np.asarray(conf); pred_cls = np.asarray(pred_cls)
    order = np.argsort(-conf, kind="stable")
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    unique_classes, nt = np.unique(target_cls, return_counts=True)
    nc = unique_classes.shape[0]
    x = np.linspace(0, 1, 1000)
    ap = np.zeros((nc, tp.shape[1])); p_curve = np.zeros((nc, 1000)); r_curve = np.zeros((nc, 1000))
    prec_values = []
    for ci, c in enumerate(unique_classes):
        sel = pred_cls == c
        n_l, n_p = nt[ci], int(sel.sum())
        if n_p == 0 or n_l == 0:
            continue
        tpc = tp[sel].cumsum(0)
        fpc = (1 - tp[sel]).cumsum(0)
        recall = tpc / (n_l + eps)
        precision = tpc / (tpc + fpc)
        r_curve[ci] = _interp(-x, -conf[sel], recall[:, 0], left=0)
        p_curve[ci] = _interp(-x, -conf[sel], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j], mpre, mrec = compute_ap(recall[:, j], precision[:, j])
            if j == 0:
                prec_values.append(_interp(x, mrec, mpre))
    prec_values = np.array(prec_values)
    f1_curve = 2 * p_curve * r_curve / (p_curve + r_curve + eps)
    i = f1_head(f1_curve)
    p, r, f1 = p_curve[:, i], r_curve[:, i], f1_curve[:, i]
    tpn = (r * nt).round()
    fpn = (tpn / (p + eps) - tpn).round()
    return tpn, fpn, p, r, f1, ap, unique_classes.astype(int), p_curve, r_curve, f1_curve, x, prec_values


def f1_head(f1_curve, f=0.1):
    """Index of the maximum of the box-filtered mean F1 curve (utils/metrics.py:530-536 smooth + :701)."""
    y = f1_curve.mean(0)
    nf = round(len(y) * f * 2) // 2 + 1
    pad = np.ones(nf // 2)
    yp = np.concatenate((pad * y[0], y, pad * y[-1]), 0)
    return int(np.convolve(yp, np.ones(nf) / nf, mode="valid").argmax())
