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
