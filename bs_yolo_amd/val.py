"""Validator matching step on the device: which detections are true positives at each IoU threshold.

Mirrors DetectionValidator._process_batch (models/yolo/detect/val.py:209-228) = utils/metrics.py:52-70 box_iou +
engine/validator.py:222-258 match_predictions, for a whole batch in one launch (csrc/val_match.hip).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import lib as L


def _iouv_host(iouv) -> "C.Array":
    vals = [float(v) for v in (iouv.detach().float().cpu().tolist() if isinstance(iouv, torch.Tensor) else iouv)]
    return (C.c_float * len(vals))(*vals), len(vals)


def match_batched(det: torch.Tensor, counts: torch.Tensor, gt_boxes: torch.Tensor, gt_cls: torch.Tensor,
                  gt_counts: torch.Tensor, iouv=None) -> torch.Tensor:
    """det (B, max_det, >=6) fp32 + counts (B) int32 as returned by nms.nms_batched; gt_boxes (B, Lmax, 4) xyxy in the same
    coordinate frame, gt_cls (B, Lmax), gt_counts (B) int32 -> (B, max_det, n_iou) bool (rows past counts[b] are False)."""
    if iouv is None:
        iouv = torch.linspace(0.5, 0.95, 10)  # detect/val.py:36
    B, max_det, row = det.shape
    assert det.is_cuda and det.dtype == torch.float32 and det.is_contiguous()
    Lmax = max(int(gt_boxes.shape[1]), 1)
    gb = torch.zeros((B, Lmax, 4), dtype=torch.float32, device=det.device)
    gc = torch.full((B, Lmax), -1.0, dtype=torch.float32, device=det.device)
    if gt_boxes.shape[1]:
        gb.copy_(gt_boxes.to(det.device, torch.float32))
        gc.copy_(gt_cls.to(det.device, torch.float32))
    counts = counts.to(det.device, torch.int32).contiguous()
    gt_counts = gt_counts.to(det.device, torch.int32).contiguous()
    arr, n = _iouv_host(iouv)
    out = torch.empty((B, max_det, n), dtype=torch.uint8, device=det.device)
    stream = torch.cuda.current_stream(det.device).cuda_stream
    L.check(L.lib.bsy_val_match(C.c_void_p(det.data_ptr()), row, C.c_void_p(counts.data_ptr()), B, max_det,
                                C.c_void_p(gb.data_ptr()), C.c_void_p(gc.data_ptr()), C.c_void_p(gt_counts.data_ptr()), Lmax,
                                arr, n, C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
    return out.bool()


def process_batch(detections: torch.Tensor, gt_bboxes: torch.Tensor, gt_cls: torch.Tensor, iouv=None) -> torch.Tensor:
    """Drop-in for DetectionValidator._process_batch (one image): detections (N, 6), gt_bboxes (M, 4), gt_cls (M,)
    -> (N, n_iou) bool on detections.device."""
    n = int(detections.shape[0])
    dev = detections.device
    niou = 10 if iouv is None else len(iouv)
    if n == 0 or gt_bboxes.shape[0] == 0:
        return torch.zeros((n, niou), dtype=torch.bool, device=dev)
    if n > 1024:
        raise ValueError("process_batch: more than 1024 detections per image")
    det = detections[:, :6].float().contiguous().unsqueeze(0)
    one = torch.tensor([n], dtype=torch.int32, device=dev)
    m = torch.tensor([int(gt_bboxes.shape[0])], dtype=torch.int32, device=dev)
    return match_batched(det, one, gt_bboxes.unsqueeze(0), gt_cls.reshape(1, -1), m, iouv)[0]
