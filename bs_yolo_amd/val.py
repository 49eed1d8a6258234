"""Validator metric path on the device.

  * match_batched / process_batch -- which detections are true positives at each IoU threshold: mirrors
    DetectionValidator._process_batch (models/yolo/detect/val.py:209-228) = utils/metrics.py:52-70 box_iou +
    engine/validator.py:222-258 match_predictions, for a whole batch in one launch (csrc/val_match.hip).
  * ap_per_class -- utils/metrics.py:620-706 (+ compute_ap :588-617): ranking, cumulative TP / FP, the 1000-point P / R
    curves and the 101-point interpolated AP per class and threshold on the device in float64 (csrc/val_ap.hip); only the
    max-F1 operating point (a box filter over a (nc, 1000) array) is taken on the host, with numpy as the reference does.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
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


def ap_per_class(tp, conf, pred_cls, target_cls, plot=False, on_plot=None, save_dir=None, names=None, eps=1e-16, prefix="",
                 device="cuda:0"):
    """Reference signature (utils/metrics.py:620-622; plotting arguments are accepted and ignored: the plots stay with
    the reference) -> the same 12-tuple of numpy arrays.  tp (N, T) bool, conf (N), pred_cls (N), target_cls (M): numpy
    arrays or tensors, host or device."""
    if plot:
        raise NotImplementedError("plot=True: call the reference's plot_pr_curve / plot_mc_curve on the returned curves")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("bs_yolo_amd.val.ap_per_class needs a GPU (no CPU fallback)")
    tcls = (target_cls.detach().cpu().numpy() if torch.is_tensor(target_cls) else np.asarray(target_cls))
    unique_classes, nt = np.unique(tcls, return_counts=True)
    nc = int(unique_classes.shape[0])
    t_tp = torch.as_tensor(tp).to(dev)
    T = int(t_tp.shape[1]) if t_tp.dim() == 2 else 0
    N = int(t_tp.shape[0])
    x = np.linspace(0, 1, 1000)
    ap, p_curve, r_curve = np.zeros((nc, T)), np.zeros((nc, 1000)), np.zeros((nc, 1000))
    prec_values = np.zeros((0, 1000))
    if N and nc:
        if unique_classes.min() < 0 or unique_classes.max() >= 4095 or np.any(unique_classes != np.floor(unique_classes)):
            raise ValueError("class ids must be integers in [0, 4095)")
        t_tp = t_tp.to(torch.uint8).contiguous()
        t_conf = torch.as_tensor(conf).to(dev, torch.float32).contiguous()
        t_cls = torch.as_tensor(pred_cls).to(dev, torch.float32).contiguous()
        if float(t_conf.min()) < 0:
            raise ValueError("conf must be non-negative")
        d_cls = torch.as_tensor(unique_classes.astype(np.int32)).to(dev)
        d_nt = torch.as_tensor(nt.astype(np.int32)).to(dev)
        d_x101, d_x1000 = torch.as_tensor(np.linspace(0, 1, 101)).to(dev), torch.as_tensor(x).to(dev)
        d_ap = torch.empty((nc, T), dtype=torch.float64, device=dev)
        d_p, d_r, d_pv = (torch.empty((nc, 1000), dtype=torch.float64, device=dev) for _ in range(3))
        d_np = torch.zeros((nc,), dtype=torch.int32, device=dev)
        ws = torch.empty(L.lib.bsy_ap_workspace_bytes(N, T), dtype=torch.uint8, device=dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(L.lib.bsy_ap_per_class(C.c_void_p(t_tp.data_ptr()), C.c_void_p(t_conf.data_ptr()), C.c_void_p(t_cls.data_ptr()), N, T,
                                       C.c_void_p(d_cls.data_ptr()), C.c_void_p(d_nt.data_ptr()), nc, C.c_void_p(d_x101.data_ptr()),
                                       C.c_void_p(d_x1000.data_ptr()), float(eps), C.c_void_p(d_ap.data_ptr()),
                                       C.c_void_p(d_p.data_ptr()), C.c_void_p(d_r.data_ptr()), C.c_void_p(d_pv.data_ptr()),
                                       C.c_void_p(d_np.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), stream))
        ap, p_curve, r_curve = d_ap.cpu().numpy(), d_p.cpu().numpy(), d_r.cpu().numpy()
        has = (d_np.cpu().numpy() > 0) & (nt > 0)
        prec_values = d_pv.cpu().numpy()[has]  # the reference appends a row only for classes with predictions (:682)
    # utils/metrics.py:687-706: F1, its box-filtered maximum, the operating point's P / R / F1 / TP / FP
    f1_curve = 2 * p_curve * r_curve / (p_curve + r_curve + eps)
    y = f1_curve.mean(0) if nc else np.zeros(1000)
    nf = round(len(y) * 0.1 * 2) // 2 + 1
    pad = np.ones(nf // 2)
    i = int(np.convolve(np.concatenate((pad * y[0], y, pad * y[-1]), 0), np.ones(nf) / nf, mode="valid").argmax())
    p, r, f1 = p_curve[:, i], r_curve[:, i], f1_curve[:, i]
    tpn = (r * nt).round()
    fpn = (tpn / (p + eps) - tpn).round()
    return tpn, fpn, p, r, f1, ap, unique_classes.astype(int), p_curve, r_curve, f1_curve, x, prec_values
