"""Host-side mirror of ``ultralytics.utils.ops.non_max_suppression`` (utils/ops.py:167-316) over the HIP kernels.

Same name, argument meaning, return value and error behaviour as the reference function so that
``ultralytics.utils.ops.non_max_suppression = bs_yolo_amd.nms.non_max_suppression`` is a drop-in
(call sites: models/yolo/detect/predict.py:25, detect/val.py:95, segment/predict.py:30, segment/val.py:73).
Differences, by design: one batched device call with no host sync until the per-image slicing; no wall-clock
bail-out (ops.py:238,312-314); NMS arithmetic is fp32 even for fp16 predictions.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch

from . import lib as L


def _workspace(device, nbytes: int) -> torch.Tensor:
    """Scratch for one call, taken from torch's caching allocator: the block is bound to the CURRENT stream (a later
    call on another stream or thread never shares it while this call's kernels are pending), so the boundary keeps no
    state of its own -- SURVEY section 8(b): "no global state except last-error TLS"."""
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def nms_batched(prediction: torch.Tensor, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False,
                multi_label=False, max_det=300, nc=0, max_nms=30000, max_wh=7680, in_place=True):
    """Device-resident result: (det (B, max_det, 6+nm) fp32, counts (B,) int32).  No host synchronisation."""
    if not prediction.is_cuda:
        raise RuntimeError("bs_yolo_amd.nms needs a GPU tensor (no CPU fallback)")
    if prediction.dim() != 3:
        raise ValueError("prediction must be (B, 4+nc+nm, A)")
    if not prediction.is_contiguous():
        if in_place:
            raise ValueError("in_place NMS needs a contiguous (B, C, A) prediction")
        prediction = prediction.contiguous()
    B, Cc, A = prediction.shape
    nc = nc or (Cc - 4)
    nm = Cc - nc - 4
    dev = prediction.device
    cls_t = None
    if classes is not None:
        cls_t = torch.as_tensor(list(classes), dtype=torch.int32, device=dev)
    ml = bool(multi_label) and nc > 1
    nbytes = L.lib.bsy_nms_workspace_bytes(B, A, nc, int(ml), int(max_nms))
    ws = _workspace(dev, nbytes)
    det = torch.empty((B, max_det, 6 + nm), dtype=torch.float32, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    L.check(L.lib.bsy_nms(C.c_void_p(prediction.data_ptr()), L.dtype_code(prediction.dtype), B, nc, nm, A,
                          float(conf_thres), float(iou_thres),
                          C.c_void_p(cls_t.data_ptr()) if cls_t is not None else None,
                          int(cls_t.numel()) if cls_t is not None else 0, int(bool(agnostic)), int(ml), int(max_det),
                          int(max_nms), float(max_wh), int(bool(in_place)), C.c_void_p(det.data_ptr()),
                          C.c_void_p(counts.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), stream))
    return det, counts


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        labels=(), max_det=300, nc=0, max_time_img=0.05, max_nms=30000, max_wh=7680, in_place=True,
                        rotated=False) -> List[torch.Tensor]:
    """Reference signature (utils/ops.py:167-182) -> list of (n_i, 6+nm) tensors in the prediction's dtype."""
    assert 0 <= conf_thres <= 1, f"Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0"
    assert 0 <= iou_thres <= 1, f"Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0"
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    if rotated or (labels is not None and len(labels)) or prediction.shape[-1] == 6:
        raise NotImplementedError("rotated / autolabel / end2end NMS stay on the reference implementation")
    det, counts = nms_batched(prediction, conf_thres, iou_thres, classes, agnostic, multi_label, max_det, nc, max_nms,
                              max_wh, in_place)
    n = counts.tolist()  # the only host sync
    # the reference yields rows in the prediction's dtype (ops.py:310 x[i]); empty images are fp32 zeros((0, 6+nm))
    out = []
    for b, k in enumerate(n):
        out.append(det[b, :k].to(prediction.dtype) if k else torch.zeros((0, det.shape[-1]), device=prediction.device))
    return out


def scale_boxes_batched(det: torch.Tensor, counts: torch.Tensor, img1_shape, img0_shapes, ratio_pads=None):
    """scale_boxes + clip_boxes (utils/ops.py:92-127, :319-337) for a whole batch, in place on `det` (B, max_det, row).
    The gain / pad per image are computed on the host exactly as the reference does (Python round())."""
    B, max_det, row = det.shape
    gains, pxs, pys, h0s, w0s = [], [], [], [], []
    for b in range(B):
        h0, w0 = img0_shapes[b][:2]
        if ratio_pads is None or ratio_pads[b] is None:
            gain = min(img1_shape[0] / h0, img1_shape[1] / w0)
            pad = (round((img1_shape[1] - w0 * gain) / 2 - 0.1), round((img1_shape[0] - h0 * gain) / 2 - 0.1))
        else:
            gain, pad = ratio_pads[b][0][0], ratio_pads[b][1]
        gains.append(gain); pxs.append(pad[0]); pys.append(pad[1]); h0s.append(h0); w0s.append(w0)
    dev = det.device
    t = torch.tensor([gains, pxs, pys, h0s, w0s], dtype=torch.float32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    L.check(L.lib.bsy_scale_boxes(C.c_void_p(det.data_ptr()), C.c_void_p(counts.data_ptr()), B, max_det, row,
                                  C.c_void_p(t[0].data_ptr()), C.c_void_p(t[1].data_ptr()), C.c_void_p(t[2].data_ptr()),
                                  C.c_void_p(t[3].data_ptr()), C.c_void_p(t[4].data_ptr()), stream))
    return det
