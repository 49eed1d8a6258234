"""Host-side mirror of ``ultralytics.utils.ops.process_mask`` (utils/ops.py:663-694) over the HIP mask kernels.

Same signature and return convention as the reference (a float tensor of 0/1, shape (n, h, w)), so
``ultralytics.utils.ops.process_mask = bs_yolo_amd.masks.process_mask`` is a drop-in for segment/predict.py:53 and
segment/val.py (non-native path).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as L


def process_mask(protos: torch.Tensor, masks_in: torch.Tensor, bboxes: torch.Tensor, shape, upsample: bool = False,
                 out_dtype=torch.float32) -> torch.Tensor:
    if not protos.is_cuda:
        raise RuntimeError("bs_yolo_amd.masks needs GPU tensors (no CPU fallback)")
    c, mh, mw = protos.shape
    ih, iw = int(shape[0]), int(shape[1])
    n = int(masks_in.shape[0])
    dev = protos.device
    oh, ow = (ih, iw) if upsample else (mh, mw)
    out = torch.empty((n, oh, ow), dtype=out_dtype, device=dev)
    if n == 0:
        return out
    protos = protos.contiguous()
    coef = masks_in.to(torch.float32).contiguous()
    boxes = bboxes.to(torch.float32).contiguous()
    low = torch.empty((n, mh, mw), dtype=torch.float32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    L.check(L.lib.bsy_process_mask(C.c_void_p(protos.data_ptr()), L.dtype_code(protos.dtype), c, mh, mw,
                                   C.c_void_p(coef.data_ptr()), coef.shape[1], C.c_void_p(boxes.data_ptr()),
                                   boxes.shape[1], n, ih, iw, int(bool(upsample)), C.c_void_p(low.data_ptr()),
                                   C.c_void_p(out.data_ptr()), L.dtype_code(out_dtype), stream))
    return out


def _window(mh: int, mw: int, shape, padding: bool = True):
    """scale_masks' crop of the letterbox padding, in the reference's own Python arithmetic (utils/ops.py:723-732)."""
    gain = min(mh / shape[0], mw / shape[1])  # gain = old / new
    pad = [mw - shape[1] * gain, mh - shape[0] * gain]  # wh padding
    if padding:
        pad[0] /= 2
        pad[1] /= 2
    top, left = (int(pad[1]), int(pad[0])) if padding else (0, 0)
    bottom, right = int(mh - pad[1]), int(mw - pad[0])
    return top, left, bottom, right


def scale_masks(masks: torch.Tensor, shape, padding: bool = True) -> torch.Tensor:
    """Drop-in for ``ultralytics.utils.ops.scale_masks`` (utils/ops.py:712-737): masks (N, C, H, W) fp16 / fp32 on the GPU ->
    (N, C, shape[0], shape[1]), bilinear, align_corners=False, after cutting the letterbox padding away."""
    if not masks.is_cuda:
        raise RuntimeError("bs_yolo_amd.masks needs GPU tensors (no CPU fallback)")
    N, Cc, mh, mw = masks.shape
    oh, ow = int(shape[0]), int(shape[1])
    top, left, bottom, right = _window(mh, mw, (oh, ow), padding)
    out = torch.empty((N, Cc, oh, ow), dtype=masks.dtype, device=masks.device)
    if N * Cc == 0:
        return out
    m = masks.contiguous()
    stream = C.c_void_p(torch.cuda.current_stream(masks.device).cuda_stream)
    L.check(L.lib.bsy_scale_masks(C.c_void_p(m.data_ptr()), L.dtype_code(m.dtype), N * Cc, mh, mw, top, left, bottom, right, oh, ow,
                                  C.c_void_p(out.data_ptr()), stream))
    return out


def process_mask_native(protos: torch.Tensor, masks_in: torch.Tensor, bboxes: torch.Tensor, shape, out_dtype=torch.float32) -> torch.Tensor:
    """Drop-in for ``ultralytics.utils.ops.process_mask_native`` (utils/ops.py:696-709; segment/predict.py:48-50 with
    retina_masks): masks at the ORIGINAL image size, cropped to boxes given in original-image pixels -> (n, h, w) of 0 / 1."""
    if not protos.is_cuda:
        raise RuntimeError("bs_yolo_amd.masks needs GPU tensors (no CPU fallback)")
    c, mh, mw = protos.shape
    oh, ow = int(shape[0]), int(shape[1])
    n = int(masks_in.shape[0])
    dev = protos.device
    out = torch.empty((n, oh, ow), dtype=out_dtype, device=dev)
    if n == 0:
        return out
    top, left, bottom, right = _window(mh, mw, (oh, ow), True)
    protos = protos.contiguous()
    coef = masks_in.to(torch.float32).contiguous()
    boxes = bboxes.to(torch.float32).contiguous()
    low = torch.empty((n, mh, mw), dtype=torch.float32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    L.check(L.lib.bsy_process_mask_native(C.c_void_p(protos.data_ptr()), L.dtype_code(protos.dtype), c, mh, mw,
                                          C.c_void_p(coef.data_ptr()), coef.shape[1], C.c_void_p(boxes.data_ptr()), boxes.shape[1], n,
                                          top, left, bottom, right, oh, ow, C.c_void_p(low.data_ptr()), C.c_void_p(out.data_ptr()),
                                          L.dtype_code(out_dtype), stream))
    return out
