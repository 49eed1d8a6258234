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
