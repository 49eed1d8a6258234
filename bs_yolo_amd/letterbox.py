"""Host-side mirror of LetterBox + BasePredictor.pre_transform/preprocess over the HIP letterbox kernel.

  * ``LetterBox``   -- same constructor/attributes as data/augment.py:1477-1533; ``geometry()`` restates the
                       arithmetic of ``__call__`` (:1556-1587) that must run on the host (Python ``round``).
  * ``preprocess``  -- engine/predictor.py:116-161 for a list of HWC BGR uint8 images -> (B,3,H,W) device tensor.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np
import torch

from . import lib as L


class LetterBox:
    def __init__(self, new_shape=(640, 640), auto=False, scaleFill=False, scaleup=True, center=True, stride=32):
        self.new_shape = new_shape
        self.auto = auto
        self.scaleFill = scaleFill
        self.scaleup = scaleup
        self.stride = stride
        self.center = center

    def geometry(self, shape: Sequence[int]):
        """shape (h, w) -> (out_h, out_w, new_unpad_w, new_unpad_h, left, top, ratio)."""
        new_shape = self.new_shape
        if isinstance(new_shape, int):
            new_shape = (new_shape, new_shape)
        r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
        if not self.scaleup:
            r = min(r, 1.0)
        ratio = r, r
        new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
        dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
        if self.auto:
            dw, dh = dw % int(self.stride), dh % int(self.stride)
        elif self.scaleFill:
            dw, dh = 0.0, 0.0
            new_unpad = (new_shape[1], new_shape[0])
            ratio = new_shape[1] / shape[1], new_shape[0] / shape[0]
        if self.center:
            dw /= 2
            dh /= 2
        top, bottom = (int(round(dh - 0.1)) if self.center else 0), int(round(dh + 0.1))
        left, right = (int(round(dw - 0.1)) if self.center else 0), int(round(dw + 0.1))
        return (new_unpad[1] + top + bottom, new_unpad[0] + left + right, new_unpad[0], new_unpad[1], left, top, ratio)


def preprocess(ims: List, imgsz=(640, 640), half=True, pt=True, stride=32, device="cuda:0") -> torch.Tensor:
    """ims: list of (h,w,3) BGR uint8 numpy arrays or uint8 tensors (host or device)."""
    dev = torch.device(device)
    same_shapes = len({tuple(x.shape) for x in ims}) == 1
    lb = LetterBox(imgsz, auto=same_shapes and pt, stride=stride)
    geoms = [lb.geometry(tuple(x.shape[:2])) for x in ims]
    out_hw = {(g[0], g[1]) for g in geoms}
    if len(out_hw) != 1:
        raise ValueError(f"letterboxed images differ in shape: {out_hw}")  # np.stack would raise in the reference
    H2, W2 = out_hw.pop()
    dimgs = []
    for x in ims:
        t = torch.from_numpy(np.ascontiguousarray(x)) if isinstance(x, np.ndarray) else x.contiguous()
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
            raise TypeError("images must be (h, w, 3) uint8")
        dimgs.append(t.to(dev, non_blocking=True))
    B = len(ims)
    ptrs = torch.tensor([t.data_ptr() for t in dimgs], dtype=torch.int64).to(dev)
    hw = torch.tensor([[t.shape[0], t.shape[1]] for t in dimgs], dtype=torch.int32).to(dev)
    geom = torch.tensor([[g[2], g[3], g[4], g[5]] for g in geoms], dtype=torch.int32).to(dev)
    out = torch.empty((B, 3, H2, W2), dtype=torch.float16 if half else torch.float32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    L.check(L.lib.bsy_letterbox(C.c_void_p(ptrs.data_ptr()), C.c_void_p(hw.data_ptr()), C.c_void_p(geom.data_ptr()), B,
                                H2, W2, C.c_void_p(out.data_ptr()), L.dtype_code(out.dtype), stream))
    # keep the sources alive until the kernel has consumed them
    out._bsy_keepalive = (dimgs, ptrs, hw, geom)
    return out
