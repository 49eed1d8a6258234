"""CPU restatement of the reference's predict-time preprocessing.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

  * ``letterbox_geometry`` / ``letterbox`` : data/augment.py:1535-1601 (LetterBox.__call__, labels=None path)
  * ``pre_transform`` / ``preprocess``      : engine/predictor.py:116-161
  * ``resize_linear_u8``                   : cv2.resize(..., INTER_LINEAR) for 8-bit images -- opencv-python is an
        un-vendored dependency (``opencv-python>=4.6.0``, unpinned, pyproject.toml:67; call site
        data/augment.py:1586).  Restates OpenCV's published fixed-point algorithm (modules/imgproc/src/resize.cpp):
        half-pixel centres, float source coordinate, 11-bit coefficients (saturate_cast<short>(c*2048), round half to
        even), horizontal pass S[sx]*a0 + S[sx+1]*a1 in int32, vertical pass
        ((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2; and the exact-2x-downscale shortcut to the 2x2 box mean
        (INTER_AREA fast path).  The reference's tests pin no pixel values -> PARITY UNPINNED for pixels;
        geometry (ratio, pad, output shape) is pinned against the reference class (tests/golden).
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _coeffs(dst: int, src: int):
    """Per-destination source index and the two int16 weights along one axis (OpenCV resize.cpp, linear)."""
    scale = float(src) / float(dst)  # double
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0.0
    s[lo] = 0
    hi = s >= src - 1
    f[hi] = 0.0
    s[hi] = src - 1
    c0 = np.float32(1.0) - f
    c1 = f
    # saturate_cast<short>(float) == cvRound -> round half to even
    a0 = np.rint(c0 * np.float32(COEF_SCALE)).astype(np.int32)
    a1 = np.rint(c1 * np.float32(COEF_SCALE)).astype(np.int32)
    return s, a0, a1


def _coeffs_y(dst: int, src: int):
    """Vertical axis: the row index is clamped at fetch time, the weights are NOT reset (resize.cpp resizeGeneric_)."""
    scale = float(src) / float(dst)
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    b0 = np.rint((np.float32(1.0) - f) * np.float32(COEF_SCALE)).astype(np.int32)
    b1 = np.rint(f * np.float32(COEF_SCALE)).astype(np.int32)
    r0 = np.clip(s, 0, src - 1)
    r1 = np.clip(s + 1, 0, src - 1)
    return r0, r1, b0, b1


def resize_linear_u8(img: np.ndarray, new_wh: Tuple[int, int]) -> np.ndarray:
    """img (h, w, c) uint8 -> (new_h, new_w, c) uint8."""
    h, w = img.shape[:2]
    nw, nh = int(new_wh[0]), int(new_wh[1])
    if (nw, nh) == (w, h):
        return img.copy()
    if w == 2 * nw and h == 2 * nh:  # INTER_LINEAR with exact 2x decimation -> INTER_AREA fast path
        s = img.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, a0, a1 = _coeffs(nw, w)
    r0, r1, b0, b1 = _coeffs_y(nh, h)
    sx1 = np.minimum(sx + 1, w - 1)
    src = img.astype(np.int32)
    hor = src[:, sx, :] * a0[None, :, None] + src[:, sx1, :] * a1[None, :, None]  # (h, nw, c) int32
    top = hor[r0]
    bot = hor[r1]
    out = (((b0[:, None, None] * (top >> 4)) >> 16) + ((b1[:, None, None] * (bot >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox_geometry(shape: Sequence[int], new_shape=(640, 640), auto=False, scale_fill=False, scaleup=True,
                       center=True, stride=32):
    """Returns (new_unpad (w,h), (top, bottom, left, right), ratio (rw, rh)).  data/augment.py:1556-1587."""
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    ratio = r, r
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = dw % stride, dh % stride
    elif scale_fill:
        dw, dh = 0.0, 0.0
        new_unpad = (new_shape[1], new_shape[0])
        ratio = new_shape[1] / shape[1], new_shape[0] / shape[0]
    if center:
        dw /= 2
        dh /= 2
    top, bottom = (int(round(dh - 0.1)) if center else 0), int(round(dh + 0.1))
    left, right = (int(round(dw - 0.1)) if center else 0), int(round(dw + 0.1))
    return new_unpad, (top, bottom, left, right), ratio


def letterbox(img: np.ndarray, new_shape=(640, 640), auto=False, scale_fill=False, scaleup=True, center=True,
              stride=32) -> np.ndarray:
    new_unpad, (top, bottom, left, right), _ = letterbox_geometry(img.shape[:2], new_shape, auto, scale_fill, scaleup,
                                                                 center, stride)
    if tuple(img.shape[:2][::-1]) != tuple(new_unpad):
        img = resize_linear_u8(img, new_unpad)
    h, w = img.shape[:2]
    out = np.full((h + top + bottom, w + left + right, img.shape[2]), 114, dtype=np.uint8)
    out[top:top + h, left:left + w] = img
    return out


def pre_transform(ims: List[np.ndarray], imgsz=(640, 640), pt=True, stride=32) -> List[np.ndarray]:
    """engine/predictor.py:145-161."""
    same_shapes = len({x.shape for x in ims}) == 1
    return [letterbox(x, imgsz, auto=same_shapes and pt, stride=stride) for x in ims]


def preprocess(ims: List[np.ndarray], imgsz=(640, 640), half=False, pt=True, stride=32):
    """engine/predictor.py:116-134 -> torch tensor (B,3,H,W) fp32 or fp16 in [0,1]."""
    import torch
    im = np.stack(pre_transform(ims, imgsz, pt, stride))
    im = im[..., ::-1].transpose((0, 3, 1, 2))
    im = torch.from_numpy(np.ascontiguousarray(im))
    im = im.half() if half else im.float()
    im /= 255
    return im
