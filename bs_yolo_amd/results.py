"""The data containers a predict loop hands back: host-side mirror of `Results` / `Boxes` (engine/results.py:187-330, :939-1155).

Only the DATA side -- what downstream code reads (`r.boxes.xyxy`, `.conf`, `.cls`, `.xywh`, `.xyxyn`, `.xywhn`, `r.orig_shape`, `r.path`,
`r.names`, `len(r)`, indexing, `.cpu()` / `.numpy()`); plotting, saving and export stay with the reference (UI, out of scope).  The
detections come from `bsy_nms` + `bsy_scale_boxes` already in original-image pixels (models/yolo/detect/predict.py:20-45); these
classes add views and the same few lines of coordinate arithmetic the reference's properties run (ops.py xyxy2xywh).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch


class Boxes:
    """engine/results.py:939-1155.  `data` (n, 6) [x1, y1, x2, y2, conf, cls] (or (n, 7) with a track id before conf) in
    original-image pixels, torch tensor or numpy array; `orig_shape` = (height, width)."""

    def __init__(self, boxes, orig_shape: Tuple[int, int]):
        if boxes.ndim == 1:
            boxes = boxes[None, :]
        n = boxes.shape[-1]
        assert n in (6, 7), f"expected 6 or 7 values but got {n}"
        self.data, self.orig_shape, self.is_track = boxes, tuple(orig_shape), n == 7

    # ---- BaseTensor (results.py:22-185) ----
    @property
    def shape(self):
        return self.data.shape

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        return self.__class__(self.data[idx], self.orig_shape)

    def cpu(self):
        return self if isinstance(self.data, np.ndarray) else self.__class__(self.data.cpu(), self.orig_shape)

    def numpy(self):
        return self if isinstance(self.data, np.ndarray) else self.__class__(self.data.cpu().numpy(), self.orig_shape)

    def cuda(self):
        return self.__class__(torch.as_tensor(self.data).cuda(), self.orig_shape)

    def to(self, *args, **kwargs):
        return self.__class__(torch.as_tensor(self.data).to(*args, **kwargs), self.orig_shape)

    # ---- views ----
    @property
    def xyxy(self):
        return self.data[:, :4]

    @property
    def conf(self):
        return self.data[:, -2]

    @property
    def cls(self):
        return self.data[:, -1]

    @property
    def id(self):
        return self.data[:, -3] if self.is_track else None

    @staticmethod
    def _xyxy2xywh(x):  # ops.py xyxy2xywh: centre, then extents
        y = torch.empty_like(x) if isinstance(x, torch.Tensor) else np.empty_like(x)
        y[..., 0] = (x[..., 0] + x[..., 2]) / 2
        y[..., 1] = (x[..., 1] + x[..., 3]) / 2
        y[..., 2] = x[..., 2] - x[..., 0]
        y[..., 3] = x[..., 3] - x[..., 1]
        return y

    @property
    def xywh(self):
        return self._xyxy2xywh(self.xyxy)

    @property
    def xyxyn(self):
        xyxy = self.xyxy.clone() if isinstance(self.xyxy, torch.Tensor) else np.copy(self.xyxy)
        xyxy[..., [0, 2]] /= self.orig_shape[1]
        xyxy[..., [1, 3]] /= self.orig_shape[0]
        return xyxy

    @property
    def xywhn(self):
        xywh = self._xyxy2xywh(self.xyxy)
        xywh[..., [0, 2]] /= self.orig_shape[1]
        xywh[..., [1, 3]] /= self.orig_shape[0]
        return xywh


class Results:
    """engine/results.py:187-330, data side: one image's detections."""

    def __init__(self, orig_img, path: Optional[str], names: Dict[int, str], boxes=None, masks=None, speed: Optional[dict] = None,
                 orig_shape: Optional[Tuple[int, int]] = None):
        # orig_img may be None (the pinned loader keeps the decoded images only on request, keep_im0): orig_shape then says (h, w)
        self.orig_img = orig_img
        self.orig_shape = tuple(orig_img.shape[:2]) if orig_img is not None else tuple(orig_shape)
        self.boxes = Boxes(boxes, self.orig_shape) if boxes is not None else None
        self.masks = masks  # (n, H, W) tensor as `process_mask` returns it, or None
        self.probs = self.keypoints = self.obb = None
        self.speed = speed or {"preprocess": None, "inference": None, "postprocess": None}
        self.names, self.path = names, path

    def __len__(self):
        for k in ("boxes", "masks"):
            v = getattr(self, k)
            if v is not None:
                return len(v)
        return 0

    def __getitem__(self, idx):
        r = Results(self.orig_img, self.path, self.names, speed=self.speed, orig_shape=self.orig_shape)
        r.boxes = self.boxes[idx] if self.boxes is not None else None
        r.masks = self.masks[idx] if self.masks is not None else None
        return r

    def _apply(self, fn: str):
        r = Results(self.orig_img, self.path, self.names, speed=self.speed, orig_shape=self.orig_shape)
        r.boxes = getattr(self.boxes, fn)() if self.boxes is not None else None
        if self.masks is not None:
            r.masks = self.masks.cpu() if fn == "cpu" else (self.masks.cpu().numpy() if fn == "numpy" else self.masks.cuda())
        return r

    def cpu(self):
        return self._apply("cpu")

    def numpy(self):
        return self._apply("numpy")

    def cuda(self):
        return self._apply("cuda")

    def summary(self, normalize: bool = False, decimals: int = 5) -> List[dict]:
        """results.py:757-820 for detections: one dict per box."""
        out = []
        if self.boxes is None:
            return out
        d = self.boxes.numpy()
        h, w = self.orig_shape if normalize else (1, 1)
        for i in range(len(d)):
            x1, y1, x2, y2 = (float(v) for v in d.xyxy[i])
            c = int(d.cls[i])
            out.append({"name": self.names[c], "class": c, "confidence": round(float(d.conf[i]), decimals),
                        "box": {"x1": round(x1 / w, decimals), "y1": round(y1 / h, decimals), "x2": round(x2 / w, decimals), "y2": round(y2 / h, decimals)}})
        return out


def build_results(det: torch.Tensor, counts: torch.Tensor, orig_imgs: Optional[Sequence], paths: Sequence[Optional[str]], names: Dict[int, str],
                  orig_shapes: Optional[Sequence[Tuple[int, int]]] = None) -> List[Results]:
    """models/yolo/detect/predict.py:34-45 (construct_results) on the batched NMS output: det (B, max_det, 6) with boxes already
    scaled to each original image, counts (B,) -> one `Results` per image (row views, no copies; ONE host read of `counts`).
    orig_imgs may be None when orig_shapes is given."""
    n = counts.tolist()
    imgs = orig_imgs if orig_imgs is not None else [None] * len(paths)
    shapes = orig_shapes if orig_shapes is not None else [None] * len(paths)
    return [Results(img, p, names, boxes=det[i, :n[i], :6], orig_shape=sh) for i, (img, p, sh) in enumerate(zip(imgs, paths, shapes))]
