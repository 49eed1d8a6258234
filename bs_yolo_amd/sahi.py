"""Sliced inference on the device: host-side mirror of the `sahi` calls the reference makes.

The reference reaches sliced inference through the un-vendored `sahi` package -- `detect-sahi.py:1-13`
(``sahi.predict.predict(slice_height=800, slice_width=800, overlap_*_ratio=0)``) and
`examples/YOLOv8-SAHI-Inference-Video/yolov8_sahi.py:70-75` (``get_sliced_prediction(frame, model, 512, 512)``).
Names and argument meaning below follow sahi 0.11.x so those two scripts read the same:

  * ``get_slice_bboxes``       sahi.slicing.get_slice_bboxes (host integer arithmetic, explicit slice size)
  * ``slice_image``            sahi.slicing.slice_image + the predictor's preprocess of every crop -> one HIP kernel
  * ``postprocess``            sahi.postprocess.combine.GreedyNMMPostprocess / NMMPostprocess / NMSPostprocess (LSNMS runs as NMS)
                               -> four HIP kernels
  * ``get_sliced_prediction``  sahi.predict.get_sliced_prediction: slice -> forward + NMS per tile batch ->
                               (optional full-image prediction) -> cross-tile merge.  With torch.distributed
                               initialised the tiles are sharded over the ranks (parallel.shard_bounds), the fixed-size
                               per-tile detections are all-gathered (RCCL) and every rank merges the same list.

There is no CPU fallback: everything here needs the HIP library and a GPU tensor.  The algorithm these kernels follow
is restated, with its sources, in oracle/sahi_ref.py (parity unpinned -- the reference holds nothing that fixes it).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as L
from .nms import _workspace, nms_batched, scale_boxes_batched
from .parallel import gather_detections, shard_bounds

METRICS = {"IOU": 0, "IOS": 1}
# sahi.predict POSTPROCESS_NAME_TO_CLASS -> bsy_sahi_merge's do_merge.  LSNMS (the `lsnms` package: an R-tree accelerated NMS,
# IOU only) keeps what NMS keeps, so it runs the NMS kernels.
_POSTPROCESS = {"NMS": 0, "LSNMS": 0, "GREEDYNMM": 1, "NMM": 2}


def get_slice_bboxes(image_height: int, image_width: int, slice_height: int, slice_width: int,
                     overlap_height_ratio: float = 0.2, overlap_width_ratio: float = 0.2) -> List[List[int]]:
    """[[x0, y0, x1, y1], ...] row-major.  Border slices are moved back inside the image (so they keep the full slice
    size and overlap their neighbours); 6000 x 4000 with 640 x 640 / overlap 0 -> 10 x 7 = 70 slices."""
    if slice_height <= 0 or slice_width <= 0:
        raise ValueError("slice size must be positive")
    y_overlap = int(overlap_height_ratio * slice_height)
    x_overlap = int(overlap_width_ratio * slice_width)
    if y_overlap >= slice_height or x_overlap >= slice_width:
        raise ValueError("overlap ratio must be < 1")
    boxes = []
    y_max = y_min = 0
    while y_max < image_height:
        x_min = x_max = 0
        y_max = y_min + slice_height
        while x_max < image_width:
            x_max = x_min + slice_width
            if y_max > image_height or x_max > image_width:
                xe, ye = min(image_width, x_max), min(image_height, y_max)
                boxes.append([max(0, xe - slice_width), max(0, ye - slice_height), xe, ye])
            else:
                boxes.append([x_min, y_min, x_max, y_max])
            x_min = x_max - x_overlap
        y_min = y_max - y_overlap
    return boxes


def _as_device_image(image, device) -> torch.Tensor:
    t = torch.from_numpy(np.ascontiguousarray(image)) if isinstance(image, np.ndarray) else image
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
        raise TypeError("image must be (h, w, 3) uint8")
    return t.to(device, non_blocking=True).contiguous()


def slice_image(image, bboxes: Sequence[Sequence[int]], half: bool = True, swap_rb: bool = True,
                device="cuda:0") -> torch.Tensor:
    """(H, W, 3) u8 image + slice boxes of one common size -> (T, 3, h, w) device tensor in [0, 1]."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("bs_yolo_amd.sahi needs a GPU (no CPU fallback)")
    img = _as_device_image(image, dev)
    H, W = int(img.shape[0]), int(img.shape[1])
    sizes = {(b[2] - b[0], b[3] - b[1]) for b in bboxes}
    if len(sizes) != 1:
        raise ValueError(f"slices differ in size ({sorted(sizes)}): the image is smaller than the slice -- use the "
                         "letterbox path for it")
    tw, th = sizes.pop()
    for x0, y0, x1, y1 in bboxes:
        if x0 < 0 or y0 < 0 or x1 > W or y1 > H:
            raise ValueError(f"slice {[x0, y0, x1, y1]} leaves the {W} x {H} image")
    boxes = torch.tensor(np.asarray(bboxes, dtype=np.int32)).to(dev)
    out = torch.empty((len(bboxes), 3, th, tw), dtype=torch.float16 if half else torch.float32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    L.check(L.lib.bsy_slice_tiles(C.c_void_p(img.data_ptr()), H, W, 3 * W, C.c_void_p(boxes.data_ptr()), len(bboxes), th,
                                  tw, int(bool(swap_rb)), C.c_void_p(out.data_ptr()), L.dtype_code(out.dtype), stream))
    out._bsy_keepalive = (img, boxes)
    return out


def postprocess(det: torch.Tensor, counts: torch.Tensor, shifts, postprocess_type: str = "GREEDYNMM",
                match_metric: str = "IOS", match_threshold: float = 0.5, class_agnostic: bool = False,
                full_shape: Optional[Tuple[int, int]] = None, max_out: Optional[int] = None):
    """det (T, max_det, row >= 6) fp32 + counts (T,) int32 as nms_batched returns them (tile pixels), shifts (T, 2) =
    tile origins (x0, y0) -> (out (max_out, 6) fp32, out_count () int32), both on the device, no host sync."""
    if not det.is_cuda:
        raise RuntimeError("bs_yolo_amd.sahi needs GPU tensors (no CPU fallback)")
    if postprocess_type not in _POSTPROCESS:
        raise ValueError(f"postprocess_type {postprocess_type!r}: one of {sorted(_POSTPROCESS)}")
    if postprocess_type == "LSNMS" and match_metric != "IOU":
        raise NotImplementedError("LSNMS: IOU only (as in sahi)")
    if match_metric not in METRICS:
        raise ValueError(f"match_metric {match_metric!r}")
    det = det.contiguous().float()
    counts = counts.contiguous().to(torch.int32)
    T, max_det, row = det.shape
    dev = det.device
    sh = torch.as_tensor(np.asarray(shifts, dtype=np.float32).reshape(T, 2)).to(dev) if not torch.is_tensor(shifts) \
        else shifts.to(dev, torch.float32).contiguous()
    max_out = int(max_out or T * max_det)
    out = torch.zeros((max_out, 6), dtype=torch.float32, device=dev)
    n = torch.zeros((), dtype=torch.int32, device=dev)
    ws = _workspace(dev, L.lib.bsy_sahi_merge_workspace_bytes(T, max_det))
    fh, fw = (float(full_shape[0]), float(full_shape[1])) if full_shape is not None else (0.0, 0.0)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    L.check(L.lib.bsy_sahi_merge(C.c_void_p(det.data_ptr()), C.c_void_p(counts.data_ptr()), C.c_void_p(sh.data_ptr()), T,
                                 max_det, row, METRICS[match_metric], float(match_threshold), int(bool(class_agnostic)),
                                 _POSTPROCESS[postprocess_type], fw, fh, C.c_void_p(out.data_ptr()),
                                 C.c_void_p(n.data_ptr()), max_out, C.c_void_p(ws.data_ptr()), ws.numel(), stream))
    return out, n


def get_sliced_prediction(image, detection_model, slice_height: int = 640, slice_width: int = 640,
                          overlap_height_ratio: float = 0.2, overlap_width_ratio: float = 0.2,
                          perform_standard_pred: bool = True, postprocess_type: str = "GREEDYNMM",
                          postprocess_match_metric: str = "IOS", postprocess_match_threshold: float = 0.5,
                          postprocess_class_agnostic: bool = False, conf: float = 0.25, iou: float = 0.7,
                          max_det: int = 300, half: bool = True, batch: Optional[int] = None, bgr: bool = True,
                          imgsz: Optional[int] = None, group=None):
    """`image`: (H, W, 3) u8 (BGR as cv2 reads it when bgr=True);  `detection_model`: a YoloEngine (or any callable
    tiles (B, 3, h, w) -> (pred (B, 4+nc, A), ...)).  Returns (det (K, 6) fp32 [x1 y1 x2 y2 score cls] in image
    pixels, tile boxes).  `conf` is sahi's model_confidence_threshold; iou / max_det are the per-tile NMS settings of
    the predictor (cfg/default.yaml).  `imgsz`: letterbox every slice to this model input size (the reference's flow: sahi hands
    each slice to the ultralytics predictor); None = feed the slices natively when their size is a multiple of 32, else imgsz 640.
    Slices larger than the image shrink to it (get_slice_bboxes)."""
    dev = detection_model.device if hasattr(detection_model, "device") else torch.device("cuda:0")
    img = _as_device_image(image, dev)
    H, W = int(img.shape[0]), int(img.shape[1])
    bboxes = get_slice_bboxes(H, W, slice_height, slice_width, overlap_height_ratio, overlap_width_ratio)
    T = len(bboxes)
    world = torch.distributed.get_world_size(group) if torch.distributed.is_available() and \
        torch.distributed.is_initialized() else 1
    rank = torch.distributed.get_rank(group) if world > 1 else 0
    s, e = shard_bounds(T, world)[rank]
    dets, cnts = [], []
    step = batch or max(e - s, 1)
    th, tw = bboxes[0][3] - bboxes[0][1], bboxes[0][2] - bboxes[0][0]
    # Native tiles (the slice IS the model input: one kernel cuts all of them) when no imgsz is asked for and the slice size suits the
    # model's stride; otherwise every slice goes through the predictor's letterbox to `imgsz` and its boxes back through scale_boxes
    # -- what sahi's ultralytics wrapper does with EVERY slice (detect-sahi.py's 800 x 800 slices reach the model at imgsz 640)
    native = imgsz is None and th % 32 == 0 and tw % 32 == 0
    for i in range(s, e, step):
        chunk = bboxes[i:min(i + step, e)]
        if native:
            tiles = slice_image(img, chunk, half=half, swap_rb=bgr, device=dev)
        else:
            from .letterbox import preprocess
            size = imgsz or 640  # cfg/default.yaml imgsz
            crops = [img[b[1]:b[3], b[0]:b[2]] for b in chunk]
            if not bgr:  # preprocess flips BGR -> RGB (predictor.py:127); an RGB source is flipped back first
                crops = [c.flip(-1) for c in crops]
            tiles = preprocess(crops, imgsz=(size, size), half=half, device=dev)
        pred = detection_model(tiles, want_raw=False)[0] if hasattr(detection_model, "plan_for") else detection_model(tiles)[0]
        d, c = nms_batched(pred, conf, iou, max_det=max_det, in_place=True)
        if not native:
            scale_boxes_batched(d, c, tiles.shape[2:], [(b[3] - b[1], b[2] - b[0]) for b in chunk])
        dets.append(d)
        cnts.append(c)
    if dets:
        det, cnt = torch.cat(dets), torch.cat(cnts)
    else:  # a rank without tiles (more ranks than tiles)
        det = torch.zeros((0, max_det, 6), dtype=torch.float32, device=dev)
        cnt = torch.zeros((0,), dtype=torch.int32, device=dev)
    det, cnt = gather_detections(det, cnt, T, group)
    shifts = [[b[0], b[1]] for b in bboxes]
    if perform_standard_pred and T > 1:  # sahi adds a full-image prediction to the slices'
        from .letterbox import preprocess
        size = imgsz or max(slice_height, slice_width)
        full = preprocess([img if bgr else img.flip(-1)], imgsz=(size, size), half=half, device=dev)
        pred = detection_model(full, want_raw=False)[0] if hasattr(detection_model, "plan_for") else detection_model(full)[0]
        d, c = nms_batched(pred, conf, iou, max_det=max_det, in_place=True)
        scale_boxes_batched(d, c, full.shape[2:], [(H, W)])
        det, cnt = torch.cat((det, d)), torch.cat((cnt, c))
        shifts = shifts + [[0, 0]]
    out, n = postprocess(det, cnt, shifts, postprocess_type, postprocess_match_metric, postprocess_match_threshold,
                         postprocess_class_agnostic, full_shape=(H, W))
    return out[: int(n.item())], bboxes
