"""CPU restatement of the sliced-inference arithmetic the reference delegates to the `sahi` package.

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
the product path (bs_yolo_amd/sahi.py drives HIP kernels and has no CPU fallback).

PARITY UNPINNED.  The reference's sliced path is `detect-sahi.py:1-13` (sahi.predict.predict, slice 800 x 800,
overlap 0) and `examples/YOLOv8-SAHI-Inference-Video/yolov8_sahi.py:70-75` (sahi.predict.get_sliced_prediction,
slice 512 x 512).  `sahi` is not vendored under /root/reference, is not listed in pyproject.toml (no pinned
version), is not installed in this image, and the reference holds no test or fixture for it.  What follows restates
the published algorithm of sahi 0.11.x from its documentation:

  * get_slice_bboxes         -- sahi/slicing.py `get_slice_bboxes` (explicit slice size; the auto-slice branch is not
                                restated)
  * tile_detections          -- sahi/models/ultralytics.py `_create_object_prediction_list_from_original_predictions`:
                                clamp to >= 0, clamp to the FULL image shape (in slice coordinates, as sahi does), drop
                                boxes that are not x1 < x2 and y1 < y2, then shift by the slice origin
  * greedy_nmm               -- sahi/postprocess/combine.py `greedy_nmm` / `batched_greedy_nmm` (float32 torch
                                arithmetic: a candidate stays when metric < threshold)
  * merge                    -- `GreedyNMMPostprocess.__call__` + sahi/postprocess/utils.py `has_match`
                                (metric > threshold, float64 numpy arithmetic on the growing merged box),
                                `calculate_box_union`, `get_merged_score` (max), `get_merged_category` (the
                                higher score's; the candidate's on a tie)
  * nms                      -- sahi/postprocess/combine.py `nms` / `batched_nms` (same keep set, no merging)
  * nmm                      -- sahi/postprocess/combine.py `nmm` / `batched_nmm` + `NMMPostprocess.__call__` (the non-greedy
                                merge: every prediction, in score order, hands the boxes it matches to ITS keeper, so a
                                keeper also collects boxes it does not overlap itself; merging as above, in list order)
  * LSNMS                    -- `LSNMSPostprocess` calls the `lsnms` package (an R-tree accelerated non-maximum suppression,
                                IOU only, sahi marks it experimental): same keep set as `nms` with the IOU metric, which is
                                what postprocess() runs for it; like sahi it refuses the IOS metric
sahi's defaults (predict(): postprocess_type GREEDYNMM, match metric IOS, threshold 0.5, class_agnostic False) are
the defaults here.  Score ties: sahi's `scores.argsort()` is unstable; here ties go to the lower flat index
(tile-major, row-minor), which is what both this file and the HIP path implement.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def get_slice_bboxes(image_height, image_width, slice_height, slice_width, overlap_height_ratio=0.2,
                     overlap_width_ratio=0.2):
    """[[x0, y0, x1, y1], ...] row-major; border slices are shifted back inside the image (so they overlap)."""
    out = []
    y_max = y_min = 0
    y_overlap = int(overlap_height_ratio * slice_height)
    x_overlap = int(overlap_width_ratio * slice_width)
    while y_max < image_height:
        x_min = x_max = 0
        y_max = y_min + slice_height
        while x_max < image_width:
            x_max = x_min + slice_width
            if y_max > image_height or x_max > image_width:
                xmax, ymax = min(image_width, x_max), min(image_height, y_max)
                out.append([max(0, xmax - slice_width), max(0, ymax - slice_height), xmax, ymax])
            else:
                out.append([x_min, y_min, x_max, y_max])
            x_min = x_max - x_overlap
        y_min = y_max - y_overlap
    return out


def slice_image(img_hwc_u8, bboxes, swap_rb=True):
    """(H, W, 3) u8 -> (T, 3, h, w) float32 in [0, 1] (crop, channel swap, /255: what the predictor's preprocess makes of
    a crop that already has the model's input size)."""
    tiles = []
    for x0, y0, x1, y1 in bboxes:
        t = img_hwc_u8[y0:y1, x0:x1]
        if swap_rb:
            t = t[..., ::-1]
        tiles.append(np.ascontiguousarray(t.transpose(2, 0, 1)).astype(f32) / f32(255))
    return np.stack(tiles)


def tile_detections(det, counts, shifts, full_shape=None):
    """det (T, max_det, >=6) f32 rows [x1, y1, x2, y2, score, cls] in tile pixels, counts (T,), shifts (T, 2) = (ox, oy)
    -> (N, 6) float32 in full-image pixels, flat (tile-major) order, invalid boxes dropped."""
    rows = []
    for t in range(det.shape[0]):
        for j in range(int(counts[t])):
            b = det[t, j, :4].astype(f32).copy()
            b = np.maximum(b, f32(0))
            if full_shape is not None:
                h, w = full_shape
                b[0] = min(f32(w), b[0]); b[1] = min(f32(h), b[1]); b[2] = min(f32(w), b[2]); b[3] = min(f32(h), b[3])
            if not (b[0] < b[2]) or not (b[1] < b[3]):
                continue
            ox, oy = f32(shifts[t][0]), f32(shifts[t][1])
            rows.append([b[0] + ox, b[1] + oy, b[2] + ox, b[3] + oy, f32(det[t, j, 4]), f32(det[t, j, 5])])
    return np.asarray(rows, dtype=f32).reshape(-1, 6)


def _metric32(kept, others, metric):
    """greedy_nmm's float32 metric of one kept box against an array of boxes."""
    xx1 = np.maximum(others[:, 0], kept[0]); yy1 = np.maximum(others[:, 1], kept[1])
    xx2 = np.minimum(others[:, 2], kept[2]); yy2 = np.minimum(others[:, 3], kept[3])
    w = np.maximum(xx2 - xx1, f32(0)); h = np.maximum(yy2 - yy1, f32(0))
    inter = (w * h).astype(f32)
    rem = ((others[:, 2] - others[:, 0]) * (others[:, 3] - others[:, 1])).astype(f32)
    ka = f32((kept[2] - kept[0]) * (kept[3] - kept[1]))
    with np.errstate(divide="ignore", invalid="ignore"):
        if metric == "IOU":
            return inter / ((rem - inter) + ka)
        return inter / np.minimum(rem, ka)


def greedy_nmm(boxes, metric="IOS", threshold=0.5):
    """boxes (n, 6) f32 -> {keep_index: [merge candidates, score descending]} in keep (score descending) order."""
    n = boxes.shape[0]
    order = sorted(range(n), key=lambda i: (-float(boxes[i, 4]), i))  # descending score, ties by lower index
    keep_to_merge = {}
    alive = order
    while alive:
        idx, rest = alive[0], alive[1:]
        if not rest:
            keep_to_merge[idx] = []
            break
        v = _metric32(boxes[idx], boxes[rest], metric)
        unmatched = v < f32(threshold)          # NaN compares False -> matched, as in torch
        keep_to_merge[idx] = [r for r, u in zip(rest, unmatched) if not u]
        alive = [r for r, u in zip(rest, unmatched) if u]
    return keep_to_merge


def nmm(boxes, metric="IOS", threshold=0.5):
    """boxes (n, 6) f32 -> {keep_index: [merge candidates in the order sahi's loop appends them]}, keeps in descending score
    order.  Prediction i (descending score; ties by lower index) matches every OTHER box whose metric is not < threshold;
    if i has no keeper it becomes one and takes its unassigned matches, otherwise its keeper takes those of them that are
    neither keepers nor assigned.  Within one step the matches are appended in ascending score order (sahi flips the
    descending list)."""
    n = boxes.shape[0]
    order = sorted(range(n), key=lambda i: (-float(boxes[i, 4]), i))
    keep_to_merge, merge_to_keep = {}, {}
    for pred in order:
        others = [o for o in order if o != pred]
        if not others:
            if pred not in merge_to_keep:
                keep_to_merge[pred] = []
            break
        v = _metric32(boxes[pred], boxes[others], metric)
        matched = [o for o, u in zip(others, v < f32(threshold)) if not u][::-1]
        if pred not in merge_to_keep:
            keep_to_merge[pred] = []
            for m in matched:
                if m not in merge_to_keep:
                    keep_to_merge[pred].append(m)
                    merge_to_keep[m] = pred
        else:
            keep = merge_to_keep[pred]
            for m in matched:
                if m not in keep_to_merge and m not in merge_to_keep:
                    keep_to_merge[keep].append(m)
                    merge_to_keep[m] = keep
    return keep_to_merge


def _metric64(b1, b2, metric):
    """has_match's float64 metric (numpy on Python floats)."""
    b1 = np.asarray(b1[:4], dtype=np.float64); b2 = np.asarray(b2[:4], dtype=np.float64)
    a1 = (b1[2] - b1[0]) * (b1[3] - b1[1]); a2 = (b2[2] - b2[0]) * (b2[3] - b2[1])
    lt = np.maximum(b1[:2], b2[:2]); rb = np.minimum(b1[2:], b2[2:])
    wh = np.clip(rb - lt, 0, None)
    inter = wh[0] * wh[1]
    with np.errstate(divide="ignore", invalid="ignore"):
        if metric == "IOU":
            return inter / (a1 + a2 - inter)
        return inter / np.minimum(a1, a2)


def postprocess(boxes, postprocess_type="GREEDYNMM", metric="IOS", threshold=0.5, class_agnostic=False):
    """boxes (N, 6) f32 full-image detections -> (K, 6) f32.  Order: class ascending (unless class_agnostic), score
    descending inside -- the iteration order of sahi's keep_to_merge_list."""
    boxes = np.asarray(boxes, dtype=f32).reshape(-1, 6)
    if postprocess_type not in ("GREEDYNMM", "NMM", "NMS", "LSNMS"):
        raise ValueError(postprocess_type)
    if postprocess_type == "LSNMS":
        if metric != "IOU":
            raise NotImplementedError("LSNMS: IOU only (as in sahi)")
        postprocess_type = "NMS"
    groups = [np.arange(boxes.shape[0])] if class_agnostic else \
        [np.nonzero(boxes[:, 5] == c)[0] for c in np.unique(boxes[:, 5])]
    out = []
    for g in groups:
        if g.size == 0:
            continue
        sub = boxes[g]
        assign = nmm if postprocess_type == "NMM" else greedy_nmm
        for keep, cands in assign(sub, metric, threshold).items():
            cur = [float(v) for v in sub[keep]]
            if postprocess_type in ("GREEDYNMM", "NMM"):
                for c in cands:
                    cand = [float(v) for v in sub[c]]
                    if _metric64(cur, cand, metric) > threshold:
                        cls = cur[5] if cur[4] > cand[4] else cand[5]
                        cur = [min(cur[0], cand[0]), min(cur[1], cand[1]), max(cur[2], cand[2]), max(cur[3], cand[3]),
                               max(cur[4], cand[4]), cls]
            out.append(cur)
    return np.asarray(out, dtype=f32).reshape(-1, 6)


def sliced_merge(det, counts, shifts, full_shape=None, **kw):
    return postprocess(tile_detections(det, counts, shifts, full_shape), **kw)
