"""Stock model descriptions in the reference's yaml-dict schema (what ``model.yaml`` holds on a live reference model).

The reference ships these graphs as cfg/models/11/yolo11-seg.yaml:15-47 (stock YOLO11 backbone+neck),
cfg/models/v8/yolov8-seg.yaml:15-46, cfg/models/v5/yolov5.yaml:14-50 and cfg/models/11/yolo11.yaml:15-52 (the modified BS-YOLO
graph, family "bsyolo11").  They are
restated here as data so the engine can be built where the reference is not installed (benchmarks, GPU box).
"""
from __future__ import annotations

import copy

_UP = [-1, 1, "nn.Upsample", [None, 2, "nearest"]]


def _cat(j):
    return [[-1, j], 1, "Concat", [1]]


_YOLO11 = {
    "scales": {"n": [0.50, 0.25, 1024], "s": [0.50, 0.50, 1024], "m": [0.50, 1.00, 512], "l": [1.00, 1.00, 512],
               "x": [1.00, 1.50, 512]},
    "backbone": [
        [-1, 1, "Conv", [64, 3, 2]], [-1, 1, "Conv", [128, 3, 2]], [-1, 2, "C3k2", [256, False, 0.25]],
        [-1, 1, "Conv", [256, 3, 2]], [-1, 2, "C3k2", [512, False, 0.25]], [-1, 1, "Conv", [512, 3, 2]],
        [-1, 2, "C3k2", [512, True]], [-1, 1, "Conv", [1024, 3, 2]], [-1, 2, "C3k2", [1024, True]],
        [-1, 1, "SPPF", [1024, 5]], [-1, 2, "C2PSA", [1024]],
    ],
    "head": [
        _UP, _cat(6), [-1, 2, "C3k2", [512, False]],
        _UP, _cat(4), [-1, 2, "C3k2", [256, False]],
        [-1, 1, "Conv", [256, 3, 2]], _cat(13), [-1, 2, "C3k2", [512, False]],
        [-1, 1, "Conv", [512, 3, 2]], _cat(10), [-1, 2, "C3k2", [1024, True]],
    ],
    "detect_from": [16, 19, 22],
}

_YOLOV8 = {
    "scales": {"n": [0.33, 0.25, 1024], "s": [0.33, 0.50, 1024], "m": [0.67, 0.75, 768], "l": [1.00, 1.00, 512],
               "x": [1.00, 1.25, 512]},
    "backbone": [
        [-1, 1, "Conv", [64, 3, 2]], [-1, 1, "Conv", [128, 3, 2]], [-1, 3, "C2f", [128, True]],
        [-1, 1, "Conv", [256, 3, 2]], [-1, 6, "C2f", [256, True]], [-1, 1, "Conv", [512, 3, 2]],
        [-1, 6, "C2f", [512, True]], [-1, 1, "Conv", [1024, 3, 2]], [-1, 3, "C2f", [1024, True]],
        [-1, 1, "SPPF", [1024, 5]],
    ],
    "head": [
        _UP, _cat(6), [-1, 3, "C2f", [512]],
        _UP, _cat(4), [-1, 3, "C2f", [256]],
        [-1, 1, "Conv", [256, 3, 2]], _cat(12), [-1, 3, "C2f", [512]],
        [-1, 1, "Conv", [512, 3, 2]], _cat(9), [-1, 3, "C2f", [1024]],
    ],
    "detect_from": [15, 18, 21],
}

# The fork's own graph, cfg/models/11/yolo11.yaml:15-52 (nc = 12 there): C3k2_gai / SCDown / MSCAAttention in the backbone,
# an ELA gate after every neck C3k2, Detect on the ELA outputs.  Layer 21 concatenates layer 13, itself a Concat.
_BSYOLO11 = {
    "scales": _YOLO11["scales"],
    "backbone": [
        [-1, 1, "Conv", [64, 3, 2]], [-1, 1, "Conv", [128, 3, 2]], [-1, 2, "C3k2_gai", [256, False, 0.25]],
        [-1, 1, "Conv", [256, 3, 2]], [-1, 2, "C3k2_gai", [512, False, 0.25]], [-1, 1, "SCDown", [512, 3, 2]],
        [-1, 2, "C3k2_gai", [512, True]], [-1, 1, "SCDown", [1024, 3, 2]], [-1, 2, "C3k2_gai", [1024, True]],
        [-1, 1, "SPPF", [1024, 5]], [-1, 2, "C2PSA", [1024]], [-1, 1, "MSCAAttention", []],
    ],
    "head": [
        _UP, _cat(6), [-1, 2, "C3k2", [512, False]], [-1, 1, "ELA", [512]],
        _UP, _cat(4), [-1, 2, "C3k2", [256, False]], [-1, 1, "ELA", [256]],
        [-1, 1, "Conv", [256, 3, 2]], _cat(13), [-1, 2, "C3k2", [512, False]], [-1, 1, "ELA", [512]],
        [-1, 1, "SCDown", [512, 3, 2]], _cat(10), [-1, 2, "C3k2", [1024, True]], [-1, 1, "ELA", [1024]],
    ],
    "detect_from": [19, 23, 27],
}

# YOLOv5u, cfg/models/v5/yolov5.yaml:14-50: 6x6 stride-2 pad-2 stem, C3 blocks, SPPF, the anchor-free Detect head
_YOLOV5 = {
    "scales": {"n": [0.33, 0.25, 1024], "s": [0.33, 0.50, 1024], "m": [0.67, 0.75, 1024], "l": [1.00, 1.00, 1024],
               "x": [1.33, 1.25, 1024]},
    "backbone": [
        [-1, 1, "Conv", [64, 6, 2, 2]], [-1, 1, "Conv", [128, 3, 2]], [-1, 3, "C3", [128]],
        [-1, 1, "Conv", [256, 3, 2]], [-1, 6, "C3", [256]], [-1, 1, "Conv", [512, 3, 2]],
        [-1, 9, "C3", [512]], [-1, 1, "Conv", [1024, 3, 2]], [-1, 3, "C3", [1024]],
        [-1, 1, "SPPF", [1024, 5]],
    ],
    "head": [
        [-1, 1, "Conv", [512, 1, 1]], _UP, _cat(6), [-1, 3, "C3", [512, False]],
        [-1, 1, "Conv", [256, 1, 1]], _UP, _cat(4), [-1, 3, "C3", [256, False]],
        [-1, 1, "Conv", [256, 3, 2]], _cat(14), [-1, 3, "C3", [512, False]],
        [-1, 1, "Conv", [512, 3, 2]], _cat(10), [-1, 3, "C3", [1024, False]],
    ],
    "detect_from": [17, 20, 23],
}

_FAMILIES = {"yolo11": _YOLO11, "yolov8": _YOLOV8, "bsyolo11": _BSYOLO11, "yolov5": _YOLOV5}


def stock_cfg(family: str = "yolo11", scale: str = "s", nc: int = 80, task: str = "detect") -> dict:
    """e.g. stock_cfg("yolo11", "s") == what the reference builds for 'yolo11s.yaml' upstream."""
    fam = copy.deepcopy(_FAMILIES[family])
    head = list(fam["head"])
    if task == "detect":
        head.append([fam["detect_from"], 1, "Detect", ["nc"]])
    elif task == "segment":
        head.append([fam["detect_from"], 1, "Segment", ["nc", 32, 256]])
    else:
        raise ValueError(task)
    return {"nc": nc, "scale": scale, "scales": fam["scales"], "backbone": fam["backbone"], "head": head}
