"""CPU restatement of the reference's detection post-processing.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

  * ``greedy_nms``  : torchvision.ops.nms (un-vendored third-party dependency, ``torchvision>=0.9.0``
                      unpinned in the reference's pyproject.toml:74; call site utils/ops.py:296).
                      Restates the published CPU algorithm (torchvision/csrc/ops/cpu/nms_kernel.cpp,
                      ``nms_kernel_impl``): order = argsort(scores, descending); keep i unless
                      suppressed; suppress j when inter/(area_i+area_j-inter) > thr (strict).
                      Tie order among equal scores is implementation-defined upstream; the oracle
                      (and the HIP kernel) break ties by ascending candidate index.  PARITY UNPINNED.
  * ``non_max_suppression`` : utils/ops.py:167-316 (rotated / labels / end2end branches omitted).
  * ``xywh2xyxy`` :416-433, ``scale_boxes`` :92-127, ``clip_boxes`` :319-337,
    ``process_mask`` :663-694, ``crop_mask`` :644-660.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def greedy_nms(boxes: torch.Tensor, scores: torch.Tensor, iou_thres: float) -> torch.Tensor:
    """Indices of kept boxes, in descending-score order.  fp32 arithmetic, op order as nms_kernel_impl."""
    if boxes.numel() == 0:
        return torch.zeros((0,), dtype=torch.long)
    b = boxes.detach().to(torch.float32).cpu().numpy()
    s = scores.detach().to(torch.float32).cpu().numpy()
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = ((x2 - x1) * (y2 - y1)).astype(np.float32)
    order = np.argsort(-s, kind="stable")  # ties: ascending index
    thr = np.float32(iou_thres)
    n = len(order)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    # work in sorted space so the inner step vectorises
    x1, y1, x2, y2, areas = x1[order], y1[order], x2[order], y2[order], areas[order]
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(order[i])
        if i + 1 == n:
            break
        xx1 = np.maximum(x1[i], x1[i + 1:])
        yy1 = np.maximum(y1[i], y1[i + 1:])
        xx2 = np.minimum(x2[i], x2[i + 1:])
        yy2 = np.minimum(y2[i], y2[i + 1:])
        w = np.maximum(np.float32(0), xx2 - xx1)
        h = np.maximum(np.float32(0), yy2 - yy1)
        inter = (w * h).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[i + 1:] - inter)
        suppressed[i + 1:] |= ovr > thr
    return torch.from_numpy(np.asarray(keep, dtype=np.int64))


def xywh2xyxy(x):  # utils/ops.py:416-433
    y = torch.empty_like(x)
    xy = x[..., :2]
    wh = x[..., 2:] / 2
    y[..., :2] = xy - wh
    y[..., 2:] = xy + wh
    return y


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False,
                        multi_label=False, max_det=300, nc=0, max_nms=30000, max_wh=7680, in_place=True,
                        nms_fn=greedy_nms, boxes_xyxy=False):
    """utils/ops.py:167-316 without the wall-clock bail-out (:238,:312-314)."""
    assert 0 <= conf_thres <= 1 and 0 <= iou_thres <= 1
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    if classes is not None:
        classes = torch.tensor(classes, device=prediction.device)
    bs = prediction.shape[0]
    nc = nc or (prediction.shape[1] - 4)
    nm = prediction.shape[1] - nc - 4
    mi = 4 + nc
    xc = prediction[:, 4:mi].amax(1) > conf_thres
    multi_label &= nc > 1
    prediction = prediction.transpose(-1, -2)
    if boxes_xyxy:  # test hook: the caller already converted the boxes (fp16 parity case)
        pass
    elif in_place:
        prediction[..., :4] = xywh2xyxy(prediction[..., :4])
    else:
        prediction = torch.cat((xywh2xyxy(prediction[..., :4]), prediction[..., 4:]), dim=-1)
    output = [torch.zeros((0, 6 + nm), device=prediction.device)] * bs
    for xi, x in enumerate(prediction):
        x = x[xc[xi]]
        if not x.shape[0]:
            continue
        box, cls, mask = x.split((4, nc, nm), 1)
        if multi_label:
            i, j = torch.where(cls > conf_thres)
            x = torch.cat((box[i], x[i, 4 + j, None], j[:, None].float(), mask[i]), 1)
        else:
            conf, j = cls.max(1, keepdim=True)
            x = torch.cat((box, conf, j.float(), mask), 1)[conf.view(-1) > conf_thres]
        if classes is not None:
            x = x[(x[:, 5:6] == classes).any(1)]
        n = x.shape[0]
        if not n:
            continue
        if n > max_nms:
            x = x[x[:, 4].argsort(descending=True, stable=True)[:max_nms]]
        c = x[:, 5:6] * (0 if agnostic else max_wh)
        scores = x[:, 4]
        boxes = x[:, :4] + c
        i = nms_fn(boxes, scores, iou_thres)
        i = i[:max_det]
        output[xi] = x[i]
    return output


def clip_boxes(boxes, shape):  # utils/ops.py:319-337
    boxes[..., 0] = boxes[..., 0].clamp(0, shape[1])
    boxes[..., 1] = boxes[..., 1].clamp(0, shape[0])
    boxes[..., 2] = boxes[..., 2].clamp(0, shape[1])
    boxes[..., 3] = boxes[..., 3].clamp(0, shape[0])
    return boxes


def scale_boxes(img1_shape, boxes, img0_shape, ratio_pad=None, padding=True, xywh=False):  # utils/ops.py:92-127
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1),
               round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1))
    else:
        gain = ratio_pad[0][0]
        pad = ratio_pad[1]
    if padding:
        boxes[..., 0] -= pad[0]
        boxes[..., 1] -= pad[1]
        if not xywh:
            boxes[..., 2] -= pad[0]
            boxes[..., 3] -= pad[1]
    boxes[..., :4] /= gain
    return clip_boxes(boxes, img0_shape)


def crop_mask(masks, boxes):  # utils/ops.py:644-660
    _, h, w = masks.shape
    x1, y1, x2, y2 = torch.chunk(boxes[:, :, None], 4, 1)
    r = torch.arange(w, dtype=x1.dtype)[None, None, :]
    c = torch.arange(h, dtype=x1.dtype)[None, :, None]
    return masks * ((r >= x1) * (r < x2) * (c >= y1) * (c < y2))


def process_mask(protos, masks_in, bboxes, shape, upsample=False):  # utils/ops.py:663-694
    c, mh, mw = protos.shape
    ih, iw = shape
    masks = (masks_in @ protos.float().view(c, -1)).view(-1, mh, mw)
    width_ratio = mw / iw
    height_ratio = mh / ih
    db = bboxes.clone()
    db[:, 0] *= width_ratio
    db[:, 2] *= width_ratio
    db[:, 3] *= height_ratio
    db[:, 1] *= height_ratio
    masks = crop_mask(masks, db)
    if upsample:
        masks = F.interpolate(masks[None], shape, mode="bilinear", align_corners=False)[0]
    return masks.gt_(0.0)


def scale_masks(masks, shape, padding=True):  # utils/ops.py:712-737
    import torch.nn.functional as F
    mh, mw = masks.shape[2:]
    gain = min(mh / shape[0], mw / shape[1])
    pad = [mw - shape[1] * gain, mh - shape[0] * gain]
    if padding:
        pad[0] /= 2
        pad[1] /= 2
    top, left = (int(pad[1]), int(pad[0])) if padding else (0, 0)
    bottom, right = (int(mh - pad[1]), int(mw - pad[0]))
    masks = masks[..., top:bottom, left:right]
    return F.interpolate(masks, shape, mode="bilinear", align_corners=False)


def process_mask_native(protos, masks_in, bboxes, shape):  # utils/ops.py:696-709
    c, mh, mw = protos.shape
    masks = (masks_in @ protos.float().view(c, -1)).view(-1, mh, mw)
    masks = scale_masks(masks[None], shape)[0]
    masks = crop_mask(masks, bboxes)
    return masks.gt_(0.0)
