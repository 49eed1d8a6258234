"""The CPU oracle against the golden vectors produced by the reference itself (tests/golden/make_fixtures.py).

This is what PINS the oracle: every comparison here is oracle-output vs an output of /root/reference code
run in the build container on the same inputs and the same (synth_param) weights.
"""
import json

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import letterbox_ref as LB
from oracle import postproc_ref as PP
from oracle import yolo_ref as R

TOL = dict(rtol=1e-4, atol=1e-4)  # same ATen kernels on both sides; slack only for thread-count-dependent summation


def _load(name):
    return np.load(GOLDEN / name, allow_pickle=False)


@pytest.mark.parametrize("tag", ["yolov5n_detect", "yolov5s_detect", "yolo11n_detect", "yolo11s_detect", "yolo11m_detect", "yolo11n_segment",
                                 "yolov8n_segment", "bsyolo11n_detect", "bsyolo11s_detect"])
def test_graph_matches_reference(tag):
    z = _load(f"graph_{tag}.npz")
    meta = json.loads(str(z["meta"]))
    m = R.Model(meta["family"], meta["scale"], meta["nc"], meta["task"])
    # parameter names, shapes, count, strides, save-list: identical to the reference's own model
    ours = [(n, list(s)) for n, s in m.param_specs()]
    assert sorted(map(tuple, map(lambda t: (t[0], tuple(t[1])), ours))) == \
        sorted((n, tuple(s)) for n, s in meta["names"])
    assert m.num_params() == meta["nparam"]
    assert m.strides == meta["stride"]
    assert m.save == meta["save"]
    P = R.synth_params(m, meta["seed"])
    si = 0
    while f"x{si}" in z:
        x = torch.from_numpy(z[f"x{si}"])
        with torch.inference_mode():
            res = m.forward(P, x)
        if meta["task"] == "detect":
            y, raw = res
        else:
            y, (raw, mc, proto) = res
            np.testing.assert_allclose(proto.numpy(), z[f"proto{si}"], **TOL)
        # box rows are O(640) pixels (dist*stride): fp32 eps * 512 ~ 6e-5 absolute per op; the deeper BS-YOLO graph
        # (GroupNorm, branch softmax) shows up to 1.5e-3 px of thread-count-dependent summation noise
        np.testing.assert_allclose(y.numpy()[:, :4], z[f"y{si}"][:, :4], rtol=1e-5, atol=3e-3)
        np.testing.assert_allclose(y.numpy()[:, 4:], z[f"y{si}"][:, 4:], **TOL)
        for li, r in enumerate(raw):
            np.testing.assert_allclose(r.numpy(), z[f"raw{si}_{li}"], **TOL)
        si += 1
    assert si >= 1


@pytest.mark.parametrize("tag", ["yolo11n_detect", "bsyolo11n_detect", "yolov5n_detect"])
def test_per_layer_outputs_match_reference(tag):
    z = _load(f"graph_{tag}.npz")
    meta = json.loads(str(z["meta"]))
    m = R.Model(meta["family"], meta["scale"], meta["nc"], meta["task"])
    P = R.synth_params(m, meta["seed"])
    x = torch.from_numpy(z["x0"])
    ys = []
    with torch.inference_mode():
        for f, mod in m.layers[:-1]:
            if f != -1:
                x = ys[f] if isinstance(f, int) else [x if j == -1 else ys[j] for j in f]
            if mod == "up":
                x = torch.nn.functional.interpolate(x, scale_factor=2.0, mode="nearest")
            elif mod == "cat":
                x = torch.cat(x, 1)
            else:
                x = mod(P, x)
            ys.append(x)
    for i, t in enumerate(ys):
        np.testing.assert_allclose(t.numpy(), z[f"layer0_{i}"], err_msg=f"layer {i}", **TOL)


def _build_module(ctor, name="m"):
    kind, args = ctor[0], ctor[1:]
    if kind == "Conv":
        return R.Conv(name, *args)
    if kind == "ConvNoAct":
        return R.Conv(name, *args, act=False)
    if kind == "DWConv":
        return R.DWConv(name, *args)
    return getattr(R, kind)(name, *args)


def test_modules_match_reference():
    z = _load("modules.npz")
    cases = json.loads(str(z["cases"]))
    assert len(cases) >= 15
    for tag, ctor in cases.items():
        mod = _build_module(ctor)
        P = {n: R.synth_param(n, s, 7) for n, s in mod.specs()}
        with torch.inference_mode():
            y = mod(P, torch.from_numpy(z[tag + ".x"]))
        np.testing.assert_allclose(y.numpy(), z[tag + ".y"], err_msg=tag, **TOL)


def test_c3_module_matches_reference():
    """oracle C3 (block.py:3320-3334, the YOLOv5u block: Bottlenecks with k = ((1,1),(3,3))) against the reference's module."""
    z = _load("modules_c3.npz")
    for tag, ctor in json.loads(str(z["cases"])).items():
        mod = _build_module(ctor)
        P = {n: R.synth_param(n, s, 13) for n, s in mod.specs()}
        with torch.inference_mode():
            y = mod(P, torch.from_numpy(z[tag + ".x"]))
        np.testing.assert_allclose(y.numpy(), z[tag + ".y"], err_msg=tag, **TOL)


def test_bsyolo_modules_match_reference():
    """PMSFA / C3k2_gai / SCDown / MSCAAttention / ELA against the fork's own modules."""
    z = _load("modules_bsyolo.npz")
    cases = json.loads(str(z["cases"]))
    assert set(c[0] for c in cases.values()) == {"PMSFA", "C3k2_gai", "SCDown", "MSCAAttention", "ELA"}
    for tag, ctor in cases.items():
        mod = _build_module(ctor)
        P = {n: R.synth_param(n, s, 9) for n, s in mod.specs()}
        with torch.inference_mode():
            y = mod(P, torch.from_numpy(z[tag + ".x"]))
        np.testing.assert_allclose(y.numpy(), z[tag + ".y"], err_msg=tag, **TOL)


def test_val_match_matches_reference():
    """oracle/val_ref.py against the reference's own box_iou + BaseValidator.match_predictions (val_match.npz)."""
    from oracle import val_ref as V
    z = _load("val_match.npz")
    iouv = torch.linspace(0.5, 0.95, 10).numpy()
    for ci in json.loads(str(z["cases"])):
        det, lab, lcls = z[f"c{ci}.det"], z[f"c{ci}.lab"], z[f"c{ci}.lcls"]
        if len(det) and len(lab):
            np.testing.assert_allclose(V.box_iou(lab, det[:, :4]), z[f"c{ci}.iou"], rtol=1e-6, atol=1e-7)
            got = V.match_predictions(det[:, 5], lcls, z[f"c{ci}.iou"], iouv)
        else:
            got = np.zeros((len(det), 10), bool)
        assert (got == z[f"c{ci}.correct"]).all(), ci


def test_fuse_conv_bn_known_answer():
    z = _load("modules.npz")
    c = R.Conv("m", 8, 12, 3, 1)
    P = {n: R.synth_param(n, s, 11) for n, s in c.specs()}
    w, b = c.folded(P)
    np.testing.assert_allclose(w.numpy(), z["fuse.w"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(b.numpy(), z["fuse.b"], rtol=1e-6, atol=1e-7)


def test_detect_decode_known_answer():
    z = _load("modules.npz")
    det = R.Detect("d", 5, (16, 32, 64), legacy=False, strides=[8.0, 16.0, 32.0])
    raws = [torch.from_numpy(z[f"decode.raw{i}"]) for i in range(3)]
    y = det.decode(raws)
    np.testing.assert_allclose(y.numpy(), z["decode.y"], rtol=1e-5, atol=1e-5)


def test_nms_wrapper_matches_reference():
    z = _load("nms.npz")
    cases = json.loads(str(z["cases"]))
    assert len(cases) >= 9
    for tag, kw, n_out in cases:
        pred = torch.from_numpy(z[tag + ".pred"].copy())
        res = PP.non_max_suppression(pred, **kw)
        assert len(res) == n_out
        np.testing.assert_array_equal(pred[:, :4].numpy(), z[tag + ".pred_after"], err_msg=tag)  # in-place xyxy
        for i, r in enumerate(res):
            exp = z[f"{tag}.out{i}"]
            assert tuple(r.shape) == exp.shape, (tag, i, r.shape, exp.shape)
            np.testing.assert_array_equal(r.numpy(), exp, err_msg=f"{tag}[{i}]")


def test_greedy_nms_properties():
    """No golden exists for torchvision.ops.nms (parity unpinned): check the algorithm's defining properties."""
    g = torch.Generator().manual_seed(0)
    xy = torch.rand(400, 2, generator=g) * 100
    wh = torch.rand(400, 2, generator=g) * 30 + 1
    boxes = torch.cat((xy, xy + wh), 1)
    scores = torch.rand(400, generator=g)
    keep = PP.greedy_nms(boxes, scores, 0.5)
    ks = scores[keep]
    assert torch.all(ks[:-1] >= ks[1:])  # descending score order
    kb = boxes[keep]

    def iou(a, b):
        lt = torch.max(a[:, None, :2], b[None, :, :2])
        rb = torch.min(a[:, None, 2:], b[None, :, 2:])
        inter = (rb - lt).clamp(min=0).prod(-1)
        aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
        ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        return inter / (aa[:, None] + ab[None] - inter)

    m = iou(kb, kb)
    m.fill_diagonal_(0)
    assert m.max() <= 0.5  # kept boxes do not overlap above the threshold
    dropped = torch.tensor(sorted(set(range(400)) - set(keep.tolist())))
    md = iou(boxes[dropped], kb)
    # every dropped box is suppressed by a kept box with a score at least as high
    ok = ((md > 0.5) & (ks[None, :] >= scores[dropped][:, None])).any(1)
    assert bool(ok.all())
    # idempotent on its own output
    again = PP.greedy_nms(kb, ks, 0.5)
    assert again.tolist() == list(range(len(keep)))
    assert PP.greedy_nms(torch.zeros(0, 4), torch.zeros(0), 0.5).numel() == 0


def test_scale_boxes_matches_reference():
    z = _load("nms.npz")
    cases = json.loads(str(z["scale_boxes_cases"]))
    for i, (s1, s0) in enumerate(cases):
        res = PP.scale_boxes(tuple(s1), torch.from_numpy(z[f"scale_boxes{i}.in"].copy()), tuple(s0))
        np.testing.assert_array_equal(res.numpy(), z[f"scale_boxes{i}.out"])


def test_letterbox_geometry_matches_reference():
    z = _load("letterbox.npz")
    cases = json.loads(str(z["cases"]))
    assert len(cases) >= 40
    for c in cases:
        kw = dict(c["kw"])
        new_shape = tuple(kw.pop("new_shape"))
        h, w = c["shape"]
        new_unpad, (top, bottom, left, right), _ = LB.letterbox_geometry((h, w), new_shape, **kw)
        out_shape = [new_unpad[1] + top + bottom, new_unpad[0] + left + right, 3]
        assert out_shape == c["out_shape"], c
        assert [top, top + new_unpad[1], left, left + new_unpad[0]] == c["box"], c


def test_letterbox_pixels_selfconsistent():
    """Pixels are pinned only against the oracle's own cv2.resize restatement (parity unpinned upstream)."""
    z = _load("letterbox.npz")
    for k, c in enumerate(json.loads(str(z["pixel_cases"]))):
        kw = dict(c["kw"])
        new_shape = tuple(kw.pop("new_shape"))
        res = LB.letterbox(z[f"img{k}"], new_shape, **kw)
        np.testing.assert_array_equal(res, z[f"lb{k}"])


def test_resize_linear_properties():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(LB.resize_linear_u8(img, (53, 37)), img)  # identity size -> untouched
    const = np.full((20, 30, 3), 200, np.uint8)
    assert np.all(LB.resize_linear_u8(const, (77, 41)) == 200)  # constant image stays constant (weights sum to 2048)
    up = LB.resize_linear_u8(img, (106, 74))
    assert up.min() >= img.min() and up.max() <= img.max()  # convex combination
    half = LB.resize_linear_u8(img[:36, :52], (26, 18))  # exact 2x -> box mean
    s = img[:36, :52].astype(np.int32)
    assert np.array_equal(half, ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2))


def test_ap_per_class_golden():
    """oracle.val_ref.ap_per_class against the reference's own ap_per_class outputs (tests/golden/ap_per_class.npz, written by
    make_fixtures.py `ap`): AP per class and threshold, the 1000-point P / R / F1 / precision-at-0.5 curves, the max-F1 operating
    point.  Tolerance 1e-12 (float64; np.trapz sums pairwise, the restatement sequentially)."""
    from oracle import val_ref as V
    z = np.load(GOLDEN / "ap_per_class.npz")
    names = ["tp", "fp", "p", "r", "f1", "ap", "unique_classes", "p_curve", "r_curve", "f1_curve", "x", "prec_values"]
    for ci in json.loads(str(z["cases"])):
        got = V.ap_per_class(z[f"c{ci}.tp"], z[f"c{ci}.conf"], z[f"c{ci}.pred_cls"], z[f"c{ci}.target_cls"])
        for k, g in zip(names, got):
            want = z[f"c{ci}.out.{k}"]
            assert np.asarray(g).shape == want.shape, (ci, k, np.asarray(g).shape, want.shape)
            np.testing.assert_allclose(np.asarray(g, dtype=np.float64), want.astype(np.float64), rtol=0, atol=1e-12, err_msg=f"case {ci} {k}")


def test_masks_native_oracle_matches_reference_golden():
    """oracle.postproc_ref.process_mask_native / scale_masks against the reference's own outputs (utils/ops.py:696-737;
    fixture written by tests/golden/make_fixtures.py masks_native)."""
    import json
    from oracle import postproc_ref as PP
    z = np.load(GOLDEN / "masks_native.npz")
    for case in json.loads(str(z["cases"])):
        ci, shape = case["ci"], tuple(case["shape"])
        protos, coef, boxes = (torch.from_numpy(z[f"c{ci}.{k}"]) for k in ("protos", "coef", "boxes"))
        if len(coef):
            got = PP.process_mask_native(protos, coef, boxes, shape)
            assert np.array_equal(got.numpy().astype(np.uint8), z[f"c{ci}.native"])
        np.testing.assert_allclose(PP.scale_masks(protos[None, :2], shape).numpy(), z[f"c{ci}.scaled"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(PP.scale_masks(protos[None, :2], shape, padding=False).numpy(), z[f"c{ci}.scaled_nopad"], rtol=0, atol=1e-6)


def test_config1_on_the_reference_image():
    """BASELINE config 1 on ultralytics/assets/bus.jpg (tests/golden/config1_bus.npz: every expected value was produced by the
    reference -- LetterBox, YOLO11n forward, non_max_suppression, scale_boxes; make_fixtures.py config1_fixture): the oracle's
    letterbox gives the reference's pixels (crc32), its forward the reference's y at the 1000 highest-scoring anchors, its NMS and
    scale_boxes the reference's detections."""
    import zlib
    z = _load("config1_bus.npz")
    meta = json.loads(str(z["meta"]))
    bgr = z["bgr"]
    assert bgr.shape == (1080, 810, 3) and bgr.dtype == np.uint8
    lb = LB.pre_transform([bgr], (640, 640), True, 32)[0]
    assert list(lb.shape) == meta["lb_shape"] and zlib.crc32(np.ascontiguousarray(lb).tobytes()) == meta["lb_crc32"]
    x = LB.preprocess([bgr], (640, 640), half=False, pt=True, stride=32)
    m = R.Model("yolo11", "n", 80, "detect")
    P = R.synth_params(m, meta["seed"])
    for k in P:
        if ".cv3." in k and k.endswith(".2.bias"):
            P[k] = P[k] + meta["cls_shift"]
    with torch.inference_mode():
        y, _ = m.forward(P, x)
    assert list(y.shape) == meta["y_shape"]
    ytop = y[0][:, torch.from_numpy(z["y_idx"])].numpy()
    np.testing.assert_allclose(ytop[:4], z["y_top"][:4], rtol=1e-5, atol=3e-3)
    np.testing.assert_allclose(ytop[4:], z["y_top"][4:], **TOL)
    pred = PP.non_max_suppression(y.clone(), meta["conf"], meta["iou"], max_det=300)[0]
    assert pred.shape[0] == meta["n_det"]
    np.testing.assert_allclose(pred.numpy(), z["pred"], rtol=1e-5, atol=3e-3)
    boxes = PP.scale_boxes(x.shape[2:], pred[:, :4].clone(), bgr.shape)
    np.testing.assert_allclose(boxes.numpy(), z["boxes"], rtol=1e-5, atol=5e-3)
