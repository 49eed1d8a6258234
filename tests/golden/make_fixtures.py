#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE (build container only).

    python tests/golden/make_fixtures.py            # writes tests/golden/*.npz

The reference (/root/reference, Ultralytics 8.3.56 fork) is imported in-process.  Third-party modules that are
absent from this image (cv2, pywt, seaborn, cpuinfo, torchvision) are satisfied with in-process ``sys.modules``
entries; nothing of the reference is modified or copied.  The two third-party *functions* the hot path calls are
bound to the oracle's restatements so that the reference's own wrapper logic runs end to end:
    torchvision.ops.nms  -> oracle.postproc_ref.greedy_nms      (utils/ops.py:296)
    cv2.resize / cv2.copyMakeBorder -> oracle.letterbox_ref     (data/augment.py:1586-1591)
(fixtures produced through those two are therefore pinned for the wrapper logic / geometry only; see oracle/__init__.py).

Model parameters are NOT stored: both sides regenerate them from ``oracle.yolo_ref.synth_param(name, shape, seed)``
keyed by the reference's own state_dict names, which this script writes INTO the reference modules.
"""
import importlib.metadata as md
import json
import os
import sys
import types
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
os.environ.update(YOLO_OFFLINE="true", YOLO_AUTOINSTALL="false", YOLO_CONFIG_DIR="/tmp/yolocfg",
                  PYTHONDONTWRITEBYTECODE="1", YOLO_VERBOSE="false")
sys.dont_write_bytecode = True
os.makedirs("/tmp/yolocfg", exist_ok=True)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import letterbox_ref, postproc_ref  # noqa: E402
from oracle.yolo_ref import synth_param  # noqa: E402


class _Dummy(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Dummy(self.__name__ + "." + k)

    def __call__(self, *a, **k):
        return None


for _n in ("pywt", "pywt.data", "seaborn", "cpuinfo"):
    sys.modules[_n] = _Dummy(_n)

cv2 = _Dummy("cv2")
cv2.INTER_LINEAR = 1
cv2.BORDER_CONSTANT = 0


def _cv2_resize(img, dsize, interpolation=1):
    return letterbox_ref.resize_linear_u8(img, dsize)


def _cv2_border(img, top, bottom, left, right, borderType, value=(114, 114, 114)):
    h, w = img.shape[:2]
    out = np.empty((h + top + bottom, w + left + right, img.shape[2]), dtype=img.dtype)
    out[:] = np.asarray(value, dtype=img.dtype)
    out[top:top + h, left:left + w] = img
    return out


cv2.resize = _cv2_resize
cv2.copyMakeBorder = _cv2_border
sys.modules["cv2"] = cv2

tv = types.ModuleType("torchvision")
tv.ops = types.ModuleType("torchvision.ops")
tv.ops.nms = postproc_ref.greedy_nms
tv.__version__ = "0.20.0"
sys.modules["torchvision"] = tv
sys.modules["torchvision.ops"] = tv.ops
_v = md.version
md.version = lambda n: "0.20.0" if n == "torchvision" else _v(n)

sys.path.insert(0, "/root/reference")
import yaml  # noqa: E402
from ultralytics.data.augment import LetterBox  # noqa: E402
from ultralytics.nn.modules import block as rb, conv as rc, head as rh  # noqa: E402
from ultralytics.nn.tasks import DetectionModel, SegmentationModel  # noqa: E402
from ultralytics.utils import ops as rops  # noqa: E402
from ultralytics.utils.torch_utils import initialize_weights  # noqa: E402

CFG = Path("/root/reference/ultralytics/cfg/models")


def fill(module, prefix, seed):
    """Write synth_param values into every parameter/buffer of a reference module (by state_dict name)."""
    names = []
    sd = module.state_dict()
    for k, t in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        name = prefix + k
        t.copy_(synth_param(name, t.shape, seed))
        names.append((name, list(t.shape)))
    return names


def stock_cfg(family, scale, task, nc=80):
    if family == "bsyolo11":  # the fork's own graph, exactly as shipped (cfg/models/11/yolo11.yaml, nc = 12)
        d = yaml.safe_load(open(CFG / "11" / "yolo11.yaml"))
        assert task == "detect"
    elif family == "yolov5":  # YOLOv5u (cfg/models/v5/yolov5.yaml: 6x6 stem, C3 blocks, anchor-free Detect)
        d = yaml.safe_load(open(CFG / "v5" / "yolov5.yaml"))
        assert task == "detect"
    elif family == "yolo11":
        d = yaml.safe_load(open(CFG / "11" / "yolo11-seg.yaml"))  # stock backbone/neck (yolo11.yaml is the BS-YOLO graph)
        if task == "detect":
            d["head"][-1] = [[16, 19, 22], 1, "Detect", ["nc"]]
    else:
        d = yaml.safe_load(open(CFG / "v8" / "yolov8-seg.yaml"))
        if task == "detect":
            d["head"][-1] = [[15, 18, 21], 1, "Detect", ["nc"]]
    d["scale"] = scale
    d["nc"] = nc
    return d


def build_ref(family, scale, task, nc=80, seed=0):
    cls = DetectionModel if task == "detect" else SegmentationModel
    m = cls(stock_cfg(family, scale, task, nc), ch=3, nc=nc, verbose=False).eval()
    names = fill(m, "", seed)
    nparam = sum(p.numel() for p in m.parameters())
    m.fuse(verbose=False)
    return m, names, nparam


def graph_fixture(tag, family, scale, task, shapes, nc=80, seed=0, keep_layers=False):
    m, names, nparam = build_ref(family, scale, task, nc, seed)
    out = {"meta": json.dumps({"family": family, "scale": scale, "task": task, "nc": nc, "seed": seed,
                               "names": names, "nparam": nparam, "stride": [float(s) for s in m.stride],
                               "save": list(m.save)})}
    for si, (b, h, w) in enumerate(shapes):
        g = torch.Generator().manual_seed(100 + si)
        x = torch.rand(b, 3, h, w, generator=g)
        layer_out = {}
        hooks = []
        if keep_layers and si == 0:
            for i, layer in enumerate(list(m.model)[:-1]):
                hooks.append(layer.register_forward_hook(
                    lambda mod, inp, o, i=i: layer_out.__setitem__(i, o.detach().clone())))
        with torch.inference_mode():
            res = m(x)
        for hk in hooks:
            hk.remove()
        out[f"x{si}"] = x.numpy()
        if task == "detect":
            y, raw = res
            out[f"y{si}"] = y.numpy()
            for li, r in enumerate(raw):
                out[f"raw{si}_{li}"] = r.numpy()
        else:
            y, (raw, mc, p) = res
            out[f"y{si}"] = y.numpy()
            out[f"proto{si}"] = p.numpy()
            for li, r in enumerate(raw):
                out[f"raw{si}_{li}"] = r.numpy()
        for i, o in layer_out.items():
            out[f"layer{si}_{i}"] = o.numpy()
    np.savez_compressed(HERE / f"graph_{tag}.npz", **out)
    print("wrote", tag, {k: (v.shape if hasattr(v, "shape") else "") for k, v in out.items() if k != "meta"})


def module_fixtures():
    """Per-module known-answer vectors: reference module (eval, fused like BaseModel.fuse) in -> out."""
    out = {}
    cases = {}

    def fuse_all(mod):
        from ultralytics.utils.torch_utils import fuse_conv_and_bn
        for sm in mod.modules():
            if isinstance(sm, rc.Conv) and hasattr(sm, "bn"):
                sm.conv = fuse_conv_and_bn(sm.conv, sm.bn)
                delattr(sm, "bn")
                sm.forward = sm.forward_fuse

    def add(tag, mod, x, ctor):
        mod.eval()
        initialize_weights(mod)  # BN eps 1e-3 (torch_utils.py:417-427)
        fill(mod, "m.", 7)
        fuse_all(mod)
        with torch.inference_mode():
            y = mod(x)
        out[tag + ".x"] = x.numpy()
        out[tag + ".y"] = y.numpy()
        cases[tag] = ctor

    g = torch.Generator().manual_seed(5)

    def rnd(*s):
        return torch.randn(*s, generator=g)

    add("conv_k1", rc.Conv(16, 32, 1, 1), rnd(2, 16, 12, 10), ["Conv", 16, 32, 1, 1])
    add("conv_k3", rc.Conv(16, 24, 3, 1), rnd(2, 16, 12, 10), ["Conv", 16, 24, 3, 1])
    add("conv_k3s2", rc.Conv(3, 16, 3, 2), rnd(2, 3, 32, 24), ["Conv", 3, 16, 3, 2])
    add("conv_k3s2_odd", rc.Conv(8, 16, 3, 2), rnd(1, 8, 13, 11), ["Conv", 8, 16, 3, 2])
    add("conv_noact", rc.Conv(32, 16, 1, 1, act=False), rnd(2, 32, 8, 8), ["ConvNoAct", 32, 16, 1, 1])
    add("dwconv", rc.DWConv(32, 32, 3, 1), rnd(2, 32, 9, 7), ["DWConv", 32, 32, 3, 1])
    add("bottleneck", rb.Bottleneck(32, 32, True), rnd(2, 32, 10, 10), ["Bottleneck", 32, 32, True])
    add("bottleneck_noadd", rb.Bottleneck(32, 32, False), rnd(2, 32, 10, 10), ["Bottleneck", 32, 32, False])
    add("c3k", rb.C3k(32, 32, 2), rnd(2, 32, 8, 8), ["C3k", 32, 32, 2])
    add("c3k2_f", rb.C3k2(32, 64, 1, False, 0.25), rnd(2, 32, 12, 12), ["C3k2", 32, 64, 1, False, 0.25])
    add("c3k2_t", rb.C3k2(64, 64, 1, True), rnd(2, 64, 8, 8), ["C3k2", 64, 64, 1, True])
    add("c2f", rb.C2f(32, 32, 2, True), rnd(2, 32, 8, 8), ["C2f", 32, 32, 2, True])
    add("sppf", rb.SPPF(64, 64, 5), rnd(2, 64, 10, 7), ["SPPF", 64, 64, 5])
    add("attention", rb.Attention(128, num_heads=2, attn_ratio=0.5), rnd(2, 128, 10, 10), ["Attention", 128, 2, 0.5])
    add("psablock", rb.PSABlock(128, 0.5, 2), rnd(2, 128, 6, 5), ["PSABlock", 128, 0.5, 2])
    add("c2psa", rb.C2PSA(256, 256, 1), rnd(1, 256, 8, 8), ["C2PSA", 256, 256, 1])
    add("proto", rb.Proto(32, 32, 16), rnd(1, 32, 8, 8), ["Proto", 32, 32, 16])
    out["cases"] = json.dumps(cases)

    # fuse_conv_and_bn known answer (utils/torch_utils.py:242-269)
    from ultralytics.utils.torch_utils import fuse_conv_and_bn
    c = rc.Conv(8, 12, 3, 1).eval()
    initialize_weights(c)
    fill(c, "m.", 11)
    f = fuse_conv_and_bn(c.conv, c.bn)
    out["fuse.w"] = f.weight.detach().numpy()
    out["fuse.b"] = f.bias.detach().numpy()

    # Detect decode (head.py:100-131 + block.py:58-77 + tal.py:371-395) on tiny raw maps
    det = rh.Detect(nc=5, ch=(16, 32, 64)).eval()
    det.stride = torch.tensor([8.0, 16.0, 32.0])
    raws = [rnd(2, 69, 8, 6), rnd(2, 69, 4, 3), rnd(2, 69, 2, 2)]
    with torch.inference_mode():
        y = det._inference([r.clone() for r in raws])
    for i, r in enumerate(raws):
        out[f"decode.raw{i}"] = r.numpy()
    out["decode.y"] = y.numpy()
    np.savez_compressed(HERE / "modules.npz", **out)
    print("wrote modules", len(cases), "cases")


def bsyolo_module_fixtures():
    """Known-answer vectors for the BS-YOLO-only modules: reference module in -> out."""
    from ultralytics.nn.Addmodules.ELA import ELA as RefELA
    from ultralytics.nn.Addmodules.MSCA import MSCAAttention as RefMSCA
    from ultralytics.utils.torch_utils import fuse_conv_and_bn
    out, cases = {}, {}

    def fuse_all(mod):
        for sm in mod.modules():
            if isinstance(sm, rc.Conv) and hasattr(sm, "bn"):
                sm.conv = fuse_conv_and_bn(sm.conv, sm.bn)
                delattr(sm, "bn")
                sm.forward = sm.forward_fuse

    def add(tag, mod, x, ctor):
        mod.eval()
        initialize_weights(mod)
        fill(mod, "m.", 9)
        fuse_all(mod)
        with torch.inference_mode():
            y = mod(x)
        out[tag + ".x"] = x.numpy()
        out[tag + ".y"] = y.numpy()
        cases[tag] = ctor

    g = torch.Generator().manual_seed(6)

    def rnd(*s):
        return torch.randn(*s, generator=g)

    add("pmsfa", rb.PMSFA(32), rnd(2, 32, 11, 9), ["PMSFA", 32])
    add("c3k2_gai_f", rb.C3k2_gai(32, 64, 1, False, 0.25), rnd(2, 32, 12, 12), ["C3k2_gai", 32, 64, 1, False, 0.25])
    add("c3k2_gai_t", rb.C3k2_gai(64, 64, 1, True), rnd(1, 64, 8, 8), ["C3k2_gai", 64, 64, 1, True])
    add("scdown", rb.SCDown(32, 64, 3, 2), rnd(2, 32, 13, 10), ["SCDown", 32, 64, 3, 2])
    add("msca", RefMSCA(32), rnd(2, 32, 9, 12), ["MSCAAttention", 32])
    add("ela64", RefELA(64), rnd(2, 64, 8, 6), ["ELA", 64])
    add("ela256", RefELA(256), rnd(1, 256, 5, 7), ["ELA", 256])
    # widths that are not multiples of 16 (round 2): GroupNorm(max(1, c // 16), c) -> 2 groups of 20, 1 group of 24
    add("ela40", RefELA(40), rnd(2, 40, 7, 9), ["ELA", 40])
    add("ela24", RefELA(24), rnd(1, 24, 6, 5), ["ELA", 24])
    # PMSFA widths that are not multiples of 16 (round 3): halves of 12 / 6 / 4, quarters of 6 / 3 / 2 channels; C3k2_gai with c3k=True
    # nests PMSFA(c / 2) inside C3k_gai
    add("pmsfa24", rb.PMSFA(24), rnd(2, 24, 9, 11), ["PMSFA", 24])
    add("pmsfa12", rb.PMSFA(12), rnd(1, 12, 7, 8), ["PMSFA", 12])
    add("pmsfa8", rb.PMSFA(8), rnd(2, 8, 6, 6), ["PMSFA", 8])
    add("c3k2_gai_f24", rb.C3k2_gai(24, 48, 1, False, 0.5), rnd(1, 24, 10, 8), ["C3k2_gai", 24, 48, 1, False, 0.5])
    add("c3k2_gai_t24", rb.C3k2_gai(48, 48, 1, True), rnd(1, 48, 8, 8), ["C3k2_gai", 48, 48, 1, True])
    out["cases"] = json.dumps(cases)
    np.savez_compressed(HERE / "modules_bsyolo.npz", **out)
    print("wrote bsyolo modules", len(cases), "cases")


def val_match_fixtures():
    """Known answers of the validator's matching step: the reference's own box_iou (utils/metrics.py:52-70) and
    BaseValidator.match_predictions (engine/validator.py:222-258) through DetectionValidator._process_batch
    (models/yolo/detect/val.py:209-228) on synthetic detections / labels (continuous random boxes: no exact IoU ties)."""
    from ultralytics.engine.validator import BaseValidator
    from ultralytics.utils.metrics import box_iou
    v = object.__new__(BaseValidator)
    v.iouv = torch.linspace(0.5, 0.95, 10)
    rng = np.random.default_rng(77)
    out, cases = {}, []
    for ci, (nd, nl, ncls, jitter) in enumerate([(40, 12, 3, 4.0), (300, 60, 80, 6.0), (7, 1, 1, 2.0), (1, 9, 2, 3.0), (120, 120, 5, 1.0),
                                                 (25, 0, 3, 1.0), (0, 5, 3, 1.0), (64, 30, 1, 12.0)]):
        lab = np.zeros((nl, 4), np.float32)
        if nl:
            xy = rng.uniform(0, 560, (nl, 2)); wh = rng.uniform(12, 120, (nl, 2))
            lab = np.concatenate([xy, xy + wh], 1).astype(np.float32)
        lcls = rng.integers(0, ncls, nl).astype(np.float32)
        det = np.zeros((nd, 6), np.float32)
        if nd:
            if nl:  # most detections are jittered copies of labels (several per label), the rest random
                src = rng.integers(0, nl, nd)
                det[:, :4] = lab[src] + rng.normal(0, jitter, (nd, 4)).astype(np.float32)
                det[:, 5] = np.where(rng.random(nd) < 0.8, lcls[src], rng.integers(0, ncls, nd))
                rnd = rng.random(nd) < 0.2
                xy = rng.uniform(0, 560, (nd, 2)); wh = rng.uniform(12, 120, (nd, 2))
                det[rnd, :4] = np.concatenate([xy, xy + wh], 1).astype(np.float32)[rnd]
            else:
                xy = rng.uniform(0, 560, (nd, 2)); wh = rng.uniform(12, 120, (nd, 2))
                det[:, :4] = np.concatenate([xy, xy + wh], 1)
                det[:, 5] = rng.integers(0, ncls, nd)
            det[:, 4] = np.sort(rng.uniform(0.25, 1.0, nd))[::-1]
        d, l, c = torch.from_numpy(det), torch.from_numpy(lab), torch.from_numpy(lcls)
        iou = box_iou(l, d[:, :4])
        correct = v.match_predictions(d[:, 5], c, iou) if nl and nd else torch.zeros(nd, 10, dtype=torch.bool)
        out[f"c{ci}.det"], out[f"c{ci}.lab"], out[f"c{ci}.lcls"] = det, lab, lcls
        out[f"c{ci}.iou"], out[f"c{ci}.correct"] = iou.numpy(), correct.numpy()
        cases.append(ci)
    out["cases"] = json.dumps(cases)
    np.savez_compressed(HERE / "val_match.npz", **out)
    print("wrote val_match", len(cases), "cases")


def c3_fixture():
    """Known answers of the reference's C3 module (block.py:3320-3334; YOLOv5u), with and without shortcut."""
    from ultralytics.utils.torch_utils import fuse_conv_and_bn
    out, cases = {}, {}
    g = torch.Generator().manual_seed(12)
    for tag, mod, x, ctor in (("c3", rb.C3(32, 32, 2), torch.randn(2, 32, 9, 8, generator=g), ["C3", 32, 32, 2]),
                              ("c3_noadd", rb.C3(48, 32, 1, False), torch.randn(1, 48, 6, 7, generator=g), ["C3", 48, 32, 1, False])):
        mod.eval()
        initialize_weights(mod)
        fill(mod, "m.", 13)
        for sm in mod.modules():
            if isinstance(sm, rc.Conv) and hasattr(sm, "bn"):
                sm.conv = fuse_conv_and_bn(sm.conv, sm.bn)
                delattr(sm, "bn")
                sm.forward = sm.forward_fuse
        with torch.inference_mode():
            y = mod(x)
        out[tag + ".x"], out[tag + ".y"] = x.numpy(), y.numpy()
        cases[tag] = ctor
    out["cases"] = json.dumps(cases)
    np.savez_compressed(HERE / "modules_c3.npz", **out)
    print("wrote modules_c3", list(cases))


def masks_native_fixtures():
    """Known answers of the retina_masks path: the reference's own process_mask_native and scale_masks (utils/ops.py:696-737)
    on synthetic prototypes / coefficients / boxes; original-image shapes with letterbox padding on either axis, none, and
    a down-scaling case."""
    out, cases = {}, []
    g = torch.Generator().manual_seed(91)
    for ci, (nm, mh, mw, n, shape) in enumerate([(32, 40, 40, 6, (120, 160)), (32, 40, 40, 5, (160, 90)), (16, 24, 32, 3, (96, 128)),
                                                 (8, 80, 80, 4, (200, 320)), (8, 20, 20, 2, (15, 17)), (32, 40, 40, 0, (64, 64))]):
        protos = torch.randn(nm, mh, mw, generator=g)
        coef = torch.randn(n, nm, generator=g) * 0.5
        xy = torch.rand(n, 2, generator=g) * torch.tensor([shape[1] * 0.6, shape[0] * 0.6])
        wh = torch.rand(n, 2, generator=g) * torch.tensor([shape[1] * 0.5, shape[0] * 0.5]) + 3.0
        boxes = torch.cat([xy, xy + wh], 1)
        with torch.inference_mode():
            native = rops.process_mask_native(protos, coef, boxes, shape) if n else torch.zeros(0, *shape)
            scaled = rops.scale_masks(protos[None, :2], shape)  # two channels: the arithmetic is per channel
            scaled_np = rops.scale_masks(protos[None, :2], shape, padding=False)
        out[f"c{ci}.protos"], out[f"c{ci}.coef"], out[f"c{ci}.boxes"] = protos.numpy(), coef.numpy(), boxes.numpy()
        out[f"c{ci}.native"] = native.numpy().astype(np.uint8)
        out[f"c{ci}.scaled"], out[f"c{ci}.scaled_nopad"] = scaled.numpy(), scaled_np.numpy()
        cases.append({"ci": ci, "shape": list(shape)})
    out["cases"] = json.dumps(cases)
    np.savez_compressed(HERE / "masks_native.npz", **out)
    print("wrote masks_native", len(cases), "cases")


def synth_pred(b, nc, a, nm, seed, peaky, dtype=torch.float32):
    """(B, 4+nc+nm, A) prediction tensor in Detect's output format with duplicated / overlapping boxes."""
    g = torch.Generator().manual_seed(seed)
    base = max(a // 8, 1)
    cx = torch.rand(b, base, generator=g) * 600 + 20
    cy = torch.rand(b, base, generator=g) * 600 + 20
    w = torch.rand(b, base, generator=g) * 150 + 10
    h = torch.rand(b, base, generator=g) * 150 + 10
    rep = (a + base - 1) // base
    jit = lambda t: (t.repeat(1, rep)[:, :a] + torch.randn(b, a, generator=g) * 4.0)  # noqa: E731
    box = torch.stack((jit(cx), jit(cy), jit(w).abs() + 2, jit(h).abs() + 2), 1)
    if peaky:
        cls = torch.rand(b, nc, a, generator=g) ** 8
    else:
        cls = torch.rand(b, nc, a, generator=g) * 0.6
    parts = [box, cls]
    if nm:
        parts.append(torch.randn(b, nm, a, generator=g))
    return torch.cat(parts, 1).to(dtype)


def nms_fixtures():
    out = {}
    cfgs = []
    cases = [
        ("predict_peaky", dict(b=3, nc=80, a=1050, nm=0, seed=1, peaky=True), dict(conf_thres=0.25, iou_thres=0.7)),
        ("predict_flat", dict(b=2, nc=12, a=640, nm=0, seed=2, peaky=False), dict(conf_thres=0.25, iou_thres=0.45)),
        ("val_multilabel", dict(b=2, nc=20, a=1000, nm=0, seed=3, peaky=True),
         dict(conf_thres=0.001, iou_thres=0.7, multi_label=True)),
        ("val_cap", dict(b=1, nc=40, a=2000, nm=0, seed=4, peaky=False),
         dict(conf_thres=0.001, iou_thres=0.7, multi_label=True, max_nms=3000, max_det=300)),
        ("agnostic", dict(b=2, nc=8, a=512, nm=0, seed=5, peaky=False),
         dict(conf_thres=0.3, iou_thres=0.5, agnostic=True, max_det=50)),
        ("classes", dict(b=2, nc=8, a=512, nm=0, seed=6, peaky=False),
         dict(conf_thres=0.3, iou_thres=0.5, classes=[1, 3, 6])),
        ("seg_masks", dict(b=2, nc=10, a=700, nm=32, seed=7, peaky=True), dict(conf_thres=0.25, iou_thres=0.7, nc=10)),
        ("empty", dict(b=2, nc=10, a=300, nm=0, seed=8, peaky=True), dict(conf_thres=0.9999, iou_thres=0.7)),
        ("single_class", dict(b=2, nc=1, a=400, nm=0, seed=9, peaky=False),
         dict(conf_thres=0.3, iou_thres=0.6, multi_label=True)),
    ]
    for tag, pk, kw in cases:
        pred = synth_pred(**pk)
        inp = pred.clone()
        res = rops.non_max_suppression(pred, **kw)
        out[tag + ".pred"] = inp.numpy()
        out[tag + ".pred_after"] = pred[:, :4].numpy()  # in_place xywh->xyxy mutation (ops.py:243-244)
        for i, r in enumerate(res):
            out[f"{tag}.out{i}"] = r.numpy()
        cfgs.append((tag, kw, len(res)))
    out["cases"] = json.dumps(cfgs)

    # scale_boxes / clip_boxes (ops.py:92-127, :319)
    sb = []
    g = torch.Generator().manual_seed(3)
    for i, (s1, s0) in enumerate([((640, 480), (1080, 810)), ((384, 640), (720, 1280)), ((640, 640), (4000, 6000)),
                                  ((640, 640), (333, 500)), ((1280, 1280), (100, 37))]):
        boxes = torch.rand(50, 4, generator=g) * 700 - 30
        res = rops.scale_boxes(s1, boxes.clone(), s0)
        out[f"scale_boxes{i}.in"] = boxes.numpy()
        out[f"scale_boxes{i}.out"] = res.numpy()
        sb.append((s1, s0))
    out["scale_boxes_cases"] = json.dumps(sb)
    np.savez_compressed(HERE / "nms.npz", **out)
    print("wrote nms", [c[0] for c in cfgs])


def letterbox_fixtures():
    out = {}
    cases = []
    rng = np.random.default_rng(0)
    shapes = [(1080, 810), (720, 1280), (4000, 6000), (333, 500), (640, 640), (100, 37), (1280, 1280), (641, 959),
              (480, 640)]
    # geometry: output shape and where the image lands, for the shapes SURVEY 8c lists (no pixels stored)
    for (h, w) in shapes:
        for kw in (dict(new_shape=(640, 640), auto=True), dict(new_shape=(640, 640), auto=False),
                   dict(new_shape=(1280, 1280), auto=False), dict(new_shape=(384, 640), auto=False, scaleup=False),
                   dict(new_shape=(640, 640), auto=False, center=False)):
            marker = np.full((h, w, 3), 7, np.uint8)
            r2 = LetterBox(stride=32, **kw)(image=marker)
            ys, xs = np.where(r2[..., 0] != 114)
            cases.append({"shape": [h, w], "kw": kw, "out_shape": list(r2.shape),
                          "box": [int(ys.min()), int(ys.max()) + 1, int(xs.min()), int(xs.max()) + 1]})
    out["cases"] = json.dumps(cases)
    # pixels (small): pinned only against the oracle's own cv2.resize restatement (see module docstring)
    pix = []
    for k, ((h, w), kw) in enumerate([((100, 37), dict(new_shape=(160, 160), auto=False)),
                                      ((75, 120), dict(new_shape=(96, 128), auto=True)),
                                      ((333, 500), dict(new_shape=(160, 224), auto=False)),
                                      ((64, 64), dict(new_shape=(128, 128), auto=False)),
                                      ((256, 192), dict(new_shape=(128, 128), auto=True))]):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        out[f"img{k}"] = img
        out[f"lb{k}"] = LetterBox(stride=32, **kw)(image=img)
        pix.append({"kw": kw})
    out["pixel_cases"] = json.dumps(pix)
    np.savez_compressed(HERE / "letterbox.npz", **out)
    print("wrote letterbox", len(cases), "geometry cases,", len(pix), "pixel cases")


def ap_fixtures():
    """Known answers of the reference's ap_per_class (utils/metrics.py:620-706, with compute_ap :588-617) on synthetic
    detection statistics: tp (N, 10) bool, conf (N) float32 (all distinct: numpy's unstable argsort leaves ties unpinned),
    pred_cls (N), target_cls (M).  Cases cover classes without predictions / without labels and a single detection."""
    from ultralytics.utils.metrics import ap_per_class
    rng = np.random.default_rng(5)
    out, cases = {}, []
    for ci, (n, m, ncls, hit) in enumerate([(400, 120, 5, 0.7), (5000, 900, 80, 0.5), (1, 3, 2, 1.0), (60, 40, 12, 0.2), (3000, 500, 3, 0.9)]):
        conf = rng.permutation(np.linspace(0.05, 0.999, n)).astype(np.float32)
        pred_cls = rng.integers(0, ncls, n).astype(np.float32)
        target_cls = rng.integers(0, ncls + 2, m).astype(np.float32)   # two classes have labels but no predictions
        if ci == 3:
            pred_cls[pred_cls == 4] = 30.0                              # a predicted class without labels
        base = rng.random(n) < hit * conf                               # better-scored detections are right more often
        tp = np.stack([base & (rng.random(n) < 1.0 - 0.08 * j) for j in range(10)], 1)
        tp = np.logical_and.accumulate(tp, 1)                           # a hit at IoU t is a hit at every lower threshold
        for c in np.unique(pred_cls):                                   # a label is matched at most once: <= n_l hits per class
            n_l = int((target_cls == c).sum())
            idx = np.nonzero(pred_cls == c)[0]
            idx = idx[np.argsort(-conf[idx])]
            for j in range(10):
                hits = idx[tp[idx, j]]
                tp[hits[n_l:], j] = False
        r = ap_per_class(tp, conf, pred_cls, target_cls, plot=False)
        names = ["tp", "fp", "p", "r", "f1", "ap", "unique_classes", "p_curve", "r_curve", "f1_curve", "x", "prec_values"]
        for k, v in zip(names, r):
            out[f"c{ci}.out.{k}"] = np.asarray(v)
        out[f"c{ci}.tp"], out[f"c{ci}.conf"], out[f"c{ci}.pred_cls"], out[f"c{ci}.target_cls"] = tp, conf, pred_cls, target_cls
        cases.append(ci)
    out["cases"] = json.dumps(cases)
    np.savez_compressed(HERE / "ap_per_class.npz", **out)
    print("wrote ap_per_class", len(cases), "cases")


def config1_fixture():
    """BASELINE config 1 on the reference's own test image (ultralytics/assets/bus.jpg, the ASSETS image of the reference's tests,
    tests/__init__.py:11): pixels -> the reference's LetterBox (predictor.pre_transform's settings: auto=True for a pt model,
    engine/predictor.py:155-160) -> BGR->RGB, /255 (predictor.py:125-133) -> the reference's YOLO11n DetectionModel (fused, fp32,
    CPU) -> the reference's non_max_suppression (predict defaults conf 0.25, iou 0.7, cfg/default.yaml) -> scale_boxes to the
    original image (models/yolo/detect/predict.py:23-41).  The JPEG is decoded ONCE here with PIL (cv2 is not in this image;
    decoders may differ by +-1 on isolated pixels, which is upstream of the path) and the u8 BGR array is the fixture's input.
    The class head's bias is shifted so that 2 % of the anchors pass conf 0.25 on this image (random weights)."""
    from PIL import Image
    bgr = np.ascontiguousarray(np.asarray(Image.open("/root/reference/ultralytics/assets/bus.jpg").convert("RGB"))[:, :, ::-1])
    assert bgr.shape == (1080, 810, 3)
    m, names, _ = build_ref("yolo11", "n", "detect", 80, 0)
    lb = LetterBox((640, 640), auto=True, stride=32)(image=bgr)
    assert lb.shape == (640, 480, 3)
    x = torch.from_numpy(np.ascontiguousarray(lb[None][..., ::-1].transpose(0, 3, 1, 2))).float() / 255.0
    det = m.model[-1]
    with torch.inference_mode():
        _, raw = m(x)
        top = torch.cat([r[:, 64:].flatten(2) for r in raw], 2).amax(1).flatten()
        shift = float(np.log(0.25 / 0.75) - torch.quantile(top, 0.98))
        for seq in det.cv3:
            seq[-1].bias.add_(shift)
        y, raw = m(x)
        pred = rops.non_max_suppression(y.clone(), 0.25, 0.7, max_det=300)[0]
        boxes = rops.scale_boxes(x.shape[2:], pred[:, :4].clone(), bgr.shape)
    # y is (1, 84, 6300): the fixture keeps the 1000 highest-scoring anchors (every NMS candidate and a wide margin below conf) and,
    # for the letterboxed pixels, shape + crc32 (the HIP letterbox is compared bit for bit with the oracle's, whose crc is this one)
    import zlib
    idx = torch.argsort(y[0, 4:].amax(0), descending=True)[:1000].sort().values
    out = {"meta": json.dumps({"family": "yolo11", "scale": "n", "nc": 80, "seed": 0, "cls_shift": shift, "conf": 0.25, "iou": 0.7,
                               "n_det": int(pred.shape[0]), "lb_shape": list(lb.shape), "lb_crc32": zlib.crc32(np.ascontiguousarray(lb).tobytes()),
                               "y_shape": list(y.shape)}),
           "bgr": bgr, "y_idx": idx.numpy(), "y_top": y[0][:, idx].numpy(), "pred": pred.numpy(), "boxes": boxes.numpy()}
    np.savez_compressed(HERE / "config1_bus.npz", **out)
    print("wrote config1_bus:", pred.shape[0], "detections, letterbox", lb.shape, "shift", shift)


if __name__ == "__main__":
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "config1":  # BASELINE config 1 on bus.jpg (round 4)
        config1_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "val":  # only the validator-matching vectors
        val_match_fixtures()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "yolov5":  # YOLOv5u graph + the C3 module (round 2)
        graph_fixture("yolov5n_detect", "yolov5", "n", "detect", [(2, 64, 64), (1, 96, 160)], keep_layers=True)
        graph_fixture("yolov5s_detect", "yolov5", "s", "detect", [(1, 64, 96)])
        c3_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "masks_native":  # only the retina-masks vectors (round 2)
        masks_native_fixtures()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ap":  # only the ap_per_class vectors
        ap_fixtures()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "bsyolo_modules":  # only the BS-YOLO module vectors (cases are appended: earlier ones keep their values)
        bsyolo_module_fixtures()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "bsyolo":  # only the BS-YOLO graph vectors (added after the stock set)
        graph_fixture("bsyolo11n_detect", "bsyolo11", "n", "detect", [(2, 64, 64), (1, 96, 160)], nc=12, keep_layers=True)
        graph_fixture("bsyolo11s_detect", "bsyolo11", "s", "detect", [(1, 64, 96)], nc=12)
        bsyolo_module_fixtures()
        sys.exit(0)
    graph_fixture("yolo11n_detect", "yolo11", "n", "detect", [(2, 64, 64), (1, 96, 160)], keep_layers=True)
    graph_fixture("yolo11s_detect", "yolo11", "s", "detect", [(1, 64, 96)])
    graph_fixture("yolo11m_detect", "yolo11", "m", "detect", [(1, 64, 64)])
    graph_fixture("yolo11n_segment", "yolo11", "n", "segment", [(1, 64, 64)])
    graph_fixture("yolov8n_segment", "yolov8", "n", "segment", [(1, 64, 96)])
    module_fixtures()
    nms_fixtures()
    letterbox_fixtures()
    val_match_fixtures()
    config1_fixture()
