"""Reference-side installation: make an unmodified Ultralytics/BS-YOLO model dispatch its hot path to libbsyolo_hip.so.

Three hooks, each using an extension point the reference already has:

  accelerate(model)        whole-graph hook.  Rebinds ``model.forward`` on the instance -- the reference's own idiom
                           (``m.forward = m.forward_fuse``, nn/tasks.py:215) -- so that AutoBackend's in-memory branch
                           (nn/autobackend.py:136-147, call at :524) reaches the engine.  The graph comes from
                           ``model.yaml`` (the dict parse_model consumed, nn/tasks.py:313-321) and the weights from
                           ``model.state_dict()``.  Anything the engine does not cover (training mode, augment / visualize /
                           embed (tasks.py:134-164), CPU tensors, unsupported modules) goes to the original forward.
  install(model)           per-module hook (the fallback level for graphs `accelerate` does not cover: a custom block, an
                           unknown head).  Rebinds ``forward`` on every ``Conv`` / ``DWConv`` INSTANCE (nn/modules/conv.py:133-151,
                           :224-229) -- again the reference's own idiom, ``m.forward = m.forward_fuse`` (nn/tasks.py:215) -- so
                           that the block modules' Python forwards (C2f, C3k, C2PSA, PMSFA ... call their child Convs) run
                           every convolution through ``bsy_conv2d`` / ``bsy_conv_first`` / ``bsy_dwconv``; and on the block
                           instances the library has an operator for: ``SPPF`` (block.py:3145-3149: ``bsy_sppf_pool`` on the
                           concat buffer), ``Attention`` (:4267-4286: ``bsy_attention`` + the ``pe`` depthwise conv with the
                           residual), ``Bottleneck`` / ``C3k2`` whose widths the fused kernels are built for (:3417-3419,
                           :3796-3804: ``bsy_bottleneck_fused`` / ``bsy_c3k2_fused``) and ``Detect._inference``
                           (head.py:100-131: ``bsy_detect_decode``).  Activations stay torch tensors in channels_last memory
                           format (= the kernels' NHWC, no copies); the rest (cat, chunk, upsample ...) stays with torch.
  install_nms(ops_module)  ``ultralytics.utils.ops.non_max_suppression`` is looked up as a module attribute on every
                           call (models/yolo/detect/predict.py:25, detect/val.py:95): replacing the attribute suffices.
  install_masks(ops_module) same for ``ultralytics.utils.ops.process_mask`` (models/yolo/segment/predict.py:53).
  install_val_metrics(v)   rebinds ``DetectionValidator._process_batch`` (models/yolo/detect/val.py:209-228) on a validator
                           instance: true-positive matching on the device.
  install_preprocess(p)    rebinds ``BasePredictor.preprocess`` (engine/predictor.py:116-134) on a predictor instance so
                           lists of HWC BGR uint8 images are letterboxed on the device.
"""
from __future__ import annotations

import types
from typing import Optional

import torch

from . import lib as _L
from . import nms as _nms
from .engine import YoloEngine
from .plan import Plan
from .weights import BN_EPS

SUPPORTED_HEADS = ("Detect", "Segment")


def _on_device(t) -> bool:
    """The hooks' dispatch test: the library runs tensors that live on a ROCm device; everything else goes to the reference's own
    code.  One function so that the contract test (tests/test_host_logic.py, build container, no GPU) can let CPU tensors reach
    RECORDING stand-ins for the engine / NMS / letterbox calls while the reference's predictor drives the hooks -- the product never
    rebinds it, and the library itself has no CPU path (bs_yolo_amd.lib raises without the .so, YoloEngine without a GPU)."""
    return bool(t.is_cuda)


def _device_is_gpu(device) -> bool:
    return str(device).startswith("cuda")


def model_bn_eps(model) -> float:
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            return float(m.eps)
    return BN_EPS


def cfg_of(model) -> dict:
    """The yaml dict of a reference model (DetectionModel.yaml, nn/tasks.py:313) -- raises if the graph uses modules
    outside the accelerated set, so callers can fall back."""
    cfg = getattr(model, "yaml", None)
    if not isinstance(cfg, dict) or "backbone" not in cfg or "head" not in cfg:
        raise NotImplementedError("model has no yaml graph description")
    try:
        Plan(cfg, 1, 64, 64)  # raises NotImplementedError / AssertionError on unsupported graphs
    except NotImplementedError:
        # widths only the fp32 modes take (BS-YOLO's PMSFA on a width that is not a multiple of 8, e.g. a width multiple of 0.1875): fp32
        # callers still get the engine, fp16 calls fall back to the reference forward when their engine is built
        Plan(cfg, 1, 64, 64, precision="fp32")
    return cfg


def _weights_version(model) -> tuple:
    """Cheap fingerprint of the parameters and buffers: in-place updates (optimizer steps, EMA, load_state_dict's copy_)
    bump ``Tensor._version``; ``.half()`` / ``.to()`` / ``fuse()`` replace the storages (data_ptr).  Inference tensors
    -- what ``fuse()`` creates under the predictor's / validator's ``smart_inference_mode`` (engine/predictor.py:219,
    engine/validator.py:105) -- carry no version counter (reading it raises) and cannot be updated in place outside
    inference mode, so their storage pointer is the whole fingerprint."""
    n, ver, ptr = 0, 0, 0
    for t in list(model.parameters()) + list(model.buffers()):
        n += 1
        try:  # (`is_inference()` alone does not tell: a Parameter whose .data was swapped by .half() / .to() outside inference
            ver += t._version  # mode reports False and still has no counter)
        except RuntimeError:
            pass
        ptr ^= t.data_ptr() + 0x9E3779B1 * n
    return (n, ver, ptr)


def graph_support(cfg: dict) -> dict:
    """Which engine precisions can run this yaml graph: {"fp16": bool, "fp32": bool} (the fp32 / fp32x modes also take PMSFA widths
    that are not multiples of 8 channels, e.g. BS-YOLO at a width multiple of 0.1875).  Decided once, on the host, from the plan alone."""
    out = {}
    for prec in ("fp16", "fp32"):
        try:
            Plan(cfg, 1, 64, 64, precision=prec)
            out[prec] = True
        except (NotImplementedError, AssertionError):
            out[prec] = False
    return out


FP32_MODES = {"engine_fp32x": "fp32x", "engine_fp32": "fp32", "engine": "fp16", "reference": None}


def accelerate(model, device: Optional[int] = None, verbose: bool = False, fp32_inputs: Optional[str] = None):
    """Install the engine behind ``model.forward``.  Returns the same model object.

    fp32_inputs: what an fp32 image tensor gets (the reference's ``predict()`` default is ``half=False``,
    engine/predictor.py:131).  The product path stores activations in fp16, which is NOT the fp32 model's arithmetic
    (max |dscore| up to 6e-3 against the fp32 reference, vs the north-star's 1e-3), so it is never
    applied to fp32 callers silently:
      ``"engine_fp32x"`` (default) the engine's fp32x mode: fp32 storage, dense convs on the fp16 matrix pipe with both operands
                                   split into f16 pairs (csrc/conv32x_mfma.hip; |dscore| ~1e-5, |dbox| ~1e-5 * imgsz against the
                                   fp32 reference -- the north-star's 1e-3 with two orders of margin -- at 2-3x the exact mode's
                                   throughput; a one-time log line says which mode runs);
      ``"engine_fp32"``            the exact fp32 mode: convs on the fp32 matrix pipe (csrc/conv32_mfma.hip: exact f32 products,
                                   k-ordered f32 sums, bit-reproducible against the scalar kernels);
      ``"engine"``                 opt in to the fast fp16-storage path for fp32 inputs (outputs come back as fp32);
      ``"reference"``              leave fp32 inputs to the original forward.
    ``BSY_FP32_INPUTS`` overrides the default.  fp16 inputs (``half=True`` / ``model.half()``) always run on the fp16
    engine: there the reference itself computes in fp16.

    Graphs only some precisions can run (``graph_support``): calls in a precision the engine does not take go straight to the
    reference forward -- decided here, once, not by a failing engine build on every call."""
    import os
    cfg = cfg_of(model)
    support = graph_support(cfg)
    orig_forward = model.forward
    state = {"engine": None, "engine32": None, "key": None, "fallbacks": 0, "engine_calls": 0, "fp32_calls": 0, "rebuilds": 0,
             "support": support, "failed": set()}
    fp32_mode = fp32_inputs or os.environ.get("BSY_FP32_INPUTS", "engine_fp32x")
    if fp32_mode not in FP32_MODES:
        raise ValueError(f"fp32_inputs must be one of {sorted(FP32_MODES)}, not {fp32_mode!r}")
    prec32 = FP32_MODES[fp32_mode]  # engine precision fp32 images run in (None: the reference forward)

    def _engine_for(dev: torch.device, precision: str = "fp16") -> YoloEngine:
        key = (dev.index or 0, _weights_version(model))
        if state["key"] != key:
            for k in ("engine", "engine32"):
                if state[k] is not None:
                    state[k].close()
                    state[k] = None
            state["key"] = key
            state["failed"].clear()
        slot = "engine" if precision == "fp16" else "engine32"
        if state[slot] is None:
            # weights are read at this moment: after AutoBackend has called fuse()/half() (autobackend.py:139-145); the key
            # makes a later in-place update (EMA, load_state_dict, a training step) or a .half()/.to() rebuild the engine
            sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
            state["rebuilds"] += 1
            state[slot] = YoloEngine(cfg, sd, device=dev.index or 0, bn_eps=model_bn_eps(model), precision=precision)
        return state[slot]

    def forward(self, x, *args, **kwargs):
        augment = kwargs.get("augment", False)
        visualize = kwargs.get("visualize", False)
        embed = kwargs.get("embed", None)
        profile = kwargs.get("profile", False)
        if (self.training or augment or visualize or embed or profile or args or not isinstance(x, torch.Tensor)
                or not _on_device(x) or x.dim() != 4 or x.dtype not in (torch.float16, torch.float32)
                or x.shape[1] != 3 or x.shape[2] % 32 or x.shape[3] % 32 or x.shape[0] == 0):
            state["fallbacks"] += 1
            return orig_forward(x, *args, **kwargs)
        precision = "fp16" if x.dtype == torch.float16 else prec32
        # a precision this graph cannot run in (or whose engine build has already failed for these weights): the reference runs it,
        # without another device-to-host weight copy + engine create / destroy per call
        if precision is None or not support["fp16" if precision == "fp16" else "fp32"] or precision in state["failed"]:
            state["fallbacks"] += 1
            return orig_forward(x, *args, **kwargs)
        want32 = precision != "fp16"
        if x.dtype == torch.float32 and not state.get("fp32_notice"):
            state["fp32_notice"] = True  # once per model: which engine default callers (predict(half=False)) are on, and the way out
            import logging
            logging.getLogger("bs_yolo_amd").warning(
                "fp32 images run in the engine's %s mode (%s).  fp32_inputs / BSY_FP32_INPUTS selects: engine_fp32x (default: fp32 storage, "
                "split-f16 matrix products, |dscore| ~1e-5 vs the fp32 model), engine_fp32 (exact fp32 arithmetic, 2-3x slower), engine "
                "(the fp16 product path, ~3x faster again, |dscore| up to 6e-3), reference (the original forward); predict(half=True) "
                "runs the fp16 path.", precision, fp32_mode)
        try:
            y, raws = _engine_for(x.device, precision)(x)
        except (_L.BsyError, NotImplementedError, AssertionError) as e:  # a shape / graph the engine rejects: the reference runs it
            if verbose:
                print(f"bs_yolo_amd: falling back to the reference forward ({e})")
            if state["engine" if precision == "fp16" else "engine32"] is None:
                state["failed"].add(precision)  # the BUILD failed: do not try again until the weights change
            state["fallbacks"] += 1
            return orig_forward(x, *args, **kwargs)
        state["fp32_calls" if want32 else "engine_calls"] += 1
        return y, raws

    model.forward = types.MethodType(forward, model)
    model._bsy_state = state
    model._bsy_orig_forward = orig_forward
    return model


def restore(model):
    if hasattr(model, "_bsy_orig_forward"):
        model.forward = model._bsy_orig_forward
        st = model._bsy_state
        for k in ("engine", "engine32"):
            if st.get(k) is not None:
                st[k].close()
        del model._bsy_orig_forward, model._bsy_state
    return model


# ---------------------------------------------------------------------------------------------------------------------
# per-module hook
# ---------------------------------------------------------------------------------------------------------------------
def _conv_spec(m):
    """(kind, k, s, act) if `m` is a reference Conv / DWConv instance the kernels cover, else None."""
    conv, act = getattr(m, "conv", None), getattr(m, "act", None)
    if not isinstance(conv, torch.nn.Conv2d) or act is None or type(m).__name__ not in ("Conv", "DWConv"):
        return None
    if isinstance(act, torch.nn.SiLU):
        a = 1
    elif isinstance(act, torch.nn.Identity):
        a = 0
    else:
        return None
    (kh, kw), (sh, sw), (ph, pw) = conv.kernel_size, conv.stride, conv.padding
    if kh != kw or sh != sw or conv.dilation != (1, 1) or (ph, pw) != (kh // 2, kw // 2) or conv.padding_mode != "zeros":
        return None
    c1, c2, g = conv.in_channels, conv.out_channels, conv.groups
    if g == 1 and kh in (1, 3) and sh in (1, 2) and c1 % 8 == 0 and c2 % 8 == 0:
        return ("conv", kh, sh, a)
    if g == 1 and c1 == 3 and (kh, sh) == (3, 2) and c2 % 8 == 0:
        return ("first", kh, sh, a)
    if g == c1 == c2 and kh % 2 == 1 and kh <= 31 and sh in (1, 2) and c1 % 8 == 0:
        return ("dw", kh, sh, a)
    return None


def _conv_forward(m, spec):
    """The replacement forward of one Conv / DWConv instance.  Falls back to the module's own forward for anything but a
    fp16 CUDA NCHW tensor in eval mode; re-packs its weights when they change (``Tensor._version`` / storage)."""
    from .ops import _p, _stream, pack_conv_weight
    from .weights import fold_conv_bn
    import ctypes as C
    kind, k, s, act = spec
    orig = m.forward
    st = {"ver": None, "dev": None, "w": None, "b": None, "calls": 0, "fallbacks": 0}

    def pack(dev):
        sd = {"m.conv." + n: t for n, t in m.conv.state_dict().items()}
        bn = getattr(m, "bn", None)
        if st["fold_bn"]:
            sd.update({"m.bn." + n: t for n, t in bn.state_dict().items()})
        elif "m.conv.bias" not in sd:
            sd["m.conv.bias"] = torch.zeros(m.conv.out_channels)
        w, b = fold_conv_bn(sd, "m", float(bn.eps) if isinstance(bn, torch.nn.BatchNorm2d) else BN_EPS)
        if kind == "dw":
            c = m.conv.out_channels
            st["w"] = w.view(c, k * k).t().contiguous().to(dev)
            st["b"] = b.contiguous().to(dev)
        else:
            st["w"], st["b"] = pack_conv_weight(w, b, dev)

    def fwd(self, x):
        if (not torch.is_tensor(x) or not x.is_cuda or x.dtype != torch.float16 or x.dim() != 4 or self.training
                or x.shape[1] != self.conv.in_channels):
            st["fallbacks"] += 1
            return orig(x)
        ver = _weights_version(self)
        if st["ver"] != ver or st["dev"] != x.device:
            pack(x.device)
            st["ver"], st["dev"] = ver, x.device
        B, c1, H, W = x.shape
        c2 = self.conv.out_channels
        OH, OW = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
        out = torch.empty((B, OH, OW, c2), dtype=torch.float16, device=x.device)
        with torch.cuda.device(x.device):
            if kind == "first":
                xi = x.contiguous()
                _L.check(_L.lib.bsy_conv_first(_p(xi), _L.dtype_code(xi.dtype), B, H, W, _p(st["w"]), _p(st["b"]), _p(out), c2, c2, k, s,
                                               act, _stream(x)))
            else:
                xn = x.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)  # NHWC view, no copy when already channels_last
                if kind == "conv":
                    _L.check(_L.lib.bsy_conv2d(_p(xn), c1, B, H, W, c1, _p(st["w"]), _p(st["b"]), _p(out), c2, c2, k, s, act, None, 0, 0,
                                               _stream(x)))
                else:
                    _L.check(_L.lib.bsy_dwconv(_p(xn), c1, B, H, W, c1, k, k, s, _p(st["w"]), c1, _p(st["b"]), _p(out), c2, act, _stream(x)))
        st["calls"] += 1
        return out.permute(0, 3, 1, 2)  # NCHW tensor in channels_last memory format

    # un-fused module (its forward = conv -> bn -> act): BatchNorm is folded at pack time; after BaseModel.fuse() (tasks.py:209-215:
    # conv replaced by the fused conv, `bn` deleted, forward = forward_fuse) conv.bias already holds it
    st["fold_bn"] = isinstance(getattr(m, "bn", None), torch.nn.BatchNorm2d) and getattr(orig, "__name__", "") != "forward_fuse"
    return fwd, st


def _silu_conv(m, k=None, s=1, groups=1) -> bool:
    """`m` is a reference Conv (+BN) + SiLU module with a square k x k kernel (k None: any), stride s, "same" padding."""
    conv, act = getattr(m, "conv", None), getattr(m, "act", None)
    return (isinstance(conv, torch.nn.Conv2d) and isinstance(act, torch.nn.SiLU) and conv.groups == groups
            and conv.kernel_size[0] == conv.kernel_size[1] and (k is None or conv.kernel_size[0] == k)
            and conv.stride == (s, s) and conv.padding == (conv.kernel_size[0] // 2,) * 2 and conv.dilation == (1, 1))


def _folded(m):
    """(weight, bias) fp32 of a reference Conv module, BatchNorm folded in when the module still has one (un-fused)."""
    from .weights import fold_conv_bn
    sd = {"m.conv." + n: t for n, t in m.conv.state_dict().items()}
    bn = getattr(m, "bn", None)
    if isinstance(bn, torch.nn.BatchNorm2d) and getattr(m.forward, "__name__", "") != "forward_fuse":
        sd.update({"m.bn." + n: t for n, t in bn.state_dict().items()})
    elif "m.conv.bias" not in sd:
        sd["m.conv.bias"] = torch.zeros(m.conv.out_channels)
    return fold_conv_bn(sd, "m", float(bn.eps) if isinstance(bn, torch.nn.BatchNorm2d) else BN_EPS)


def _block_spec(m):
    """Which block operator of the library covers `m` (a reference module instance), or None."""
    from . import lib as L
    name = type(m).__name__
    if name == "SPPF":
        pool = getattr(m, "m", None)
        if (isinstance(pool, torch.nn.MaxPool2d) and pool.kernel_size == 5 and pool.stride == 1 and pool.padding == 2
                and _silu_conv(getattr(m, "cv1", None), 1) and _silu_conv(getattr(m, "cv2", None), 1)
                and m.cv1.conv.out_channels % 8 == 0 and m.cv2.conv.in_channels == 4 * m.cv1.conv.out_channels):
            return "sppf"
    if name == "Attention":
        if (all(hasattr(m, a) for a in ("qkv", "proj", "pe", "num_heads", "key_dim", "head_dim", "scale"))
                and m.key_dim == 32 and m.head_dim == 64 and _conv_spec(m.qkv) == ("conv", 1, 1, 0) and _conv_spec(m.proj) == ("conv", 1, 1, 0)
                and _conv_spec(m.pe) == ("dw", 3, 1, 0)):
            return "attention"
    if name == "Bottleneck":
        cv1, cv2 = getattr(m, "cv1", None), getattr(m, "cv2", None)
        if (getattr(m, "add", False) and _silu_conv(cv1, 3) and _silu_conv(cv2, 3)
                and L.lib.bsy_bottleneck_fused_supported(cv1.conv.in_channels, cv1.conv.out_channels) == 1):
            return "bottleneck"
    if name in ("C3k2", "C2f"):
        mm = getattr(m, "m", None)
        if (isinstance(mm, torch.nn.ModuleList) and len(mm) == 1 and type(mm[0]).__name__ == "Bottleneck" and getattr(mm[0], "add", False)
                and _silu_conv(getattr(m, "cv1", None), 1) and _silu_conv(getattr(m, "cv2", None), 1)
                and _silu_conv(mm[0].cv1, 3) and _silu_conv(mm[0].cv2, 3) and mm[0].cv1.conv.out_channels * 2 == m.c
                and L.lib.bsy_c3k2_fused_supported(m.cv1.conv.in_channels, m.c, m.cv2.conv.out_channels) == 1):
            return "c3k2"
    if name in ("Detect", "Segment"):
        if (getattr(m, "reg_max", 0) == 16 and 1 <= getattr(m, "nl", 0) <= 3 and not getattr(m, "export", False)
                and getattr(m, "format", None) is None and hasattr(m, "_inference")):
            return "detect"
    return None


def _nhwc(x):
    """NCHW fp16 tensor -> its NHWC view (no copy when the tensor is already in channels_last memory format)."""
    return x.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)


def _block_forward(m, kind):
    """The replacement `forward` (`_inference` for Detect) of one block instance.  fp16 CUDA tensors in eval mode; anything
    else reaches the module's own method.  Weights are re-packed when they change (`_weights_version`)."""
    from .ops import _p, _stream, pack_conv_weight
    attr = "_inference" if kind == "detect" else "forward"
    orig = getattr(m, attr)
    st = {"ver": None, "dev": None, "w": None, "calls": 0, "fallbacks": 0, "kind": kind, "attr": attr}

    def usable(self, x):
        return torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float16 and x.dim() == 4 and not self.training

    def ready(self, dev):
        ver = _weights_version(self)
        if st["ver"] != ver or st["dev"] != dev:
            st["w"] = pack(self, dev)
            st["ver"], st["dev"] = ver, dev
        return st["w"]

    if kind == "sppf":
        def pack(self, dev):
            return [pack_conv_weight(*_folded(c), dev) for c in (self.cv1, self.cv2)]

        def fwd(self, x):
            if not usable(self, x) or x.shape[1] != self.cv1.conv.in_channels:
                st["fallbacks"] += 1
                return orig(x)
            (w1, b1), (w2, b2) = ready(self, x.device)
            B, c1, H, W = x.shape
            c_, c2 = self.cv1.conv.out_channels, self.cv2.conv.out_channels
            xn = _nhwc(x)
            cat = torch.empty((B, H, W, 4 * c_), dtype=torch.float16, device=x.device)
            out = torch.empty((B, H, W, c2), dtype=torch.float16, device=x.device)
            with torch.cuda.device(x.device):
                s = _stream(x)  # cv1 writes channels [0, c_) of the concat buffer, the pools fill the rest in place, cv2 reads all of it
                _L.check(_L.lib.bsy_conv2d(_p(xn), c1, B, H, W, c1, _p(w1), _p(b1), _p(cat), 4 * c_, c_, 1, 1, 1, None, 0, 0, s))
                _L.check(_L.lib.bsy_sppf_pool(_p(cat), 4 * c_, B, H, W, c_, s))
                _L.check(_L.lib.bsy_conv2d(_p(cat), 4 * c_, B, H, W, 4 * c_, _p(w2), _p(b2), _p(out), c2, c2, 1, 1, 1, None, 0, 0, s))
            st["calls"] += 1
            return out.permute(0, 3, 1, 2)
    elif kind == "attention":
        def pack(self, dev):
            nh, kd, hd = self.num_heads, self.key_dim, self.head_dim
            per = 2 * kd + hd  # the module emits [q k v] per head (block.py:4274-4276); the kernels want [q | k | v] by heads
            perm = torch.tensor([h * per + i for h in range(nh) for i in range(kd)] + [h * per + kd + i for h in range(nh) for i in range(kd)]
                                + [h * per + 2 * kd + i for h in range(nh) for i in range(hd)])
            wq, bq = _folded(self.qkv)
            wpe, bpe = _folded(self.pe)
            c = nh * hd
            return [pack_conv_weight(wq[perm], bq[perm], dev), pack_conv_weight(*_folded(self.proj), dev),
                    (wpe.view(c, 9).t().contiguous().to(dev), bpe.contiguous().to(dev))]

        def fwd(self, x):
            if not usable(self, x) or x.shape[1] != self.num_heads * self.head_dim:
                st["fallbacks"] += 1
                return orig(x)
            (wq, bq), (wp, bp), (wd, bd) = ready(self, x.device)
            B, c, H, W = x.shape
            nh, kd, hd = self.num_heads, self.key_dim, self.head_dim
            ld = nh * (2 * kd + hd)
            xn = _nhwc(x)
            qkv = torch.empty((B, H, W, ld), dtype=torch.float16, device=x.device)
            att = torch.empty((B, H, W, c), dtype=torch.float16, device=x.device)
            xo = torch.empty_like(att)
            out = torch.empty_like(att)
            with torch.cuda.device(x.device):
                s = _stream(x)
                _L.check(_L.lib.bsy_conv2d(_p(xn), c, B, H, W, c, _p(wq), _p(bq), _p(qkv), ld, ld, 1, 1, 0, None, 0, 0, s))
                _L.check(_L.lib.bsy_attention(_p(qkv), ld, B, H * W, nh, kd, hd, float(self.scale), _p(att), c, s))
                v = qkv[..., 2 * nh * kd:]  # v @ attn^T + pe(v): the depthwise conv reads the v slice and adds the attention output
                _L.check(_L.lib.bsy_dwconv3x3(_p(v), ld, B, H, W, c, _p(wd), _p(bd), _p(xo), c, 0, _p(att), c, s))
                _L.check(_L.lib.bsy_conv2d(_p(xo), c, B, H, W, c, _p(wp), _p(bp), _p(out), c, c, 1, 1, 0, None, 0, 0, s))
            st["calls"] += 1
            return out.permute(0, 3, 1, 2)
    elif kind == "bottleneck":
        def pack(self, dev):
            return [pack_conv_weight(*_folded(c), dev) for c in (self.cv1, self.cv2)]

        def fwd(self, x):
            if not usable(self, x) or x.shape[1] != self.cv1.conv.in_channels:
                st["fallbacks"] += 1
                return orig(x)
            (w1, b1), (w2, b2) = ready(self, x.device)
            B, c, H, W = x.shape
            xn = _nhwc(x)
            out = torch.empty((B, H, W, c), dtype=torch.float16, device=x.device)
            with torch.cuda.device(x.device):
                _L.check(_L.lib.bsy_bottleneck_fused(_p(xn), xn.stride(2), B, H, W, c, self.cv1.conv.out_channels, _p(w1), _p(b1), _p(w2), _p(b2),
                                                     _p(out), c, 1, _stream(x)))
            st["calls"] += 1
            return out.permute(0, 3, 1, 2)
    elif kind == "c3k2":
        def pack(self, dev):
            return [pack_conv_weight(*_folded(c), dev) for c in (self.cv1, self.m[0].cv1, self.m[0].cv2, self.cv2)]

        def fwd(self, x):
            if not usable(self, x) or x.shape[1] != self.cv1.conv.in_channels:
                st["fallbacks"] += 1
                return orig(x)
            pk = ready(self, x.device)
            B, cin, H, W = x.shape
            c2 = self.cv2.conv.out_channels
            xn = _nhwc(x)
            out = torch.empty((B, H, W, c2), dtype=torch.float16, device=x.device)
            with torch.cuda.device(x.device):
                _L.check(_L.lib.bsy_c3k2_fused(_p(xn), xn.stride(2), B, H, W, cin, self.c, c2, _p(pk[0][0]), _p(pk[0][1]), _p(pk[1][0]), _p(pk[1][1]),
                                               _p(pk[2][0]), _p(pk[2][1]), _p(pk[3][0]), _p(pk[3][1]), _p(out), c2, _stream(x)))
            st["calls"] += 1
            return out.permute(0, 3, 1, 2)
    else:  # detect: Detect._inference(x) with x = the per-level (B, 64 + nc, h, w) logit maps (head.py:100-131)
        import ctypes as C

        def pack(self, dev):
            return None

        def fwd(self, x):
            xs = list(x) if isinstance(x, (list, tuple)) else None
            nm = 0
            # export / format / end2end variants compute something else (head.py:66-67, 109-127: dist2bbox(xywh=not end2end), format-
            # specific rescaling), checked at CALL time: exporters flip these attributes on live modules
            if (xs is None or len(xs) != self.nl or self.training or getattr(self, "export", False) or getattr(self, "end2end", False)
                    or getattr(self, "format", None) is not None
                    or not all(torch.is_tensor(t) and t.is_cuda and t.dtype in (torch.float16, torch.float32) and t.dim() == 4
                               and t.shape[1] == self.no and t.shape[0] == xs[0].shape[0] for t in xs)):
                st["fallbacks"] += 1
                return orig(x)
            B, nl, nc = xs[0].shape[0], self.nl, self.nc
            dev, dt = xs[0].device, xs[0].dtype
            maps = [t.permute(0, 2, 3, 1).float().contiguous() for t in xs]  # (B, h, w, no) f32: what bsy_detect_decode reads
            A = sum(t.shape[1] * t.shape[2] for t in maps)
            y = torch.empty((B, 4 + nc, A), dtype=dt, device=dev)
            vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
            box = (vp * nl)(*[t.data_ptr() for t in maps])
            cls = (vp * nl)(*[t.data_ptr() + 64 * 4 for t in maps])
            ldv = (i32 * nl)(*[self.no] * nl)
            hh = (i32 * nl)(*[t.shape[1] for t in maps])
            ww = (i32 * nl)(*[t.shape[2] for t in maps])
            sv = (f32 * nl)(*[float(v) for v in self.stride])
            with torch.cuda.device(dev):
                _L.check(_L.lib.bsy_detect_decode(box, ldv, cls, ldv, None, None, hh, ww, sv, nl, B, nc, nm, _p(y), _L.dtype_code(dt), _stream(y)))
            st["calls"] += 1
            # the reference's _inference caches anchors / strides per input shape as a side effect (head.py:105-107) and code outside
            # the path may read them (exporters, Detect.decode_bboxes callers): keep `shape` as the reference does and drop the cached
            # grids, so that the reference's own _inference -- on a fallback call -- rebuilds them for ITS shape instead of reusing ours
            self.shape = None
            return y
    return fwd, st


def install(model, verbose: bool = False, blocks: bool = True) -> int:
    """Per-module hook: every covered Conv / DWConv instance of `model` runs its convolution in the HIP library, and (blocks=True)
    every SPPF / Attention / Detect instance -- and every Bottleneck / C3k2 whose widths the fused kernels are built for -- runs on
    the library's block operator (fp16 CUDA inputs in eval mode; anything else reaches the module's own method).  Returns the
    number of instances rebound; `uninstall(model)` undoes it.  Call it AFTER `model.fuse()` if the model is to be fused (fuse()
    rebinds `forward` itself).  Use `accelerate(model)` where the whole graph is covered -- it is 5-10x faster (all blocks fused,
    no per-layer Python); this is the level for graphs with modules the engine does not know."""
    n = 0
    for name, m in model.named_modules():
        if hasattr(m, "_bsy_conv") or hasattr(m, "_bsy_block"):
            continue
        spec = _conv_spec(m)
        if spec is not None:
            fwd, st = _conv_forward(m, spec)
            st["orig"] = m.__dict__.get("forward")  # an instance attribute (set by fuse()) or None (the class method)
            m._bsy_conv = st
            m.forward = types.MethodType(fwd, m)
            n += 1
            if verbose:
                print(f"bs_yolo_amd.install: {name} -> {spec}")
            continue
        kind = _block_spec(m) if blocks else None
        if kind is not None:
            fwd, st = _block_forward(m, kind)
            st["orig"] = m.__dict__.get(st["attr"])
            m._bsy_block = st
            setattr(m, st["attr"], types.MethodType(fwd, m))
            n += 1
            if verbose:
                print(f"bs_yolo_amd.install: {name} -> {kind}")
    return n


def uninstall(model) -> int:
    n = 0
    for m in model.modules():
        for key, attr in (("_bsy_conv", None), ("_bsy_block", None)):
            st = m.__dict__.pop(key, None)
            if st is None:
                continue
            a = st.get("attr", "forward")
            if st["orig"] is None:
                delattr(m, a)
            else:
                setattr(m, a, st["orig"])
            n += 1
    return n


def install_nms(ops_module):
    """ops_module = ultralytics.utils.ops.  GPU tensors go to the HIP NMS; everything else to the original."""
    orig = ops_module.non_max_suppression
    if getattr(orig, "_bsy", False):
        return orig

    def non_max_suppression(prediction, *args, **kwargs):
        p = prediction[0] if isinstance(prediction, (list, tuple)) else prediction
        # positional order after `prediction` (utils/ops.py:167-182): conf_thres, iou_thres, classes, agnostic, multi_label,
        # labels (5), max_det, nc, max_time_img, max_nms, max_wh, in_place, rotated (12)
        rotated = kwargs.get("rotated", False) or (len(args) > 12 and args[12])
        labels = kwargs.get("labels", ()) or (args[5] if len(args) > 5 else ())
        if (not isinstance(p, torch.Tensor)) or (not _on_device(p)) or rotated or len(labels) or p.shape[-1] == 6 \
                or p.dtype not in (torch.float16, torch.float32):
            return orig(prediction, *args, **kwargs)
        return _nms.non_max_suppression(prediction, *args, **kwargs)

    non_max_suppression._bsy = True
    non_max_suppression._bsy_orig = orig
    ops_module.non_max_suppression = non_max_suppression
    return non_max_suppression


def install_masks(ops_module):
    """ops_module = ultralytics.utils.ops: GPU calls of process_mask (segment/predict.py:53) go to the HIP kernels."""
    from . import masks as _masks
    orig = ops_module.process_mask
    if getattr(orig, "_bsy", False):
        return orig

    def process_mask(protos, masks_in, bboxes, shape, upsample=False):
        if not (isinstance(protos, torch.Tensor) and protos.is_cuda):
            return orig(protos, masks_in, bboxes, shape, upsample)
        return _masks.process_mask(protos, masks_in, bboxes, shape, upsample)

    process_mask._bsy = True
    process_mask._bsy_orig = orig
    ops_module.process_mask = process_mask
    # the retina_masks path (segment/predict.py:48-50): process_mask_native and scale_masks (utils/ops.py:696-737)
    if hasattr(ops_module, "process_mask_native") and not getattr(ops_module.process_mask_native, "_bsy", False):
        orig_n = ops_module.process_mask_native

        def process_mask_native(protos, masks_in, bboxes, shape):
            if not (isinstance(protos, torch.Tensor) and protos.is_cuda):
                return orig_n(protos, masks_in, bboxes, shape)
            return _masks.process_mask_native(protos, masks_in, bboxes, shape)

        process_mask_native._bsy = True
        process_mask_native._bsy_orig = orig_n
        ops_module.process_mask_native = process_mask_native
    if hasattr(ops_module, "scale_masks") and not getattr(ops_module.scale_masks, "_bsy", False):
        orig_s = ops_module.scale_masks

        def scale_masks(masks, shape, padding=True):
            if not (isinstance(masks, torch.Tensor) and masks.is_cuda and masks.dim() == 4 and masks.dtype in (torch.float16, torch.float32)):
                return orig_s(masks, shape, padding)
            return _masks.scale_masks(masks, shape, padding)

        scale_masks._bsy = True
        scale_masks._bsy_orig = orig_s
        ops_module.scale_masks = scale_masks
    return process_mask


def install_val_metrics(validator):
    """validator = a DetectionValidator instance: its per-image `_process_batch` (models/yolo/detect/val.py:209-228:
    box_iou + match_predictions, numpy on the host in the reference) goes to the device kernel for CUDA detections."""
    from . import val as _val
    orig = validator._process_batch

    def _process_batch(self, detections, gt_bboxes, gt_cls):
        if not (isinstance(detections, torch.Tensor) and _on_device(detections)) or detections.shape[0] > 1024:
            return orig(detections, gt_bboxes, gt_cls)
        return _val.process_batch(detections, gt_bboxes, gt_cls, self.iouv)

    validator._process_batch = types.MethodType(_process_batch, validator)
    return validator


def install_ap_per_class(metrics_module, device="cuda:0"):
    """metrics_module = ultralytics.utils.metrics: `ap_per_class` (:620-706, called by Metric / DetMetrics.process :940) goes
    to the device implementation when no plots are asked for (the plotting path stays the reference's own)."""
    from . import val as _val
    orig = metrics_module.ap_per_class
    if getattr(orig, "_bsy", False):
        return orig

    def ap_per_class(tp, conf, pred_cls, target_cls, plot=False, on_plot=None, save_dir=None, names={}, eps=1e-16, prefix=""):
        n = len(conf)
        if plot or not torch.cuda.is_available() or n == 0 or n > (1 << 20) or getattr(tp, "ndim", 2) != 2 or tp.shape[1] > 16:
            kw = dict(plot=plot, on_plot=on_plot, names=names, eps=eps, prefix=prefix)
            if save_dir is not None:
                kw["save_dir"] = save_dir
            return orig(tp, conf, pred_cls, target_cls, **kw)
        return _val.ap_per_class(tp, conf, pred_cls, target_cls, eps=eps, device=device)

    ap_per_class._bsy = True
    ap_per_class._bsy_orig = orig
    metrics_module.ap_per_class = ap_per_class
    return ap_per_class


def install_preprocess(predictor):
    """predictor = a BasePredictor instance whose model is already set up."""
    from . import letterbox as _lb
    orig = predictor.preprocess

    def preprocess(self, im):
        if isinstance(im, torch.Tensor) or not _device_is_gpu(self.device):
            return orig(im)
        return _lb.preprocess(list(im), tuple(self.imgsz), half=bool(self.model.fp16), pt=bool(self.model.pt),
                              stride=int(self.model.stride), device=str(self.device))

    predictor.preprocess = types.MethodType(preprocess, predictor)
    return predictor
