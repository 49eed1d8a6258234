"""Engine wrapper: owns a bsy_engine (weights in HBM, ONE activation arena) and an LRU set of bsy_plans, one per input shape.

`YoloEngine.__call__(im)` has the calling convention AutoBackend expects from an in-memory model
(nn/autobackend.py:524: ``self.model(im, augment=, visualize=, embed=)`` -> ``(y, x_list)``, head.py:74), with
``im`` a contiguous BCHW fp16/fp32 tensor on the engine's device.  PyTorch is only used for device memory and
the current stream; all compute goes through libbsyolo_hip.so.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
from collections import OrderedDict
from typing import Dict, List, Mapping, Optional, Tuple

import torch

from . import lib as L
from .plan import Plan
from .weights import BN_EPS, adopt_offsets, pack_plan_weights


class YoloEngine:
    def __init__(self, cfg: dict, state_dict: Mapping[str, torch.Tensor], device: int = 0, bn_eps: float = BN_EPS,
                 autotune: Optional[bool] = None, fuse_stem: Optional[bool] = None,
                 fuse_bneck: Optional[bool] = None, fuse_head: Optional[bool] = None, fuse_dwpw: Optional[bool] = None,
                 merge_c3k: Optional[bool] = None, fuse_msca: Optional[bool] = None, fuse_tail: Optional[bool] = None,
                 max_plans: Optional[int] = None, precision: str = "fp16", graph: Optional[bool] = None, graph_ring: int = 3,
                 latency: Optional[bool] = None, fuse_pmsfa: Optional[bool] = None, fuse_chain: Optional[bool] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("bs_yolo_amd needs a ROCm GPU (there is no CPU fallback)")
        self.cfg = cfg
        # "fp32": the correctness mode -- fp32 activation storage and arithmetic (csrc/ref32.hip, conv32_mfma.hip), for callers that
        # expect the fp32 model's numbers bit-reproducibly (|dscore| <= 1e-5 against the fp32 reference).  "fp32x": the same storage
        # and non-conv kernels, dense convs on the fp16 matrix pipe with split-f16 operands (csrc/conv32x_mfma.hip): the
        # north-star's 1e-3 at several times the fp32 mode's throughput.  "fp16" is the product path.
        self.precision = precision
        # latency mode (BSY_LATENCY=1): split-K on the long-K, few-tile conv layers -- for small per-rank batches (a rank's 8- or
        # 16-image share of a strong-scaled batch).  The split factors depend on the layer shape only, so any batch split returns the
        # bits of the whole batch IN THIS MODE; they differ from the default mode's by f32 summation order (Plan.split_factors).
        self.latency = (os.environ.get("BSY_LATENCY", "0") == "1") if latency is None else bool(latency)
        self.device = torch.device("cuda", device)
        self._h = C.c_void_p()
        L.check(L.lib.bsy_engine_create(device, C.byref(self._h)))
        # plans: least-recently-used first.  val runs with rect=True (engine/model.py:635; per-batch shapes,
        # data/base.py:261-284) and predict's `auto` letterbox also varies (H, W): a plan is cheap to rebuild (host-side
        # flattening + a handful of streams / events), its activations live in the engine's shared arena and its tuning in
        # `_tune_cache`, so old shapes are simply dropped
        self._plans: "OrderedDict[Tuple, Tuple]" = OrderedDict()
        self.max_plans = int(os.environ.get("BSY_MAX_PLANS", "8")) if max_plans is None else int(max_plans)
        self._tuned = set()
        self._elems: Dict[Tuple[int, int], int] = {}
        # autotune results keyed by conv shape (Plan.conv_signature), shared by every plan of this engine and, with
        # BSY_TUNE_CACHE=<file>, across processes.  `tune_stats` counts what bsy_plan_autotune had to time.
        self._tune_cache: Dict[tuple, int] = {}
        # second level: the same layer (channels, kernel, stride, epilogue) at a similar size (pixel count within a factor of
        # two) reuses the winner without timing -- rect batches differ from each other by a few rows or columns
        self._tune_family: Dict[tuple, int] = {}
        self.tune_stats = {"timed_ops": 0, "cached_ops": 0, "family_ops": 0, "autotune_calls": 0, "in_place_flips": 0}
        # how conv configurations are timed (see _autotune): "1" every candidate in place (default), "3" one layer at a time
        # (bsy_plan_autotune) + its top three re-timed in place, "0" one layer at a time only
        self.tune_mode = os.environ.get("BSY_TUNE_IN_PLACE", "1")
        self.tune_in_place = self.tune_mode == "3"
        self._tune_file = os.environ.get("BSY_TUNE_CACHE")
        if self._tune_file and os.path.exists(self._tune_file):
            try:
                self._tune_cache = {tuple(k): int(v) for k, v in json.load(open(self._tune_file))}
                for k, v in self._tune_cache.items():
                    self._tune_family.setdefault(self._family(k), v)
            except (OSError, ValueError):
                self._tune_cache = {}
        # liveness-based buffer reuse (Plan.assign_offsets).  Off under BSY_ARENA_REUSE=0 (tests that read intermediate
        # layers back) and under BSY_PLAN_GUARD (guard bands behind every buffer: the plan then owns a private workspace)
        self.reuse = os.environ.get("BSY_ARENA_REUSE", "1") != "0" and not os.environ.get("BSY_PLAN_GUARD")
        # Graph mode (opt-in: graph=True / BSY_GRAPH=1): the forward of a shape is captured once per set of buffer addresses and
        # replayed as ONE hipGraph launch (bsy_plan_graph_launch) -- what a rank needs at its 8- or 16-image share of a strong-
        # scaled batch, where 70-odd launches of 4-15 us are bound by the host's launch rate.  The graph bakes addresses in, so the
        # outputs come from a ring of `graph_ring` preallocated sets per shape: a returned `y` is overwritten by the graph_ring-th
        # forward after it (the reference's predict / val loops consume `preds` within the iteration, engine/predictor.py:254-262).
        # Head lanes are on at every size in this mode (fork / join are graph edges).
        self.graph = (os.environ.get("BSY_GRAPH", "0") == "1") if graph is None else bool(graph)
        self.graph_ring = max(1, int(graph_ring))
        self._rings: Dict[Tuple, dict] = {}
        self._gstream = None
        # All plans of an engine share ONE activation arena (liveness-packed): two forwards must never be in flight at once.  The
        # reference serialises predict() with a lock (engine/predictor.py:113,229); here a host lock covers the enqueue and an event
        # recorded behind every forward makes a forward enqueued on ANOTHER stream wait for the previous one (same stream: ordered
        # anyway, no event wait is issued).
        import threading
        self._lock = threading.RLock()
        self._last_done = None
        self._last_stream = None
        self.graph_stats = {"captures": 0, "replays": 0, "eager": 0}
        self.autotune = ((os.environ.get("BSY_AUTOTUNE", "1") != "0") if autotune is None else bool(autotune)) and precision == "fp16"
        # pack once with a throw-away plan (op list structure does not depend on the input size)
        self.fuse_stem = fuse_stem  # None: BSY_FUSE_STEM env (default on); False keeps layers 0 and 1 as two launches
        self.fuse_bneck = fuse_bneck
        self.fuse_head = fuse_head
        self.fuse_dwpw = fuse_dwpw
        self.merge_c3k = merge_c3k
        self.fuse_msca = fuse_msca
        self.fuse_tail = fuse_tail
        self.fuse_pmsfa = fuse_pmsfa
        self.fuse_chain = fuse_chain
        self._packed = Plan(cfg, 1, 64, 64, **self._fuse_kw())
        blob = pack_plan_weights(self._packed, state_dict, bn_eps)
        self.weight_bytes = len(blob)
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        L.check(L.lib.bsy_engine_load_weights(self._h, buf, len(blob)))
        self.meta = dict(self._packed.meta)
        self.nc = self.meta["nc"]
        self.stride = torch.tensor(self.meta["strides"])
        self.names = {i: f"{i}" for i in range(self.nc)}

    def load_weights(self, state_dict: Mapping[str, torch.Tensor], bn_eps: float = BN_EPS) -> None:
        """Replace the engine's weights in place (same graph, new values: an EMA update, a fine-tuned checkpoint).  Plans, tuning and
        the arena are kept; the library synchronises the device before it frees the old blob and drops every captured graph (they
        hold addresses inside it: include/bsyolo.h, bsy_engine_load_weights)."""
        with self._lock:
            blob = pack_plan_weights(self._packed, state_dict, bn_eps)
            if len(blob) != self.weight_bytes:
                raise ValueError(f"state_dict packs to {len(blob)} bytes, the engine was built for {self.weight_bytes}")
            for plan, _ in self._plans.values():
                adopt_offsets(plan, self._packed)
            buf = (C.c_char * len(blob)).from_buffer_copy(blob)
            L.check(L.lib.bsy_engine_load_weights(self._h, buf, len(blob)))

    def _fuse_kw(self):
        return dict(fuse_stem=self.fuse_stem, fuse_bneck=self.fuse_bneck, fuse_head=self.fuse_head, fuse_dwpw=self.fuse_dwpw,
                    merge_c3k=self.merge_c3k, fuse_msca=self.fuse_msca, fuse_tail=self.fuse_tail, precision=self.precision,
                    lanes=True if getattr(self, "graph", False) else None, latency=getattr(self, "latency", False),
                    fuse_pmsfa=self.fuse_pmsfa, fuse_chain=self.fuse_chain)

    # -- plans --------------------------------------------------------------------------------------------------
    def plan_for(self, B: int, H: int, W: int, in_dtype: torch.dtype, out_dtype: torch.dtype):
        key = (B, H, W, in_dtype, out_dtype)
        hit = self._plans.get(key)
        if hit is not None:
            self._plans.move_to_end(key)
            return hit
        if H % 32 or W % 32:
            raise ValueError(f"input {H}x{W} must be a multiple of the max stride 32 (utils/checks.py:120-172)")
        plan = Plan(self.cfg, B, H, W, L.dtype_code(in_dtype), L.dtype_code(out_dtype), **self._fuse_kw())
        adopt_offsets(plan, self._packed)
        ops = plan.c_ops()
        sizes = (C.c_int64 * len(plan.buf_bytes))(*plan.buf_bytes)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            if os.environ.get("BSY_PLAN_GUARD"):  # test aid: private workspace with guard bands, no aliasing
                L.check(L.lib.bsy_plan_create(self._h, ops, len(plan.ops), sizes, len(plan.buf_bytes), C.byref(h)))
                plan.arena_bytes = sum((b + 255) & ~255 for b in plan.buf_bytes)
            else:
                offs, arena = plan.assign_offsets(self.reuse)
                plan.arena_bytes = arena
                L.check(L.lib.bsy_plan_create_arena(self._h, ops, len(plan.ops), sizes, (C.c_int64 * len(offs))(*offs),
                                                    len(offs), arena, C.byref(h)))
        # conv shapes an earlier plan (or an earlier process, BSY_TUNE_CACHE) already tuned: preset, not re-timed
        plan.conv_sigs = [plan.conv_signature(o) for o in plan.ops]
        for sg in plan.conv_sigs:
            if sg is not None and sg not in self._tune_cache and self._family(sg) in self._tune_family:
                self._tune_cache[sg] = self._tune_family[self._family(sg)]
                self.tune_stats["family_ops"] += 1
        preset = (C.c_int32 * len(plan.ops))(*[self._tune_cache.get(sg, -1) if sg is not None else -1 for sg in plan.conv_sigs])
        L.check(L.lib.bsy_plan_set_tuning(h, preset, len(plan.ops)))
        while len(self._plans) >= max(self.max_plans, 1):  # evict the least recently used shape
            old_key, (_, old) = self._plans.popitem(last=False)
            L.lib.bsy_plan_destroy(old)
            for rk in [k for k in self._rings if k[0] == old_key]:
                del self._rings[rk]
        self._plans[key] = (plan, h)
        return plan, h

    @property
    def arena_bytes(self) -> int:
        """Bytes of the activation arena all plans of this engine share (the largest plan so far)."""
        return int(L.lib.bsy_engine_arena_bytes(self._h))

    @staticmethod
    def _family(sg: tuple) -> tuple:
        B, H, W = sg[:3]
        return sg[3:] + (int(round(math.log2(max(B * H * W, 1)))),)

    def _autotune(self, plan, h, ext, n, stream):
        """First forward of a shape: time the candidate configurations of the conv ops whose shape has not been tuned yet
        (bsy_plan_autotune skips preset ops) and remember the winners by conv shape."""
        # presets that came from the tune file or from a similar-sized shape's entry may not fit this shape (another library version,
        # another alignment): drop them from the caches and from the plan, so that they are timed below instead of being persisted
        nops = len(plan.ops)
        valid = (C.c_int32 * nops)()
        L.check(L.lib.bsy_plan_check_tuning(h, ext, n, valid, nops))
        bad = [i for i in range(nops) if valid[i] == 0]
        if bad:
            clear = (C.c_int32 * nops)(*([-1] * nops))
            for i in bad:
                sg = plan.conv_sigs[i]
                cfg = self._tune_cache.pop(sg, None)
                if cfg is not None and self._tune_family.get(self._family(sg)) == cfg:
                    del self._tune_family[self._family(sg)]
                clear[i] = -2
            L.check(L.lib.bsy_plan_set_tuning(h, clear, nops))
            self.tune_stats["invalid_presets"] = self.tune_stats.get("invalid_presets", 0) + len(bad)
        todo = sum(1 for sg in plan.conv_sigs if sg is not None and sg not in self._tune_cache)
        self.tune_stats["autotune_calls"] += 1
        self.tune_stats["timed_ops"] += todo
        self.tune_stats["cached_ops"] += sum(1 for sg in plan.conv_sigs if sg is not None) - todo
        if self.tune_mode == "1":
            # every candidate of every new conv shape timed where it runs: pass k of the forward runs candidate k of all of them
            L.check(L.lib.bsy_plan_autotune_in_place(h, ext, n, C.c_void_p(stream), 3))
        else:
            L.check(L.lib.bsy_plan_autotune(h, ext, n, C.c_void_p(stream)))
        if todo:
            nops = len(plan.ops)
            out = (C.c_int32 * nops)()
            L.check(L.lib.bsy_plan_get_tuning(h, out, nops))
            if self.tune_in_place:
                # Second opinion for near-ties: bsy_plan_autotune launches one layer back to back (operands in L2 / Infinity Cache);
                # here winner and runner-up of every newly tuned op are timed where they run -- serial profile passes of the whole
                # forward, all winners then all runner-ups, best of three each -- and the faster one is kept
                def passes(k=3):
                    best = [1e30] * nops
                    ms = (C.c_float * nops)()
                    for _ in range(k):
                        L.check(L.lib.bsy_plan_profile(h, ext, n, C.c_void_p(stream), ms))
                        best = [min(b_, float(t)) for b_, t in zip(best, ms)]
                    return best
                t_cur, flips = None, 0
                for rank in (1, 2):
                    alt = (C.c_int32 * nops)()
                    L.check(L.lib.bsy_plan_get_tuning_alt(h, rank, alt, nops))
                    idx = [i for i, sg in enumerate(plan.conv_sigs) if sg is not None and sg not in self._tune_cache and alt[i] >= 0 and out[i] >= 0]
                    if not idx:
                        break
                    if t_cur is None:
                        t_cur = passes()
                    sel = (C.c_int32 * nops)(*([-1] * nops))
                    for i in idx:
                        sel[i] = alt[i]
                    L.check(L.lib.bsy_plan_set_tuning(h, sel, nops))
                    t_alt = passes()
                    for i in idx:
                        if t_alt[i] < t_cur[i]:
                            out[i], t_cur[i] = alt[i], t_alt[i]
                            flips += 1
                        sel[i] = out[i]
                    L.check(L.lib.bsy_plan_set_tuning(h, sel, nops))
                if flips:
                    self.tune_stats["in_place_flips"] = self.tune_stats.get("in_place_flips", 0) + flips
            for sg, c in zip(plan.conv_sigs, out):
                if sg is not None and c >= 0:
                    self._tune_cache.setdefault(sg, int(c))
                    self._tune_family.setdefault(self._family(sg), int(c))
            if self._tune_file:
                try:
                    tmp = self._tune_file + f".{os.getpid()}.tmp"
                    json.dump([[list(k), v] for k, v in self._tune_cache.items()], open(tmp, "w"))
                    os.replace(tmp, self._tune_file)
                except OSError:
                    pass

    def _max_view_elems(self, H: int, W: int) -> int:
        """Largest (pixels x row stride) of any activation view of ONE image at this size: the kernels keep element offsets in
        32 bits, so a batch is split when B times this reaches 2^31 (C2f concat buffers are larger than the first
        conv's output, which the round-1 estimate used)."""
        key = (H, W)
        hit = self._elems.get(key)
        if hit is None:
            p1 = Plan(self.cfg, 1, H, W, **self._fuse_kw())
            hit = 1
            for o in p1.ops:
                for k in ("src0", "src1", "dst", "res"):
                    t = o.get(k)
                    if t is not None and t.buf < L.BSY_EXT_BASE:
                        hs, ws = (t.H // 2, t.W // 2) if t.up else (t.H, t.W)
                        hit = max(hit, hs * ws * t.ld)
            self._elems[key] = hit
        return hit

    def _ext(self, im, y, raws, proto=None):
        ptrs = [im.data_ptr(), y.data_ptr()] + [r.data_ptr() if r is not None else None for r in raws]
        ptrs.append(proto.data_ptr() if proto is not None else None)
        return (C.c_void_p * len(ptrs))(*ptrs), len(ptrs)

    def forward(self, im: torch.Tensor, want_raw: bool = True):
        if im.device != self.device:
            raise ValueError(f"input on {im.device}, engine on {self.device}")
        if im.dtype not in (torch.float16, torch.float32):
            raise TypeError("input must be fp16 or fp32 (BCHW, 0..1)")
        im = im.contiguous()
        B, Cc, H, W = im.shape
        if Cc != 3:
            raise ValueError("expected 3 input channels")
        # the kernels keep element offsets in 32 bits: split batches whose largest activation would exceed 2^31
        # elements (e.g. 256 x 1280x1280) into equal chunks -- images are independent
        big = self._max_view_elems(H, W)
        if B > 1 and B * big >= (1 << 31) - (1 << 24):
            n_chunks = -(-B * big // ((1 << 31) - (1 << 24)))
            step = -(-B // n_chunks)
            outs = [self.forward(im[i:i + step], want_raw) for i in range(0, B, step)]
            y = torch.cat([o[0] for o in outs])
            if self.meta["nm"]:
                raws = [torch.cat([o[1][0][l] for o in outs]) for l in range(3)] if want_raw else [None] * 3
                return y, (raws, y[:, 4 + self.meta["nc"]:], torch.cat([o[1][2] for o in outs]))
            return y, ([torch.cat([o[1][l] for o in outs]) for l in range(3)] if want_raw else [None] * 3)
        with self._lock:  # plan table, arena and launch order are shared state
            return self._forward_one(im, want_raw, B, H, W)

    def _forward_one(self, im, want_raw, B, H, W):
        plan, h = self.plan_for(B, H, W, im.dtype, im.dtype)
        m = plan.meta

        def outputs():
            y = torch.empty((B, 4 + m["nc"] + m["nm"], m["A"]), dtype=im.dtype, device=self.device)
            raws: List[Optional[torch.Tensor]] = [None, None, None]
            if want_raw:
                raws = [torch.empty((B, m["no"], lh, lw), dtype=im.dtype, device=self.device) for lh, lw in m["levels"]]
            proto = None
            if m["nm"]:
                ph, pw = m["proto_hw"]
                proto = torch.empty((B, m["nm"], ph, pw), dtype=im.dtype, device=self.device)
            return y, raws, proto

        if self.graph:  # outputs from this shape's ring: the captured graphs hold their addresses
            ring = self._rings.setdefault(((B, H, W, im.dtype, im.dtype), bool(want_raw)), {"i": 0, "slots": []})
            if len(ring["slots"]) < self.graph_ring:
                ring["slots"].append(outputs())
            y, raws, proto = ring["slots"][ring["i"] % len(ring["slots"])]
            ring["i"] += 1
        else:
            y, raws, proto = outputs()
        ext, n = self._ext(im, y, raws, proto)
        cur_stream = torch.cuda.current_stream(self.device)
        stream = cur_stream.cuda_stream
        if self._last_done is not None and self._last_stream != stream:
            cur_stream.wait_event(self._last_done)  # the arena is still owned by a forward on another stream
        if self.autotune and (B, H, W, im.dtype) not in self._tuned:
            # first call for this shape: pick the fastest kernel configuration per conv op (runs the plan once)
            self._tuned.add((B, H, W, im.dtype))
            self._autotune(plan, h, ext, n, stream)
        if self.graph:
            how = C.c_int(-1)
            if stream == 0:
                # the legacy default stream cannot be captured: capture / replay on a stream of the engine's, ordered after the
                # caller's work and joined back into it (two event pairs per forward)
                cur = torch.cuda.current_stream(self.device)
                if self._gstream is None:
                    self._gstream = torch.cuda.Stream(device=self.device)
                self._gstream.wait_stream(cur)
                im.record_stream(self._gstream)
                L.check(L.lib.bsy_plan_graph_launch(h, ext, n, C.c_void_p(self._gstream.cuda_stream), C.byref(how)))
                cur.wait_stream(self._gstream)
            else:
                L.check(L.lib.bsy_plan_graph_launch(h, ext, n, C.c_void_p(stream), C.byref(how)))
            self.graph_stats[{1: "captures", 0: "replays"}.get(how.value, "eager")] += 1
            if how.value == 1 and self.graph_stats["captures"] > 4 * self.graph_ring + 4 and not getattr(self, "_graph_warned", False):
                # the graph key holds the INPUT pointer: callers that hand over a freshly allocated image tensor per call re-capture
                # (and re-instantiate) on almost every forward -- slower than the eager replay; say so once
                self._graph_warned = True
                import logging
                logging.getLogger("bs_yolo_amd").warning(
                    "graph mode has captured %d graphs (%d replays): inputs arrive at new addresses on every call, so every forward is "
                    "re-captured.  Reuse the input tensor (copy_ into one buffer per shape) or switch graph mode off.",
                    self.graph_stats["captures"], self.graph_stats["replays"])
        else:
            L.check(L.lib.bsy_plan_run(h, ext, n, C.c_void_p(stream)))
        if self._last_done is None:
            self._last_done = torch.cuda.Event()
        self._last_done.record(cur_stream)
        self._last_stream = stream
        if m["nm"]:  # Segment.forward (head.py:197): (cat(y, mc), (raw, mc, proto)); y already carries the mc rows
            return y, (raws, y[:, 4 + m["nc"]:], proto)
        return y, raws

    def tuning(self, B, H, W, dtype=torch.float16):
        plan, h = self.plan_for(B, H, W, dtype, dtype)
        out = (C.c_int32 * len(plan.ops))()
        L.check(L.lib.bsy_plan_get_tuning(h, out, len(plan.ops)))
        return [(o["name"], int(c)) for o, c in zip(plan.ops, out) if c >= 0]

    def __call__(self, im, augment=False, visualize=False, embed=None, want_raw=True):
        if augment or visualize or embed:
            raise NotImplementedError("augment/visualize/embed fall back to the reference graph (tasks.py:134-164)")
        return self.forward(im, want_raw)

    def profile(self, im: torch.Tensor):
        """Per-op device time (ms) of one forward, measured with HIP events on the current stream."""
        im = im.contiguous()
        B, _, H, W = im.shape
        plan, h = self.plan_for(B, H, W, im.dtype, im.dtype)
        m = plan.meta
        y = torch.empty((B, 4 + m["nc"] + m["nm"], m["A"]), dtype=im.dtype, device=self.device)
        ext, n = self._ext(im, y, [None, None, None])
        ms = (C.c_float * len(plan.ops))()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        L.check(L.lib.bsy_plan_profile(h, ext, n, C.c_void_p(stream), ms))
        return [(o["name"], o["kind"], float(t)) for o, t in zip(plan.ops, ms)], plan

    def check_guards(self):
        """Test aid (engines created with BSY_PLAN_GUARD=<bytes> set): [(plan key, buffer index, byte offset, op names that
        write that buffer)] for every plan with a damaged guard band -- empty when no kernel stored out of bounds."""
        bad = []
        for key, (plan, h) in self._plans.items():
            b, off = C.c_int32(-1), C.c_int64(0)
            L.check(L.lib.bsy_plan_check_guards(h, C.byref(b), C.byref(off)))
            if b.value >= 0:
                writers = [o["name"] for o in plan.ops for k in ("dst", "res") if o.get(k) is not None and getattr(o[k], "buf", -1) == b.value and k == "dst"]
                bad.append((key[:3], b.value, off.value, writers))
        return bad

    def read_view(self, plan, h, t) -> torch.Tensor:
        """Debug/test aid: copy an activation view back as a (B, C, H, W) fp32 CPU tensor."""
        hs, ws = (t.H // 2, t.W // 2) if t.up else (t.H, t.W)
        n = plan.B * hs * ws * t.ld
        host = torch.empty(n, dtype=torch.float32 if t.f32 else torch.float16)
        L.check(L.lib.bsy_plan_copy_buffer(h, t.buf, C.c_void_p(host.data_ptr()), n * host.element_size()))
        v = host.view(plan.B, hs, ws, t.ld)[..., t.coff:t.coff + t.C].permute(0, 3, 1, 2).float()
        return torch.nn.functional.interpolate(v, scale_factor=2.0, mode="nearest") if t.up else v.contiguous()

    def close(self):
        for _, h in self._plans.values():
            L.lib.bsy_plan_destroy(h)
        self._plans.clear()
        if self._h:
            L.lib.bsy_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
