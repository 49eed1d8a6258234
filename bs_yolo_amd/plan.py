"""Host-side graph flattening: model description -> flat list of device ops (include/bsyolo.h `bsy_op`).

Mirrors what the reference does at run time in Python:
  * nn/tasks.py:940-1105 parse_model  -- channel scaling ``make_divisible(min(c2, max_ch) * width, 8)`` (:1016),
    repeat scaling ``max(round(n * depth), 1)`` (:972), ``c3k=True`` for m/l/x (:1046-1049), ``legacy`` Detect head
    for graphs without C3k2 (:1048, head.py:44-56);
  * nn/tasks.py:138-165 _predict_once -- the ``f`` / save routing between top-level layers;
  * the forward of every module on the path (conv.py:133-151; block.py:3114-3149, 3295-3334, 3405-3419, 3796-3815,
    4235-4288, 4348-4383, 4429-4468; head.py:64-131).
Instead of executing modules, every forward is expanded into device ops over NHWC buffers:
  chunk/split/cat -> channel-slice views of one buffer; nn.Upsample -> `up` flag on the consumer's source;
  Concat -> two-source conv; shortcut adds -> conv epilogue residual.

The model description is the reference's own yaml dict schema (``model.yaml`` of a live reference model, or the
stock tables in graphs.py), so a reference model drops in without translation.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple, Union

from . import lib as L


def make_divisible(x, d):  # utils/ops.py:130-143
    return math.ceil(x / d) * d


@dataclass
class T:
    """A logical NHWC activation: channel slice [coff, coff+C) of buffer `buf` (row stride `ld`), H x W pixels.
    up=True: stored at (H/2, W/2), read through nearest x2 (virtual nn.Upsample)."""
    buf: int
    ld: int
    coff: int
    C: int
    H: int
    W: int
    up: bool = False
    f32: bool = False
    # Zero-padded pieces (round 4; the fp16 path reads 8-channel pieces, so a C2f / C3k2 whose chunk width is 12 or 20 channels carries every
    # chunk on the next multiple of 8): cmap[j] = which REAL channel of this tensor the stored channel j is, or -1 for a padding channel
    # that is always zero (zero weight rows and bias where it is written, SiLU(0) = 0; zero weight columns where it is read).  None = all real.
    cmap: Optional[Tuple[int, ...]] = None

    def slice(self, c0: int, c: int) -> "T":
        assert 0 <= c0 and c0 + c <= self.C
        cm = None
        if self.cmap is not None:
            before = sum(1 for v in self.cmap[:c0] if v >= 0)
            cm = tuple(v - before if v >= 0 else -1 for v in self.cmap[c0:c0 + c])
            if all(v >= 0 for v in cm):
                cm = None
        return T(self.buf, self.ld, self.coff + c0, c, self.H, self.W, self.up, self.f32, cm)

    @property
    def real(self) -> int:
        """Channels that carry data (== C unless the tensor has padding pieces)."""
        return self.C if self.cmap is None else sum(1 for v in self.cmap if v >= 0)

    def padded(self, pieces: Sequence[Tuple[int, int]]) -> "T":
        """This tensor seen as consecutive (stored width, real width) pieces, each zero-padded at its end."""
        cm, base = [], 0
        for width, real in pieces:
            cm += [base + i if i < real else -1 for i in range(width)]
            base += real
        assert len(cm) == self.C
        return T(self.buf, self.ld, self.coff, self.C, self.H, self.W, self.up, self.f32, tuple(cm))

    def view(self):
        return (self.buf, self.ld, self.coff, self.C)


@dataclass
class WRec:
    """How to pack one op's parameters into the weight blob (weights.py)."""
    name: str          # state_dict prefix, e.g. "model.2.m.0.cv1"  (Conv: .conv.weight + .bn.*; plain: .weight/.bias)
    kind: str          # conv | conv2 (name + post stacked along cout) | plain | dw | first | deconv
    cout: int
    cin: int
    k: int
    perm: Optional[List[int]] = None   # output-channel permutation (qkv)
    tap: Optional[Tuple[int, int]] = None  # deconv: (dy, dx)
    kw: Optional[int] = None           # dwg: kernel width when != k (strip convs)
    post: Optional[str] = None         # dwg: name of a depthwise 1x1 conv folded in after this one (MSCA dilconv)
    coef: Tuple[float, float, float] = (0.0, 0.0, 0.0)  # ela: sigmoid(ch_weight / sp_weight / res_weight), set when packed
    real_cout: int = 0                 # the module's own output / input channels where the op runs on a width padded to a multiple of 8
    real_cin: int = 0                  # (Detect's class branch with nc-dependent widths, head.py:39): zero weights in the padding; 0 = as cout / cin
    rows: Optional[List[int]] = None   # op output channel i takes the module's output channel rows[i] (-1: a zero channel, weights and bias 0);
    cols: Optional[List[int]] = None   # op input channel j is the module's input channel cols[j] (-1: zero weights) -- PMSFA on padded pieces
    w_off: int = -1
    b_off: int = -1


def stem_supported(c0: int, c1: int, H: int, W: int) -> bool:
    """Shapes csrc/stem_fused.hip accepts (mirror of stem_fused_supported / bsy_stem_fused_supported)."""
    return (c0, c1) in ((32, 64), (16, 32)) and W % 4 == 0 and H >= 4 and W >= 4


def msca_spatial_supported(H: int, W: int) -> bool:
    """Maps whose (H x W x 8 channel) slabs fit LDS (mirror of csrc/bsyolo_ops.hip msca_spatial_supported)."""
    return H > 0 and W > 0 and H * W <= 1890


def attention_supported(kd: int, hd: int) -> bool:
    """Head shapes csrc/attention.hip is instantiated for (mirror of attention_supported there)."""
    return 0 < kd <= 64 and kd % 16 == 0 and 0 < hd <= 128 and hd % 32 == 0


def pmsfa_tail_supported(c: int) -> bool:
    """Widths csrc/pmsfa_fused.hip is instantiated for (mirror of pmsfa_tail_supported there)."""
    return c in (32, 64)


def dwpw_supported(c: int, cout: int) -> bool:
    """Widths dwpw_fused_kernel accepts (mirror of bsy_dwpw_fused_supported)."""
    return c > 0 and c % 32 == 0 and c <= 256 and cout % 8 == 0


def c3k2_supported(cin: int, c: int, c2: int) -> bool:
    """Widths csrc/c3k2_fused.hip is instantiated for (mirror of bsy_c3k2_fused_supported)."""
    return (cin, c, c2) == (64, 32, 128)


def chain_supported(ca0: int, ca1: int, n1: int, keep0: int, lc: int, ch2: int, n2: int) -> bool:
    """Shapes csrc/chain1x1.hip takes (mirror of chain_supported there): whole 64-deep K-steps, whole 128-cout passes, a resident
    tile of at most 256 channels inside stage 1's output."""
    return (ca0 > 0 and ca0 % 64 == 0 and ca1 >= 0 and ca1 % 64 == 0 and n1 > 0 and n1 % 128 == 0 and n2 > 0 and n2 % 128 == 0
            and 0 < lc <= 256 and lc % 64 == 0 and keep0 >= 0 and keep0 % 64 == 0 and keep0 + lc <= n1 and ch2 >= 0 and ch2 % 64 == 0)


# pixels from which two chained 1x1 convs run as one launch (one 128-pixel tile per workgroup, every weight streamed once per tile: with
# fewer tiles than CUs the two ordinary launches, which also tile along cout, are faster): 192 tiles
CHAIN_MIN_PIXELS = 192 * 128


def bneck_supported(c: int, ch: int) -> bool:
    """Widths csrc/bneck_fused.hip accepts (mirror of bsy_bottleneck_fused_supported)."""
    return (c, ch) in ((32, 16), (64, 32))


class Plan:
    """Flat op list + workspace buffer table for one (B, H, W)."""

    EXT_IMG, EXT_Y, EXT_RAW0 = 0, 1, 2  # external slots: image, prediction, raw level maps (2,3,4), proto (5)
    EXT_PROTO = 5

    def __init__(self, cfg: dict, B: int, H: int, W: int, in_dtype: int = L.BSY_F16, out_dtype: int = L.BSY_F16,
                 fuse_stem: Optional[bool] = None, fuse_bneck: Optional[bool] = None, fuse_head: Optional[bool] = None,
                 fuse_dwpw: Optional[bool] = None, merge_c3k: Optional[bool] = None, fuse_msca: Optional[bool] = None,
                 fuse_tail: Optional[bool] = None, precision: str = "fp16", lanes: Optional[bool] = None, latency: bool = False,
                 fuse_pmsfa: Optional[bool] = None, fuse_chain: Optional[bool] = None):
        self.cfg, self.B, self.H, self.W = cfg, B, H, W
        # latency mode (round 4; fp16 path): long-K conv layers with few tiles per image run split-K (split_factors below)
        self.latency = bool(latency) and precision == "fp16"
        if precision not in ("fp16", "fp32", "fp32x"):
            raise ValueError(f"precision must be 'fp16', 'fp32' or 'fp32x', not {precision!r}")
        # fp32: the correctness mode (csrc/ref32.hip) -- every buffer f32, no fused kernels, same op list otherwise.
        # fp32x: the same plan; its dense convs multiply on the fp16 matrix pipe with split-f16 operands (csrc/conv32x_mfma.hip)
        self.f32_mode = precision in ("fp32", "fp32x")
        self.split_f16 = precision == "fp32x"
        if self.f32_mode:
            fuse_stem = fuse_bneck = fuse_head = fuse_dwpw = merge_c3k = fuse_msca = fuse_tail = fuse_pmsfa = fuse_chain = False
        self.in_dtype, self.out_dtype = in_dtype, out_dtype
        self.fuse_stem = (os.environ.get("BSY_FUSE_STEM", "1") != "0") if fuse_stem is None else bool(fuse_stem)
        self.fuse_bneck = (os.environ.get("BSY_FUSE_BNECK", "1") != "0") if fuse_bneck is None else bool(fuse_bneck)
        self.fuse_head = (os.environ.get("BSY_FUSE_HEAD", "1") != "0") if fuse_head is None else bool(fuse_head)
        # box-branch tail (cv2.x.1 + cv2.x.2 + DFL in one launch): part of the head fusion, BSY_FUSE_BOXTAIL=0 switches it off alone
        self.fuse_boxtail = self.fuse_head and os.environ.get("BSY_FUSE_BOXTAIL", "1") != "0"
        self.fuse_dwpw = (os.environ.get("BSY_FUSE_DWPW", "1") != "0") if fuse_dwpw is None else bool(fuse_dwpw)
        self.merge_c3k = (os.environ.get("BSY_MERGE_C3K", "1") != "0") if merge_c3k is None else bool(merge_c3k)
        self.fuse_msca = (os.environ.get("BSY_FUSE_MSCA", "1") != "0") if fuse_msca is None else bool(fuse_msca)
        self.fuse_tail = (os.environ.get("BSY_FUSE_TAIL", "1") != "0") if fuse_tail is None else bool(fuse_tail)
        # PMSFA's depthwise 5x5 -> depthwise 7x7 -> 1x1 + shortcut as one launch (round 4, csrc/pmsfa_fused.hip)
        self.fuse_pmsfa = (os.environ.get("BSY_FUSE_PMSFA", "1") != "0") if fuse_pmsfa is None else bool(fuse_pmsfa)
        # per-pixel chains of two 1x1 convs as one launch (round 4, csrc/chain1x1.hip): C3k2.cv1 -> C3k.cv1|cv2, C2PSA.cv1 -> qkv,
        # C3k.cv3 -> C3k2.cv2, ffn[1] -> C2PSA.cv2; from CHAIN_MIN_PIXELS pixels.  Bit-identical to the two launches and OFF by default:
        # measured 10-25 % SLOWER than them on YOLO11s at 64 images (docs/experiments.md section 0.5: a 128-pixel tile streams every
        # weight byte once per tile, 11.7 KB staged per MFLOP against 7.8 for the two launches' 256 x 256 tiles); BSY_FUSE_CHAIN=1 opts in
        self.fuse_chain = (os.environ.get("BSY_FUSE_CHAIN", "0") == "1") if fuse_chain is None else bool(fuse_chain)
        self.buf_bytes: List[int] = []
        self.ops: List[dict] = []
        self.wrecs: "OrderedDict[str, WRec]" = OrderedDict()
        self.flops = 0  # 2*MAC of every dense/depthwise conv + attention matmuls, whole batch
        self.meta: Dict = {}
        self._lane = 0
        self._build()
        if self.latency:
            self._apply_split_k()
        # Side lanes (the Detect branches on their own streams) pay a fork / join event pair each: worth it when the branch kernels
        # are long enough to overlap -- from about 12 images of 640 x 640 (measured: 0.74 vs 0.79 ms at 1 image, 0.94 vs 0.96 at 8,
        # 1.25 vs 1.24 at 16, 1.92 vs 1.87 at 32, 3.3 vs 3.2 at 64).  BSY_LANES=0 / 1 forces one or the other (0: the tests' serial
        # reference schedule).
        # `lanes` (argument): the engine's graph mode asks for them at every size -- in a captured graph fork / join are edges, not events.
        env = os.environ.get("BSY_LANES")
        if env == "0" or (env is None and not lanes and (lanes is not None or B * H * W < 12 * 640 * 640)):
            for o in self.ops:
                if "lane" in o:
                    o["lane"] = 0
        assert all(o.get("lane", 0) == 0 or o["kind"] in (L.OP_CONV, L.OP_DWCONV, L.OP_BNECK, L.OP_DWCONV_G, L.OP_DWPW, L.OP_C3K2) for o in self.ops)

    # ---- split-K (latency mode) ------------------------------------------------------------------------------------
    @staticmethod
    def split_factors(k: int, stride: int, cin: int, c0: int, cout: int, OH: int, OW: int) -> Tuple[int, int]:
        """(channel slices, tap slices) of a dense conv's K walk in latency mode -- a function of the LAYER'S SHAPE ONLY, never of
        the batch: a rank's share of a strong-scaled batch (8 images of 64) and the whole batch cut their K walks at the same places
        and add the partial sums in the same order, so shards of a batch return the bits of the whole.
        Rule (measured, 8 images of 640 x 640, r04: a split costs a second launch and an f32 round trip of the output, ~6 us, so it
        only pays on layers that are long AND thin): at least 32 K-steps of 64, fewer than 11 tiles of 128 x 128 per image, at least
        three slices; 3 x 3 layers are cut by kernel row first, then by 64-channel groups; at most 8 slices, each at least four
        64-deep K-steps.  On YOLO11s that is Detect's cv2.2.0 (3 x 3, 512 -> 64 at 20 x 20: 26.7 -> 16.6 us) and model.20 (19.8 ->
        15.3 us); splitting the 6-10-us 1 x 1 layers of the 20 x 20 stage made each of them 4-8 us SLOWER."""
        if cin % 64 or c0 % 64 or cout % 8:
            return 1, 1
        steps = k * k * cin // 64
        tiles = -(-OH * OW // 128) * -(-cout // 128)
        want = min(32 // max(tiles, 1), 8, steps // 4)
        if steps < 32 or want < 3:
            return 1, 1
        nt = 3 if k == 3 else 1
        n64 = cin // 64
        nc = max((d for d in range(1, n64 + 1) if n64 % d == 0 and d * nt <= want), default=1)
        return (nc, nt) if nc * nt >= 3 else (1, 1)

    def _apply_split_k(self) -> None:
        for o in self.ops:
            if o["kind"] != L.OP_CONV or o.get("out_f32", 0) or o.get("dst_scale", 1) != 1 or o["ksize"] not in (1, 3):
                continue
            s0, s1 = o["src0"], o.get("src1")
            cin = s0.C + (s1.C if s1 else 0)
            cout = o.get("cout", o["dst"].C)
            if o["dst"].ld % 8 or (o.get("res") is not None and o["res"].ld % 8):
                continue
            nc, nt = self.split_factors(o["ksize"], o["stride"], cin, s0.C if s1 else cin, cout, o["OH"], o["OW"])
            if nc * nt < 2:
                continue
            ldw = (cout + 31) // 32 * 32
            self.buf_bytes.append(nc * nt * self.B * o["OH"] * o["OW"] * ldw * 4)
            o["box"] = [T(len(self.buf_bytes) - 1, ldw, 0, ldw, o["OH"], o["OW"], False, True)]
            o["ksplit"] = nc | (nt << 8)

    # ---- buffers -------------------------------------------------------------------------------------------
    def alloc(self, C: int, H: int, W: int, f32: bool = False) -> T:
        if self.f32_mode:
            self.buf_bytes.append(self.B * H * W * make_divisible(C, 8) * 4)
            return T(len(self.buf_bytes) - 1, make_divisible(C, 8), 0, C, H, W, False, True)
        ld = make_divisible(C, 4) if f32 else make_divisible(C, 8)
        self.buf_bytes.append(self.B * H * W * ld * (4 if f32 else 2))
        return T(len(self.buf_bytes) - 1, ld, 0, C, H, W, False, f32)

    # ---- op emitters -----------------------------------------------------------------------------------------
    def _wrec(self, key: str, **kw) -> str:
        if key not in self.wrecs:
            self.wrecs[key] = WRec(**kw)
        return key

    def conv(self, name: str, src: Union[T, Sequence[T]], cout: int, k: int = 1, s: int = 1, act: bool = True,
             dst: Optional[T] = None, res: Optional[T] = None, plain: bool = False, out_f32: bool = False,
             perm: Optional[List[int]] = None, name2: Optional[str] = None, wkind: Optional[str] = None,
             wshape: Optional[Tuple[int, int, int]] = None, real: Tuple[int, int] = (0, 0),
             rows: Optional[List[int]] = None, cols: Optional[List[int]] = None) -> T:
        """name2: a second Conv module of the same input and kernel whose output channels follow this one's (weights.py
        kind "conv2": the two folded weight matrices stacked along cout) -- one launch for both, cout = the total."""
        srcs = [src] if isinstance(src, T) else list(src)
        assert 1 <= len(srcs) <= 2, "at most two concat operands per conv"
        H, W = srcs[0].H, srcs[0].W
        assert all(t.H == H and t.W == W and (t.f32 == self.f32_mode) for t in srcs)
        cin = sum(t.C for t in srcs)
        if not self.f32_mode and any(t.C % 8 or t.coff % 8 for t in srcs):
            # csrc/conv_mfma.hip reads its sources in 16-byte (8-channel) pieces; a block whose hidden width is 12 or 20 channels
            # (a C2f / C3k2 of 24 or 40) is refused here rather than at the first launch
            raise NotImplementedError(f"{name}: source channels / offsets {[(t.C, t.coff) for t in srcs]} are not multiples of 8 (fp16 path)")
        p = k // 2
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        if dst is None:
            dst = self.alloc(cout, OH, OW, out_f32)
        assert dst.C == cout and dst.H == OH and dst.W == OW and dst.f32 == (out_f32 or self.f32_mode) and not dst.up
        if res is not None:
            assert res.C == cout and res.H == OH and res.W == OW and not res.up
        if rows is None and cols is None and (any(t.cmap is not None for t in srcs) or dst.cmap is not None):
            # padded pieces on either side: the op's weight matrix is the module's with zero rows / columns at the padding channels
            assert perm is None and wshape is None and not real[1]
            out_real = real[0] or dst.real
            rows = list(dst.cmap) if dst.cmap is not None else [i if i < out_real else -1 for i in range(cout)]
            cols, base = [], 0
            for t in srcs:
                cols += [base + v if v >= 0 else -1 for v in (t.cmap if t.cmap is not None else range(t.C))]
                base += t.real
            real = (out_real, base)
            if real[0] < cout and dst.cmap is None:
                dst = dst.padded([(cout, real[0])])
        if name2 is not None:
            assert not plain and perm is None and cout % 2 == 0
            key = self._wrec(name + "+" + name2, name=name, kind="conv2", cout=cout, cin=cin, k=k, post=name2, rows=rows, cols=cols,
                             real_cout=real[0], real_cin=real[1])
        else:
            wc, wi, wk = wshape or (cout, cin, k)  # wshape: the module's own weight shape where the op runs a re-laid-out copy
            key = self._wrec(name, name=name, kind=wkind or ("plain" if plain else "conv"), cout=wc, cin=wi, k=wk, perm=perm,
                             real_cout=real[0], real_cin=real[1], rows=rows, cols=cols)
        self.ops.append(dict(kind=L.OP_CONV, H=H, W=W, OH=OH, OW=OW, src0=srcs[0],
                             src1=srcs[1] if len(srcs) > 1 else None, dst=dst, res=res, ksize=k, stride=s, pad=p,
                             act=int(act), out_f32=int(out_f32), wkey=key, dst_scale=1, name=name, lane=self._lane, cout=cout,
                             mfma_flops=2 * self.B * OH * OW * cout * cin * k * k))
        self.flops += 2 * self.B * OH * OW * cout * cin * k * k
        return dst

    def conv_first(self, name: str, cout: int, k: int, s: int, p: Optional[int] = None) -> T:
        p = k // 2 if p is None else p
        if (k, s, p) == (6, 2, 2) and not self.f32_mode:
            # YOLOv5u's stem (cfg/models/v5/yolov5.yaml:16): space-to-depth of the image (one elementwise launch) + an ordinary
            # 3x3 stride-1 conv over its 12 (+4 zero) channels with the weights re-laid-out to match (weights.py "first_s2d")
            assert self.H % 2 == 0 and self.W % 2 == 0
            sd = self.alloc(16, self.H // 2, self.W // 2)
            self.ops.append(dict(kind=L.OP_S2D, H=self.H, W=self.W, OH=self.H // 2, OW=self.W // 2,
                                 src0=T(L.BSY_EXT_BASE + self.EXT_IMG, 0, 0, 3, self.H, self.W), dst=sd, in_dtype=self.in_dtype,
                                 name=name + ".s2d"))
            return self.conv(name, sd, cout, 3, 1, wkind="first_s2d", wshape=(cout, 3, 6))
        OH, OW = (self.H + 2 * p - k) // s + 1, (self.W + 2 * p - k) // s + 1
        dst = self.alloc(cout, OH, OW)
        key = self._wrec(name, name=name, kind="first", cout=cout, cin=3, k=k)
        self.ops.append(dict(kind=L.OP_CONV_FIRST, H=self.H, W=self.W, OH=OH, OW=OW,
                             src0=T(L.BSY_EXT_BASE + self.EXT_IMG, 0, 0, 3, self.H, self.W), dst=dst, ksize=k, stride=s,
                             pad=p, act=1, wkey=key, in_dtype=self.in_dtype, name=name,
                             mfma_flops=2 * self.B * OH * OW * cout * 3 * k * k))
        self.flops += 2 * self.B * OH * OW * cout * 3 * k * k
        return dst

    def dwconv(self, name: str, src: T, act: bool, dst: Optional[T] = None, res: Optional[T] = None, real_c: int = 0) -> T:
        assert not src.up
        if dst is None:
            dst = self.alloc(src.C, src.H, src.W)
        key = self._wrec(name, name=name, kind="dw", cout=src.C, cin=1, k=3, real_cout=real_c)
        self.ops.append(dict(kind=L.OP_DWCONV, H=src.H, W=src.W, OH=src.H, OW=src.W, src0=src, dst=dst, res=res,
                             ksize=3, stride=1, pad=1, act=int(act), wkey=key, name=name, lane=self._lane))
        self.flops += 2 * self.B * src.H * src.W * src.C * 9
        return dst

    def dwconv_g(self, name: str, src: T, kh: int, kw: int, s: int, act_c: int, dst: Optional[T] = None, kind: str = "dwg",
                 post: Optional[str] = None, rows: Optional[List[int]] = None, real_c: int = 0) -> T:
        """Depthwise kh x kw conv with "same" padding (csrc/bsyolo_ops.hip); SiLU on the first act_c channels.
        kind: "dwg" (Conv + BN), "dwg_plain" (bare nn.Conv2d with bias), "dwg_ext" (PMSFA.conv3, see pmsfa())."""
        assert not src.up and src.f32 == self.f32_mode
        OH, OW = (src.H + 2 * (kh // 2) - kh) // s + 1, (src.W + 2 * (kw // 2) - kw) // s + 1
        if dst is None:
            dst = self.alloc(src.C, OH, OW)
        assert dst.C == src.C and dst.H == OH and dst.W == OW
        key = self._wrec(name, name=name, kind=kind, cout=src.C, cin=1, k=kh, kw=kw, post=post, rows=rows, real_cout=real_c)
        self.ops.append(dict(kind=L.OP_DWCONV_G, H=src.H, W=src.W, OH=OH, OW=OW, src0=src, dst=dst, ksize=kh, pad=kw, stride=s,
                             act=int(act_c), wkey=key, heads=src.C, key_dim=0, name=name, lane=self._lane,
                             mid_c=src.C // 2 if kind == "dwg_ext" else 0))  # first channel of the identity-kernel half
        self.flops += 2 * self.B * OH * OW * src.C * kh * kw
        return dst

    def dwpw(self, name: str, src: T, cout: int, real: Tuple[int, int] = (0, 0)) -> T:
        """nn.Sequential(DWConv(c, c, 3), Conv(c, cout, 1)) (head.py:49-57): one fused launch where conv_mfma.hip's
        dwpw_fused_kernel takes the widths, else the two ordinary launches.  real = the modules' own (cout, cin) where `cout` / `src`
        are padded to a multiple of 8."""
        if not (self.fuse_dwpw and dwpw_supported(src.C, cout) and not src.up):
            t = self.dwconv(name + ".0", src, act=True, real_c=real[1])
            return self.conv(name + ".1", t, cout, 1, 1, real=real)
        kd = self._wrec(name + ".0", name=name + ".0", kind="dw", cout=src.C, cin=1, k=3, real_cout=real[1])
        kp = self._wrec(name + ".1", name=name + ".1", kind="conv", cout=cout, cin=src.C, k=1, perm=None, real_cout=real[0], real_cin=real[1])
        dst = self.alloc(cout, src.H, src.W)
        self.ops.append(dict(kind=L.OP_DWPW, H=src.H, W=src.W, OH=src.H, OW=src.W, src0=src, dst=dst, ksize=3, stride=1, pad=1,
                             act=1, wkey=kd, wkey2=kp, name=name, lane=self._lane,
                             mfma_flops=2 * self.B * src.H * src.W * src.C * cout))  # the dense 1x1 part (the depthwise is VALU work)
        self.flops += 2 * self.B * src.H * src.W * src.C * (9 + cout)
        return dst

    def copy(self, name: str, src: T, dst: T):
        assert dst.C == src.C and dst.H == src.H and dst.W == src.W and not dst.up
        self.ops.append(dict(kind=L.OP_COPY, H=src.H, W=src.W, OH=src.H, OW=src.W, src0=src, dst=dst, name=name, lane=self._lane))

    # ---- module expansions -----------------------------------------------------------------------------------
    def bottleneck(self, name: str, x: T, dst: T, shortcut: bool, k=(3, 3), e=0.5):
        """block.py:3405-3419: x + cv2(cv1(x))."""
        c_ = int(dst.real * e)
        if (self.fuse_bneck and tuple(k) == (3, 3) and shortcut and x.C == dst.C and not x.up and not x.f32 and x.cmap is None and dst.cmap is None
                and bneck_supported(x.C, c_)):
            # one launch, hidden map kept in LDS (csrc/bneck_fused.hip)
            k1 = self._wrec(name + ".cv1", name=name + ".cv1", kind="conv", cout=c_, cin=x.C, k=3, perm=None)
            k2 = self._wrec(name + ".cv2", name=name + ".cv2", kind="conv", cout=dst.C, cin=c_, k=3, perm=None)
            self.ops.append(dict(kind=L.OP_BNECK, H=x.H, W=x.W, OH=x.H, OW=x.W, src0=x, dst=dst, ksize=3, stride=1,
                                 pad=1, act=1, wkey=k1, wkey2=k2, mid_c=c_, name=name, lane=self._lane,
                                 mfma_flops=2 * self.B * x.H * x.W * 9 * 2 * c_ * x.C))
            self.flops += 2 * self.B * x.H * x.W * 9 * 2 * c_ * x.C
            return
        res = x if (shortcut and x.C == dst.C) else None
        if c_ % 8 and not self.f32_mode:
            # hidden width 4 / 12 / 20 (a C3k2 of 16 / 48 / 80 channels with e = 0.25 x 0.5, e.g. YOLO11 at width 0.125 / 0.375): the
            # hidden map is carried on the next multiple of 8 with zero weights and bias in the padding (weights.py real_cout /
            # real_cin, as Detect's class branch does for nc-dependent widths) -- SiLU(0) = 0 there, the real channels are unchanged
            c8 = make_divisible(c_, 8)
            if x.cmap is not None or dst.cmap is not None:  # inside a block whose chunks are padded pieces themselves (c2f below)
                t = self.conv(name + ".cv1", x, c8, k[0], 1, dst=self.alloc(c8, x.H, x.W).padded([(c8, c_)]))
                self.conv(name + ".cv2", t, dst.C, k[1], 1, dst=dst, res=res)
                return
            t = self.conv(name + ".cv1", x, c8, k[0], 1, real=(c_, 0))
            self.conv(name + ".cv2", t, dst.C, k[1], 1, dst=dst, res=res, real=(0, c_))
            return
        t = self.conv(name + ".cv1", x, c_, k[0], 1)
        self.conv(name + ".cv2", t, dst.C, k[1], 1, dst=dst, res=res)

    def c3k(self, name: str, x, dst: T, n: int, shortcut: bool, k=(3, 3)):
        """block.py:3320-3334 + :3807-3815: cv3(cat(m(cv1 x), cv2 x)), m = n x Bottleneck(c_, c_, k=(3,3), e=1); C3 itself
        (YOLOv5u) is the same with k = ((1,1), (3,3)) and may read a two-operand Concat."""
        cr = int(dst.real * 0.5)  # the module's hidden width
        x0 = x if isinstance(x, T) else x[0]
        # fp16 path, hidden width not a multiple of 8 (a C3k of 12 / 20 / 24 channels ...): both halves of the concat are zero-padded pieces
        pad = bool(cr % 8) and not self.f32_mode
        c_ = make_divisible(cr, 8) if pad else cr
        new = (lambda: self.alloc(c_, x0.H, x0.W).padded([(c_, cr)])) if pad else (lambda: self.alloc(c_, x0.H, x0.W))
        cat = self.alloc(2 * c_, x0.H, x0.W)
        if pad:
            cat = cat.padded([(c_, cr), (c_, cr)])
        if self.merge_c3k and n >= 2:
            # cv1 and cv2 read the same x: ONE launch writes [cv1 x | cv2 x] into the concat buffer (twice the cout per
            # pixel tile, one launch less); the last bottleneck then overwrites the cv1 half, which only the first one reads
            self.conv(name + ".cv1", x, 2 * c_, 1, 1, dst=cat, name2=name + ".cv2")
            cur = cat.slice(0, c_)
        else:
            cur = self.conv(name + ".cv1", x, c_, 1, 1, dst=new())
            self.conv(name + ".cv2", x, c_, 1, 1, dst=cat.slice(c_, c_))
        for i in range(n):
            out = cat.slice(0, c_) if i == n - 1 else new()
            self.bottleneck(f"{name}.m.{i}", cur, out, shortcut, k, 1.0)
            cur = out
        self.conv(name + ".cv3", cat, dst.C, 1, 1, dst=dst)

    def c2f(self, name: str, x, c2: int, n: int, shortcut: bool, e: float, inner: str) -> T:
        """block.py:3295-3317 (C2f) / :3796-3804 (C3k2).  inner: 'c2f' | 'bottleneck' | 'c3k'."""
        xs = [x] if isinstance(x, T) else list(x)
        c = int(c2 * e)
        if (self.fuse_tail and inner == "bottleneck" and n == 1 and shortcut and len(xs) == 1 and not xs[0].up and not xs[0].f32
                and c3k2_supported(xs[0].C, c, c2)):
            # the whole block in one launch (csrc/c3k2_fused.hip): x in, out out, the [y0 | y1 | y2] concat never reaches HBM
            names = [name + ".cv1", name + ".m.0.cv1", name + ".m.0.cv2", name + ".cv2"]
            shapes = [(2 * c, xs[0].C, 1), (c // 2, c, 3), (c, c // 2, 3), (c2, 3 * c, 1)]
            keys = [self._wrec(nm, name=nm, kind="conv", cout=co, cin=ci, k=k, perm=None) for nm, (co, ci, k) in zip(names, shapes)]
            dst = self.alloc(c2, xs[0].H, xs[0].W)
            fl = 2 * self.B * xs[0].H * xs[0].W * sum(co * ci * k * k for co, ci, k in shapes)
            self.ops.append(dict(kind=L.OP_C3K2, H=xs[0].H, W=xs[0].W, OH=xs[0].H, OW=xs[0].W, src0=xs[0], dst=dst, ksize=3, stride=1, pad=1,
                                 act=1, wkeys=keys, mid_c=c, name=name, lane=self._lane, mfma_flops=fl))
            self.flops += fl
            return dst
        # fp16 path, chunk width not a multiple of 8 (c = 12 / 20 ...; tasks.py:1016 scales c2 by any width multiple): every chunk of the
        # concat is carried on the next multiple of 8 as a zero-padded piece (T.cmap) -- cv1 writes zero rows there, cv2 reads zero columns
        cr = c
        if cr % 8 and not self.f32_mode and inner in ("bottleneck", "c2f", "c3k"):
            c = make_divisible(cr, 8)
        cat = self.alloc((2 + n) * c, xs[0].H, xs[0].W)
        if c != cr:
            cat = cat.padded([(c, cr)] * (2 + n))
        self.conv(name + ".cv1", xs, 2 * c, 1, 1, dst=cat.slice(0, 2 * c))
        for i in range(n):
            src, dst = cat.slice((1 + i) * c, c), cat.slice((2 + i) * c, c)
            if inner == "pmsfa":
                self.pmsfa(f"{name}.m.{i}", src, dst)
            elif inner == "c3k_gai":
                self.c3k_gai(f"{name}.m.{i}", src, dst, 2)
            elif inner == "c3k":
                self.c3k(f"{name}.m.{i}", src, dst, 2, shortcut)
            elif inner == "c2f":
                self.bottleneck(f"{name}.m.{i}", src, dst, shortcut, (3, 3), 1.0)
            else:
                self.bottleneck(f"{name}.m.{i}", src, dst, shortcut, (3, 3), 0.5)
        return self.conv(name + ".cv2", cat, c2, 1, 1)

    def pmsfa(self, name: str, x: T, dst: T):
        """block.py:3035-3054.  conv1 -> P = [p1 | p2]; depthwise 5x5 on p1 -> Q = [q1 | q2]; the depthwise 7x7 runs on ALL
        of Q with weights extended by an identity kernel for the q2 half (weights.py kind "dwg_ext": exact pass-through,
        SiLU only on the q1 half) -> S = [conv3(q1) | q2]; conv4 reads the virtual concat [S | p2] = the reference's
        cat([conv3_out, conv2_out_2, conv1_out_2]) and adds x."""
        c = x.C
        assert dst.C == c
        if c % 4:  # the reference's constructor needs it too (Conv(inc // 4, inc // 4, 7, g=inc // 4) on half of a half)
            raise NotImplementedError(f"PMSFA width {c}: must be a multiple of 4")
        if c % 8 and not self.f32_mode:  # the fp16 conv kernels read their sources in 8-channel pieces (so do the blocks around a PMSFA)
            raise NotImplementedError(f"PMSFA width {c}: must be a multiple of 8 on the fp16 path")
        if c % 16:
            return self._pmsfa_padded(name, x, dst)
        P = self.conv(name + ".conv1", x, c, 3, 1)
        if self.fuse_pmsfa and pmsfa_tail_supported(c) and not x.up:
            # everything after conv1 in one launch: the map is read once and written once (csrc/pmsfa_fused.hip); the weight records are
            # the unfused plan's, in its order (blob layout, synth_state_dict's random draws)
            k2 = self._wrec(name + ".conv2", name=name + ".conv2", kind="dwg", cout=c // 2, cin=1, k=5, kw=5, post=None, rows=None, real_cout=0)
            k3 = self._wrec(name + ".conv3", name=name + ".conv3", kind="dwg_ext", cout=c // 2, cin=1, k=7, kw=7, post=None, rows=None, real_cout=0)
            k4 = self._wrec(name + ".conv4", name=name + ".conv4", kind="conv", cout=c, cin=c, k=1, perm=None)
            self.ops.append(dict(kind=L.OP_PMSFA_TAIL, H=x.H, W=x.W, OH=x.H, OW=x.W, src0=P, res=x, dst=dst, ksize=1, stride=1, pad=0, act=1,
                                 wkeys=[k2, k3, k4], heads=c // 2, key_dim=c // 2, name=name + ".tail", lane=self._lane,
                                 mfma_flops=2 * self.B * x.H * x.W * c * c))
            self.flops += 2 * self.B * x.H * x.W * (c // 2 * 25 + c // 2 * 49 + c * c)  # what the three launches count
            return
        Q = self.dwconv_g(name + ".conv2", P.slice(0, c // 2), 5, 5, 1, c // 2)
        # S overwrites p1 (dead once conv2 has read it): conv4 then reads ONE contiguous source [S | p2] = P -- at c = 32 two 16-channel
        # sources put it on the unaligned configuration (0.10 ms at 160 x 160, B = 64); same K order, same bits
        S = self.dwconv_g(name + ".conv3", Q, 7, 7, 1, c // 4, kind="dwg_ext", dst=P.slice(0, c // 2))
        self.conv(name + ".conv4", P, c, 1, 1, dst=dst, res=x)

    def _pmsfa_padded(self, name: str, x: T, dst: T):
        """PMSFA on a width that is not a multiple of 16: the depthwise kernels work on 8-channel pieces, so the three pieces of
        conv1's output -- p1a, p1b (the halves of the depthwise 5x5's input) and p2 -- each start at a multiple of 8: conv1 writes
        P' = [p1a 0.. | p1b 0.. | p2 0..] (weights.py `rows`: zero weights and bias in the padding, so the padding channels are
        SiLU(0) = 0 and stay 0 through the depthwise convs), and conv4 reads [S' | p2'] with zero weights on the padding (`cols`).
        The real channels see exactly the reference's arithmetic."""
        c = x.C
        q, h = c // 4, c // 2
        Q8, H8 = make_divisible(q, 8), make_divisible(h, 8)
        pad = lambda real, width, base: [base + i if i < real else -1 for i in range(width)]
        lay = pad(q, Q8, 0) + pad(q, Q8, q) + pad(h, H8, h)           # P' channel -> conv1 output channel
        P = self.conv(name + ".conv1", x, len(lay), 3, 1, rows=lay, real=(c, 0))
        Q = self.dwconv_g(name + ".conv2", P.slice(0, 2 * Q8), 5, 5, 1, 2 * Q8, rows=pad(q, Q8, 0) + pad(q, Q8, q), real_c=h)
        S = self.dwconv_g(name + ".conv3", Q, 7, 7, 1, Q8, kind="dwg_ext", rows=pad(q, Q8, 0), real_c=q, dst=P.slice(0, 2 * Q8))
        self.conv(name + ".conv4", P, c, 1, 1, dst=dst, res=x, cols=lay, real=(0, c))

    def c3k_gai(self, name: str, x: T, dst: T, n: int):
        """block.py:3079-3086: C3 (cv3(cat(m(cv1 x), cv2 x))) with m = n x PMSFA(c_)."""
        c_ = int(dst.C * 0.5)
        cat = self.alloc(2 * c_, x.H, x.W)
        cur = self.conv(name + ".cv1", x, c_, 1, 1)
        self.conv(name + ".cv2", x, c_, 1, 1, dst=cat.slice(c_, c_))
        for i in range(n):
            out = cat.slice(0, c_) if i == n - 1 else self.alloc(c_, x.H, x.W)
            self.pmsfa(f"{name}.m.{i}", cur, out)
            cur = out
        self.conv(name + ".cv3", cat, dst.C, 1, 1, dst=dst)

    def scdown(self, name: str, x: T, c2: int, k: int, s: int) -> T:
        """block.py:4503-4535."""
        t = self.conv(name + ".cv1", x, c2, 1, 1)
        return self.dwconv_g(name + ".cv2", t, k, k, s, 0)

    def msca(self, name: str, x: T) -> T:
        """nn/Addmodules/MSCA.py:19-88.  The depthwise 1x1 `dilconv` that follows the first three branches' column convs is
        folded into their weights; the four SE convs run on (B, 1, 1, C) maps through the ordinary 1x1 conv kernel."""
        assert not x.up
        C, H, W = x.C, x.H, x.W
        strips = ((5, "conv0"), (7, "conv1"), (11, "conv2"), (21, "conv3"))
        branches, logits = [], []
        if self.fuse_msca and msca_spatial_supported(H, W):
            # one launch for the nine depthwise convs and the four global means (csrc/bsyolo_ops.hip msca_spatial_kernel)
            keys = [self._wrec(name + ".conv0", name=name + ".conv0", kind="dwg_plain", cout=C, cin=1, k=5, kw=5)]
            for i, (k, base) in enumerate(strips):
                keys.append(self._wrec(f"{name}.{base}_1", name=f"{name}.{base}_1", kind="dwg_plain", cout=C, cin=1, k=1, kw=k))
                keys.append(self._wrec(f"{name}.{base}_2", name=f"{name}.{base}_2", kind="dwg_plain", cout=C, cin=1, k=k, kw=1,
                                       post=(name + ".dilconv") if i < 3 else None))
                se = f"{name}.SE{i + 1}.conv.0"  # registered here so the records keep the unfused plan's order (blob layout,
                self._wrec(se, name=se, kind="plain", cout=C, cin=C, k=1, perm=None)  # synth_state_dict's random draws)
            branches = [self.alloc(C, H, W) for _ in strips]
            gaps = [self.alloc(C, 1, 1) for _ in strips]
            self.ops.append(dict(kind=L.OP_MSCA_SPATIAL, H=H, W=W, OH=H, OW=W, src0=x, box=branches[:3], res=branches[3],
                                 cls=gaps[:3], msk=[gaps[3]], wkeys=keys, name=name + ".spatial", lane=self._lane))
            self.flops += 2 * self.B * H * W * C * (25 + 2 * sum(k for k, _ in strips))
            for i, g in enumerate(gaps):
                logits.append(self.conv(f"{name}.SE{i + 1}.conv.0", g, C, 1, 1, act=False, plain=True, out_f32=True))
        else:
            attn = self.dwconv_g(name + ".conv0", x, 5, 5, 1, 0, kind="dwg_plain")
            for i, (k, base) in enumerate(strips):
                a = self.dwconv_g(f"{name}.{base}_1", attn, 1, k, 1, 0, kind="dwg_plain")
                a = self.dwconv_g(f"{name}.{base}_2", a, k, 1, 1, 0, kind="dwg_plain", post=(name + ".dilconv") if i < 3 else None)
                branches.append(a)
                g = self.alloc(C, 1, 1)
                self.ops.append(dict(kind=L.OP_GAP, H=H, W=W, OH=1, OW=1, src0=a, dst=g, name=f"{name}.gap{i}", lane=self._lane))
                logits.append(self.conv(f"{name}.SE{i + 1}.conv.0", g, C, 1, 1, act=False, plain=True, out_f32=True))
        mix = self.alloc(C, H, W)
        self.ops.append(dict(kind=L.OP_MSCA_MIX, H=H, W=W, OH=H, OW=W, dst=mix, box=branches[:3], res=branches[3], cls=logits[:3],
                             msk=[logits[3]], name=name + ".mix", lane=self._lane))
        t = self.conv(name + ".conv4", mix, C, 1, 1, act=False, plain=True)
        out = self.alloc(C, H, W)
        self.ops.append(dict(kind=L.OP_MUL, H=H, W=W, OH=H, OW=W, src0=t, src1=x, dst=out, name=name + ".mul", lane=self._lane))
        return out

    def ela(self, name: str, x: T) -> T:
        """nn/Addmodules/ELA.py:33-101."""
        assert not x.up
        if x.C % 8 or x.C % max(1, x.C // 16):  # GroupNorm(max(1, C // 16), C): the reference's constructor needs the second
            raise NotImplementedError(f"ELA on {x.C} channels: must be a multiple of 8 that splits into {x.C // 16} groups")
        k = int(abs((math.log(x.C, 2) + 1) / 2))
        k = k if k % 2 else k + 1
        key = self._wrec(name, name=name, kind="ela", cout=x.C, cin=1, k=k)
        self.buf_bytes.append(self.B * (2 * (x.H + x.W) + 2) * x.C * 4)
        scratch = T(len(self.buf_bytes) - 1, x.C, 0, x.C, 1, 1, False, True)
        dst = self.alloc(x.C, x.H, x.W)
        self.ops.append(dict(kind=L.OP_ELA, H=x.H, W=x.W, OH=x.H, OW=x.W, src0=x, dst=dst, res=scratch, ksize=k, wkey=key,
                             name=name, lane=self._lane))
        return dst

    def sppf(self, name: str, x: T, c2: int) -> T:
        """block.py:3114-3149."""
        c_ = x.C // 2
        cat = self.alloc(4 * c_, x.H, x.W)
        self.conv(name + ".cv1", x, c_, 1, 1, dst=cat.slice(0, c_))
        self.ops.append(dict(kind=L.OP_SPPF_POOL, H=x.H, W=x.W, OH=x.H, OW=x.W, src0=cat.slice(0, c_), name=name + ".m"))
        return self.conv(name + ".cv2", cat, c2, 1, 1)

    def c2psa(self, name: str, x: T, c2: int, n: int) -> T:
        """block.py:4429-4468 with PSABlock :4348-4383 and Attention :4235-4288."""
        assert x.C == c2
        c = int(c2 * 0.5)
        nh = c // 64
        hd = c // nh
        kd = int(hd * 0.5)
        if not self.f32_mode and not attention_supported(kd, hd):
            # e.g. a width multiple of 0.3125: heads of 40 / 80 channels; the fp32-storage modes run any head shape
            raise NotImplementedError(f"{name}: attention heads of key_dim {kd} / head_dim {hd} are not on the fp16 path (16..64 in steps of 16 / 32..128 in steps of 32)")
        ab = self.alloc(2 * c, x.H, x.W)
        self.conv(name + ".cv1", x, 2 * c, 1, 1, dst=ab)
        b = ab.slice(c, c)
        # qkv rows are emitted per head as [q(kd) k(kd) v(hd)] by the reference (block.py:4274-4276); reorder to
        # [q(all heads) | k(all heads) | v(all heads)] so that v is one channel slice and heads are contiguous
        per = 2 * kd + hd
        perm = [h * per + i for h in range(nh) for i in range(kd)] + \
               [h * per + kd + i for h in range(nh) for i in range(kd)] + \
               [h * per + 2 * kd + i for h in range(nh) for i in range(hd)]
        for i in range(n):
            pn = f"{name}.m.{i}"
            qkv = self.conv(pn + ".attn.qkv", b, nh * per, 1, 1, act=False, perm=perm)
            att = self.alloc(c, x.H, x.W)
            self.ops.append(dict(kind=L.OP_ATTN, H=x.H, W=x.W, OH=x.H, OW=x.W, src0=qkv, dst=att, heads=nh, key_dim=kd,
                                 head_dim=hd, scale=float(kd ** -0.5), name=pn + ".attn"))
            N = x.H * x.W
            self.flops += 2 * self.B * nh * N * N * (kd + hd)
            xo = self.dwconv(pn + ".attn.pe", qkv.slice(2 * nh * kd, c), act=False, res=att)  # v@attn^T + pe(v)
            self.conv(pn + ".attn.proj", xo, c, 1, 1, act=False, dst=b, res=b)                # b = b + proj(.)
            f = self.conv(pn + ".ffn.0", b, 2 * c, 1, 1)
            self.conv(pn + ".ffn.1", f, c, 1, 1, act=False, dst=b, res=b)                     # b = b + ffn(b)
        return self.conv(name + ".cv2", ab, c2, 1, 1)

    def proto(self, name: str, x: T, c_: int, nm: int) -> T:
        """block.py:80-97 Proto: cv3(cv2(ConvTranspose2d(c_, c_, 2, 2)(cv1(x)))) -> (B, nm, 2H, 2W)."""
        t = self.conv(name + ".cv1", x, c_, 3, 1)
        up = self.alloc(c_, 2 * x.H, 2 * x.W)
        for dy in (0, 1):
            for dx in (0, 1):
                key = f"{name}.upsample@{dy}{dx}"
                if key not in self.wrecs:
                    self.wrecs[key] = WRec(name=name + ".upsample", kind="deconv", cout=c_, cin=c_, k=1, tap=(dy, dx))
                self.ops.append(dict(kind=L.OP_CONV, H=x.H, W=x.W, OH=x.H, OW=x.W, src0=t, src1=None,
                                     dst=T(up.buf, up.ld, 0, c_, x.H, x.W, False, up.f32), res=None, ksize=1, stride=1, pad=0, act=0,
                                     out_f32=0, wkey=key, dst_scale=2, dst_dy=dy, dst_dx=dx, name=key,
                                     mfma_flops=2 * self.B * x.H * x.W * c_ * c_))
                self.flops += 2 * self.B * x.H * x.W * c_ * c_
        t = self.conv(name + ".cv2", up, c_, 3, 1)
        p = self.conv(name + ".cv3", t, nm, 1, 1)
        self.ops.append(dict(kind=L.OP_NHWC2NCHW, H=p.H, W=p.W, OH=0, OW=0, src0=p,
                             dst=T(L.BSY_EXT_BASE + self.EXT_PROTO, 0, 0, nm, 0, 0), out_dtype=self.out_dtype,
                             name=name + ".nchw"))
        self.meta.update(proto_hw=(p.H, p.W))
        return p

    def _head_conv(self, name: str, src: T, cout: int, mode: int, level: int, a0: int, A: int, nc: int, stride: float, real_cin: int = 0):
        """Last 1x1 conv of a Detect branch with the decoder fused into its epilogue (out_f32 = mode 2 / 3)."""
        key = self._wrec(name, name=name, kind="plain", cout=cout, cin=src.C, k=1, perm=None, real_cin=real_cin)
        y = T(L.BSY_EXT_BASE + self.EXT_Y, 0, 0, 4 + nc, 0, 0)
        raw = T(L.BSY_EXT_BASE + self.EXT_RAW0 + level, 0, 0, 64 + nc, 0, 0)
        self.ops.append(dict(kind=L.OP_CONV, H=src.H, W=src.W, OH=src.H, OW=src.W, src0=src, src1=None, dst=y, res=None,
                             ksize=1, stride=1, pad=0, act=0, out_f32=mode, wkey=key, dst_scale=1, name=name, cout=cout,
                             lane=self._lane, nl=cout, nc=nc, nm=0, A=A, box=[raw], cls=[], msk=[], level=level,
                             lvl_h=[src.H, a0], lvl_w=[src.W], lvl_stride=[stride], out_dtype=self.out_dtype,
                             mfma_flops=2 * self.B * src.H * src.W * cout * src.C))
        self.flops += 2 * self.B * src.H * src.W * cout * src.C

    def _box_tail(self, name: str, i: int, src: T, level: int, a0: int, A: int, nc: int, stride: float):
        """cv2.i.1 (3x3, 64 -> 64, SiLU) + cv2.i.2 (1x1, 64 -> 64 box logits) + DFL / dist2bbox (head.py:49-57, :141-146) as ONE
        conv op: the patch kernel's TAIL form (conv_mfma.hip) multiplies its activated fp16 tile with the 1x1 weights straight from
        LDS and decodes the accumulators -- bit for bit what the two launches produce, without the 64-channel map's round trip."""
        n1, n2 = f"{name}.cv2.{i}.1", f"{name}.cv2.{i}.2"
        k1 = self._wrec(n1, name=n1, kind="conv", cout=64, cin=src.C, k=3, perm=None)
        k2 = self._wrec(n2, name=n2, kind="plain", cout=64, cin=64, k=1, perm=None)
        y = T(L.BSY_EXT_BASE + self.EXT_Y, 0, 0, 4 + nc, 0, 0)
        raw = T(L.BSY_EXT_BASE + self.EXT_RAW0 + level, 0, 0, 64 + nc, 0, 0)
        fl = 2 * self.B * src.H * src.W * 64 * (9 * src.C + 64)
        self.ops.append(dict(kind=L.OP_CONV, H=src.H, W=src.W, OH=src.H, OW=src.W, src0=src, src1=None, dst=y, res=None,
                             ksize=3, stride=1, pad=1, act=1, out_f32=3, wkey=k1, wkey2=k2, mid_c=64, dst_scale=1, name=n1 + "+2",
                             cout=64, lane=self._lane, nl=64, nc=nc, nm=0, A=A, box=[raw], cls=[], msk=[], level=level,
                             lvl_h=[src.H, a0], lvl_w=[src.W], lvl_stride=[stride], out_dtype=self.out_dtype, mfma_flops=fl))
        self.flops += fl

    def detect(self, name: str, xs: List[T], nc: int, legacy: bool, nm: int = 0, npr: int = 0):
        """head.py:21-148 (+ Segment :175-197)."""
        ch = [t.C for t in xs]
        c2, c3 = max(16, ch[0] // 4, 64), max(ch[0], min(nc, 100))
        boxes, clss, msks = [], [], []
        A = sum(t.H * t.W for t in xs)
        a0 = [sum(t.H * t.W for t in xs[:i]) for i in range(len(xs))]
        strides = [float(self.H // t.H) for t in xs]
        # Detect without mask coefficients: the last conv of every branch decodes in its epilogue (conv_mfma.hip
        # conv_epilogue_head) -- no f32 logit maps, no decode launch, no raw-map transposes
        fused = self.fuse_head and nm == 0 and len(xs) <= 3
        # every (level, branch) chain is independent until the decoder: give each its own lane so the engine runs them
        # concurrently (the 40x40 / 20x20 chains are far too small to fill 256 CUs on their own)
        for i, x in enumerate(xs):
            assert not x.up
            self._lane = 2 * i
            t = self.conv(f"{name}.cv2.{i}.0", x, c2, 3, 1)
            if fused and self.fuse_boxtail and c2 == 64:
                self._box_tail(name, i, t, i, a0[i], A, nc, strides[i])
            else:
                t = self.conv(f"{name}.cv2.{i}.1", t, c2, 3, 1)
                if fused:
                    self._head_conv(f"{name}.cv2.{i}.2", t, 64, 3, i, a0[i], A, nc, strides[i])
                else:
                    boxes.append(self.conv(f"{name}.cv2.{i}.2", t, 64, 1, 1, act=False, plain=True, out_f32=True))
            self._lane = 2 * i + 1
            # c3 = max(ch0, min(nc, 100)) (head.py:39) need not be a multiple of 8 (YOLO11n with 65 .. 100 or more classes: 100): the
            # branch then runs c3p channels wide with zero weights / biases in the padding -- SiLU(0) = 0 all the way, same results
            c3p = make_divisible(c3, 8)
            r3 = c3 if c3p != c3 else 0
            if legacy:
                t = self.conv(f"{name}.cv3.{i}.0", x, c3p, 3, 1, real=(r3, 0))
                t = self.conv(f"{name}.cv3.{i}.1", t, c3p, 3, 1, real=(r3, r3))
            else:
                t = self.dwpw(f"{name}.cv3.{i}.0", x, c3p, real=(r3, 0))
                t = self.dwpw(f"{name}.cv3.{i}.1", t, c3p, real=(r3, r3))
            if fused:
                self._head_conv(f"{name}.cv3.{i}.2", t, nc, 2, i, a0[i], A, nc, strides[i], real_cin=r3)
            else:
                clss.append(self.conv(f"{name}.cv3.{i}.2", t, nc, 1, 1, act=False, plain=True, out_f32=True, real=(0, r3)))
            if nm:
                self._lane = 2 * len(xs) + i
                c4 = max(ch[0] // 4, nm)
                t = self.conv(f"{name}.cv4.{i}.0", x, c4, 3, 1)
                t = self.conv(f"{name}.cv4.{i}.1", t, c4, 3, 1)
                msks.append(self.conv(f"{name}.cv4.{i}.2", t, nm, 1, 1, act=False, plain=True, out_f32=True))
        self._lane = 0
        self.meta.update(A=A, nc=nc, nm=nm, no=64 + nc, strides=strides, levels=[(t.H, t.W) for t in xs])
        if fused:
            return
        self.ops.append(dict(kind=L.OP_DECODE, join=1, H=self.H, W=self.W, OH=0, OW=0, nl=len(xs), nc=nc, nm=nm, A=A, box=boxes,
                             cls=clss, msk=msks, lvl_h=[t.H for t in xs], lvl_w=[t.W for t in xs], lvl_stride=strides,
                             dst=T(L.BSY_EXT_BASE + self.EXT_Y, 0, 0, 4 + nc + nm, 0, 0), out_dtype=self.out_dtype,
                             name=name + ".decode"))
        for i in range(len(xs)):
            self.ops.append(dict(kind=L.OP_RAW_NCHW, H=self.H, W=self.W, OH=0, OW=0, nl=len(xs), nc=nc, nm=0, A=A,
                                 box=boxes, cls=clss, msk=[], lvl_h=[t.H for t in xs], lvl_w=[t.W for t in xs],
                                 lvl_stride=strides, level=i,
                                 dst=T(L.BSY_EXT_BASE + self.EXT_RAW0 + i, 0, 0, 64 + nc, 0, 0),
                                 out_dtype=self.out_dtype, name=f"{name}.raw{i}"))
        self.meta.update(A=A, nc=nc, nm=nm, no=64 + nc, strides=strides, levels=[(t.H, t.W) for t in xs])

    # ---- graph walk (nn/tasks.py:138-165 + :940-1105) -----------------------------------------------------------
    def _build(self):
        d = self.cfg
        scales = d.get("scales")
        depth, width, max_ch = d.get("depth_multiple", 1.0), d.get("width_multiple", 1.0), float("inf")
        scale = d.get("scale")
        if scales:
            if not scale:
                scale = next(iter(scales))
            depth, width, max_ch = scales[scale]
        nc = d["nc"]
        if d.get("activation"):
            raise NotImplementedError("custom activation (tasks.py:956-957) is not accelerated")
        legacy = True
        outs: List = []   # per layer: T, or list[T] for a lazy Concat
        chans: List[int] = []
        x = None
        layers = list(d["backbone"]) + list(d["head"])
        for i, (f, n, m, args) in enumerate(layers):
            name = f"model.{i}"
            args = [nc if a == "nc" else a for a in args]
            n = max(round(n * depth), 1) if n > 1 else n
            if f != -1:
                x = outs[f] if isinstance(f, int) else [x if j == -1 else outs[j] for j in f]
            if m in ("Conv", "C3k2", "C2f", "SPPF", "C2PSA", "C3k2_gai", "SCDown", "C3", "DWConv"):
                c2 = make_divisible(min(args[0], max_ch) * width, 8)
                if m == "Conv":
                    k = args[1] if len(args) > 1 else 1
                    s = args[2] if len(args) > 2 else 1
                    pad = args[3] if len(args) > 3 and args[3] is not None else k // 2
                    if i == 0:
                        if (k, s, pad) not in ((3, 2, 1), (6, 2, 2)):
                            raise NotImplementedError(f"image conv k={k} s={s} p={pad}: 3x3 s2 p1 or 6x6 s2 p2 only")
                        y = self.conv_first(name, c2, k, s, pad)
                    else:
                        assert isinstance(x, T)
                        if pad != k // 2:
                            raise NotImplementedError(f"Conv with padding {pad} != k // 2 is not accelerated")
                        y = self.conv(name, x, c2, k, s)
                elif m == "C3":  # block.py:3320-3334; parse_model inserts the repeat count (tasks.py:1038)
                    xs = x if isinstance(x, list) else x
                    shortcut = bool(args[1]) if len(args) > 1 else True
                    x0 = xs[0] if isinstance(xs, list) else xs
                    y = self.alloc(c2, x0.H, x0.W)
                    self.c3k(name, xs, y, n, shortcut, k=(1, 3))
                elif m == "DWConv":  # conv.py:224-229: Conv(c1, c2, k, s, g=gcd(c1, c2))
                    assert isinstance(x, T)
                    k = args[1] if len(args) > 1 else 1
                    s = args[2] if len(args) > 2 else 1
                    if c2 != x.C:
                        raise NotImplementedError(f"DWConv {x.C} -> {c2}: only the depthwise case c1 == c2 is accelerated")
                    y = self.dwconv_g(name, x, k, k, s, x.C)
                elif m == "C3k2":
                    legacy = False
                    c3k = bool(args[1]) if len(args) > 1 else False
                    e = args[2] if len(args) > 2 else 0.5
                    if scale and scale in "mlx":
                        c3k = True
                    y = self.c2f(name, x, c2, n, True, e, "c3k" if c3k else "bottleneck")
                elif m == "C3k2_gai":  # tasks.py:1038: repeat count inserted, no m/l/x override, `legacy` untouched
                    c3k = bool(args[1]) if len(args) > 1 else False
                    e = args[2] if len(args) > 2 else 0.5
                    y = self.c2f(name, x, c2, n, True, e, "c3k_gai" if c3k else "pmsfa")
                elif m == "SCDown":
                    assert isinstance(x, T)
                    y = self.scdown(name, x, c2, args[1], args[2])
                elif m == "C2f":
                    shortcut = bool(args[1]) if len(args) > 1 else False
                    y = self.c2f(name, x, c2, n, shortcut, 0.5, "c2f")
                elif m == "SPPF":
                    assert isinstance(x, T) and (len(args) < 2 or args[1] == 5)
                    y = self.sppf(name, x, c2)
                else:
                    assert isinstance(x, T)
                    y = self.c2psa(name, x, c2, n)
                cout = c2
            elif m == "MSCAAttention":  # tasks.py:1052-1054
                assert isinstance(x, T)
                y = self.msca(name, x)
                cout = x.C
            elif m == "ELA":  # tasks.py:1066-1070: built on the input's channels
                assert isinstance(x, T)
                y = self.ela(name, x)
                cout = x.C
            elif m == "nn.Upsample":
                assert isinstance(x, T) and not x.up and args[1] == 2 and args[2] == "nearest"
                y = T(x.buf, x.ld, x.coff, x.C, x.H * 2, x.W * 2, True, x.f32)
                cout = x.C
            elif m == "Concat":
                assert isinstance(x, list) and len(x) == 2, "only two-operand Concat feeding a conv is accelerated"
                # an operand that is itself a (lazy) Concat -- BS-YOLO's layer 21 reads layer 13 -- is materialised once
                for j, t in enumerate(x):
                    if isinstance(t, list):
                        assert all(isinstance(u, T) for u in t)
                        real = self.alloc(sum(u.C for u in t), t[0].H, t[0].W)
                        c0 = 0
                        for u in t:
                            self.copy(f"{name}.cat{j}", u, real.slice(c0, u.C))
                            c0 += u.C
                        x[j] = real
                assert all(isinstance(t, T) for t in x)
                y = list(x)
                cout = sum(t.C for t in x)
            elif m in ("Detect", "Segment"):
                xs = x if isinstance(x, list) else [x]
                assert all(isinstance(t, T) for t in xs)
                if m == "Segment":  # head.py:175-197; npr scaled like parse_model does (tasks.py:1082-1083)
                    nm = args[1]
                    npr = make_divisible(min(args[2], max_ch) * width, 8)
                    self.proto(name + ".proto", xs[0], npr, nm)
                    self.detect(name, xs, args[0], legacy, nm=nm)
                else:
                    self.detect(name, xs, args[0], legacy)
                y, cout = None, 0
            else:
                raise NotImplementedError(f"module {m} is not on the accelerated path")
            outs.append(y)
            chans.append(cout)
            x = y
        self.layer_out = outs
        self.meta.update(scale=scale, n_layers=len(layers), legacy=legacy)
        self._fuse_stem()
        self._fuse_chains()

    def _fuse_chains(self):
        """Peephole: two consecutive 1x1 Conv launches A, B of which B reads (part of) A's output at the same pixel become one OP_CHAIN
        launch (csrc/chain1x1.hip) -- the same bits, one launch, A's output (or the part B needs) handed over in LDS.
        HEAD form: B's only source is a channel slice of A's output (C3k2.cv1 -> C3k.cv1|cv2; C2PSA.cv1 -> qkv): A's output is still
        written.  TAIL form: A's output is the LAST channels of B's source view (C3k.cv3 -> C3k2.cv2; ffn[1] -> C2PSA.cv2): it is written
        only if something else reads it."""
        min_px = int(os.environ.get("BSY_CHAIN_MIN_PIXELS", CHAIN_MIN_PIXELS))  # (tests lower it to reach the kernel with small inputs)
        if not self.fuse_chain or self.f32_mode or self.latency:
            return

        def plain(o):
            if o["kind"] != L.OP_CONV or o["ksize"] != 1 or o["stride"] != 1 or o.get("out_f32", 0) or o.get("dst_scale", 1) != 1 or o.get("ksplit", 0):
                return False
            if o.get("lane", 0) != 0 or o["dst"].f32 or o["dst"].up or o["dst"].cmap is not None or o.get("cout", o["dst"].C) != o["dst"].C:
                return False
            w = self.wrecs[o["wkey"]]
            if w.rows is not None or w.cols is not None or w.real_cout or w.real_cin:
                return False
            return all(t is None or (not t.up and not t.f32 and t.cmap is None) for t in (o["src0"], o.get("src1"), o.get("res")))

        def overlaps(t, u):
            return t is not None and t.buf == u.buf and t.coff < u.coff + u.C and u.coff < t.coff + t.C

        i = 0
        while i + 1 < len(self.ops):
            a, b = self.ops[i], self.ops[i + 1]
            i += 1
            if not (plain(a) and plain(b)) or b.get("src1") is not None or self.B * a["OH"] * a["OW"] < min_px:
                continue
            ad, bs = a["dst"], b["src0"]
            if bs.buf != ad.buf or bs.ld != ad.ld:
                continue
            ca0, ca1 = a["src0"].C, a["src1"].C if a.get("src1") is not None else 0
            if a.get("res") is None and ad.coff <= bs.coff and bs.coff + bs.C <= ad.coff + ad.C:      # HEAD
                keep0, lc, d1, h2 = bs.coff - ad.coff, bs.C, ad, None
            elif ad.coff > bs.coff and ad.coff + ad.C == bs.coff + bs.C:                              # TAIL
                keep0, lc, h2 = 0, ad.C, bs.slice(0, bs.C - ad.C)
                readers = any(overlaps(o.get(k), ad) for o in self.ops[i + 1:] for k in self._VIEW_KEYS if k != "dst") or \
                    any(overlaps(t, ad) for o in self.ops[i + 1:] for k in self._LIST_KEYS for t in (o.get(k) or []))
                outs = [t for lo in self.layer_out if lo is not None for t in (lo if isinstance(lo, list) else [lo])]
                d1 = ad if readers or any(overlaps(t, ad) for t in outs) else None
            else:
                continue
            if not chain_supported(ca0, ca1, ad.C, keep0, lc, h2.C if h2 is not None else 0, b["dst"].C):
                continue
            self.ops[i - 1:i + 1] = [dict(kind=L.OP_CHAIN, H=a["H"], W=a["W"], OH=a["OH"], OW=a["OW"], src0=a["src0"], src1=a.get("src1"),
                                          dst=b["dst"], res=b.get("res"), box=[d1, h2, a.get("res")], ksize=1, stride=1, pad=0, act=b["act"],
                                          nl=a["act"], heads=ad.C, key_dim=keep0, mid_c=lc, wkey=a["wkey"], wkey2=b["wkey"],
                                          name=a["name"] + "->" + b["name"].split(".", 2)[-1], lane=0,
                                          mfma_flops=a["mfma_flops"] + b["mfma_flops"], chain=(a["name"], b["name"]))]

    def _fuse_stem(self):
        """Peephole: layer 0 (image conv) + layer 1 (3x3 s2 conv fed by layer 0 alone) -> one OP_STEM launch
        (csrc/stem_fused.hip); layer 0's map is then never materialised (layer_out[0] = None)."""
        if not self.fuse_stem or len(self.ops) < 2:
            return
        a, b = self.ops[0], self.ops[1]
        if a["kind"] != L.OP_CONV_FIRST or b["kind"] != L.OP_CONV:
            return
        mid = a["dst"]
        if (b["src0"] is not mid or b.get("src1") is not None or b.get("res") is not None or b["ksize"] != 3
                or b["stride"] != 2 or b["out_f32"] or not b["act"] or a["ksize"] != 3 or a["stride"] != 2):
            return
        if not stem_supported(mid.C, b["dst"].C, self.H, self.W):
            return
        for o in self.ops[2:]:  # layer 0 must have no other reader
            for key in ("src0", "src1", "res"):
                t = o.get(key)
                if t is not None and t.buf == mid.buf:
                    return
        self.buf_bytes[mid.buf] = 16  # never written
        self.ops[0:2] = [dict(kind=L.OP_STEM, H=self.H, W=self.W, OH=b["OH"], OW=b["OW"], src0=a["src0"], dst=b["dst"],
                              ksize=3, stride=2, pad=1, act=1, wkey=a["wkey"], wkey2=b["wkey"], mid_c=mid.C,
                              in_dtype=self.in_dtype, name=a["name"] + "+" + b["name"].split(".")[-1], lane=0,
                              mfma_flops=a["mfma_flops"] + b["mfma_flops"])]
        self.layer_out[0] = None

    # ---- workspace layout -----------------------------------------------------------------------------------------
    _VIEW_KEYS = ("src0", "src1", "dst", "res")
    _LIST_KEYS = ("box", "cls", "msk")

    def op_buffers(self, o: dict) -> List[int]:
        """Indices of the workspace buffers an op reads or writes (external slots excluded)."""
        out = []
        for k in self._VIEW_KEYS:
            t = o.get(k)
            if t is not None and t.buf < L.BSY_EXT_BASE:
                out.append(t.buf)
        for k in self._LIST_KEYS:
            for t in o.get(k, []) or []:
                if t is not None and t.buf < L.BSY_EXT_BASE:
                    out.append(t.buf)
        return out

    def buffer_lifetimes(self) -> List[Optional[Tuple[int, int]]]:
        """Per buffer: (first, last) index of the ops that touch it, or None for a buffer no op uses.  Ops on side lanes run
        concurrently with everything between their fork and the next join (csrc/engine.hip bsy_plan_run), so every buffer a
        lane op touches is kept alive over that whole region: the serial op order says nothing about when a lane runs."""
        n = len(self.ops)
        life: List[Optional[List[int]]] = [None] * len(self.buf_bytes)
        for i, o in enumerate(self.ops):
            for b in self.op_buffers(o):
                if life[b] is None:
                    life[b] = [i, i]
                else:
                    life[b][1] = i
        i = 0
        while i < n:
            if self.ops[i].get("lane", 0) > 0:
                start = i
                end = n - 1  # the tail join of bsy_plan_run
                for j in range(i + 1, n):
                    if self.ops[j].get("join", 0):
                        end = j
                        break
                for j in range(start, end + 1):
                    if self.ops[j].get("lane", 0) > 0:
                        for b in self.op_buffers(self.ops[j]):
                            life[b][0] = min(life[b][0], start)
                            life[b][1] = max(life[b][1], end)
                i = end + 1
            else:
                i += 1
        return [tuple(x) if x is not None else None for x in life]

    def assign_offsets(self, reuse: bool = True) -> Tuple[List[int], int]:
        """Byte offset of every buffer inside one arena and the arena size.  reuse: buffers whose lifetimes are disjoint may
        share addresses (greedy by size: each buffer, largest first, takes the lowest offset free of every already placed
        buffer it is live together with) -- YOLO11s at batch 64, 640 x 640: 3.8 GB of buffers in a 1.3 GB arena."""
        sizes = [(b + 255) & ~255 for b in self.buf_bytes]
        offs = [0] * len(sizes)
        if not reuse:
            off = 0
            for i, sz in enumerate(sizes):
                offs[i] = off
                off += sz
            return offs, off
        life = self.buffer_lifetimes()
        placed: List[int] = []
        top = 0
        for i in sorted(range(len(sizes)), key=lambda k: (-sizes[k], k)):
            if life[i] is None or sizes[i] == 0:
                continue
            busy = sorted((offs[j], offs[j] + sizes[j]) for j in placed
                          if not (life[j][1] < life[i][0] or life[i][1] < life[j][0]))
            at = 0
            for lo, hi in busy:
                if at + sizes[i] <= lo:
                    break
                at = max(at, hi)
            offs[i] = at
            placed.append(i)
            top = max(top, at + sizes[i])
        return offs, top

    def conv_signature(self, o: dict) -> Optional[tuple]:
        """What decides which kernel configuration is fastest for a conv op (and which are valid): the key of the engine's
        autotune cache.  None for ops the autotuner does not touch."""
        if o["kind"] != L.OP_CONV or self.f32_mode:
            return None
        s0, s1, d, r = o["src0"], o.get("src1"), o["dst"], o.get("res")
        return (self.B, o["H"], o["W"], s0.C, s0.ld, int(s0.up), s1.C if s1 else 0, s1.ld if s1 else 0, int(s1.up) if s1 else 0,
                o.get("cout", d.C), d.ld, o["ksize"], o["stride"], int(r is not None), r.ld if r is not None else 0,
                o.get("out_f32", 0), o.get("dst_scale", 1), o.get("act", 0), o.get("ksplit", 0))

    # ---- serialisation ------------------------------------------------------------------------------------------
    def c_ops(self):
        arr = (L.Op * len(self.ops))()
        for o, d in zip(arr, self.ops):
            def v(t):
                return L.View(*(t.view() if t is not None else L.NO_VIEW))
            o.kind = d["kind"]
            o.B, o.H, o.W, o.OH, o.OW = self.B, d["H"], d["W"], d["OH"], d["OW"]
            o.src0, o.src1 = v(d.get("src0")), v(d.get("src1"))
            o.up0 = int(bool(d.get("src0") and d["src0"].up))
            o.up1 = int(bool(d.get("src1") and d["src1"].up))
            o.dst, o.res = v(d.get("dst")), v(d.get("res"))
            o.ksize, o.stride, o.pad = d.get("ksize", 0), d.get("stride", 0), d.get("pad", 0)
            o.act, o.out_f32 = d.get("act", 0), d.get("out_f32", 0)
            o.dst_scale, o.dst_dy, o.dst_dx = d.get("dst_scale", 1), d.get("dst_dy", 0), d.get("dst_dx", 0)
            if "wkey" in d:
                w = self.wrecs[d["wkey"]]
                assert w.w_off >= 0, "pack weights before serialising"
                o.w_off, o.b_off = w.w_off, w.b_off
            if "wkey2" in d:
                w = self.wrecs[d["wkey2"]]
                assert w.w_off >= 0, "pack weights before serialising"
                o.w2_off, o.b2_off = w.w_off, w.b_off
            o.mid_c = d.get("mid_c", 0)
            for j, k in enumerate(d.get("wkeys", [])):
                w = self.wrecs[k]
                assert w.w_off >= 0, "pack weights before serialising"
                o.aux_off[2 * j], o.aux_off[2 * j + 1] = w.w_off, w.b_off
            if d["kind"] == L.OP_ELA:
                cf = self.wrecs[d["wkey"]].coef
                d = dict(d, scale=cf[0], lvl_stride=[cf[1], cf[2]])
            o.heads, o.key_dim, o.head_dim, o.scale = d.get("heads", 0), d.get("key_dim", 0), d.get("head_dim", 0), \
                d.get("scale", 0.0)
            o.nl, o.nc, o.nm, o.A = d.get("nl", 0), d.get("nc", 0), d.get("nm", 0), d.get("A", 0)
            for j in range(3):
                o.box[j] = v(d["box"][j] if j < len(d.get("box", [])) else None)
                o.cls[j] = v(d["cls"][j] if j < len(d.get("cls", [])) else None)
                o.msk[j] = v(d["msk"][j] if j < len(d.get("msk", [])) else None)
                o.lvl_h[j] = d["lvl_h"][j] if j < len(d.get("lvl_h", [])) else 0
                o.lvl_w[j] = d["lvl_w"][j] if j < len(d.get("lvl_w", [])) else 0
                o.lvl_stride[j] = d["lvl_stride"][j] if j < len(d.get("lvl_stride", [])) else 0.0
            o.in_dtype, o.out_dtype, o.level = d.get("in_dtype", 0), d.get("out_dtype", 0), d.get("level", 0)
            o.lane, o.join, o.tuned_cfg = d.get("lane", 0), d.get("join", 0), 0
            o.prec = (2 if self.split_f16 else 1) if self.f32_mode else 0
            o.ksplit = d.get("ksplit", 0)
        return arr
