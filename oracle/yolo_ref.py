"""CPU restatement (torch fp32, ATen CPU kernels) of the reference's YOLO forward path.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Every function cites the reference
file:line it follows (paths relative to /root/reference/ultralytics).

The graph is described by our own layer tables (GRAPHS below); widths/repeats are scaled exactly as
nn/tasks.py:parse_model does (:940-1105).  Parameters are addressed by the reference's own
state_dict names ("model.<i>.<...>") so a reference state_dict drops in unchanged.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3  # utils/torch_utils.py:424 (initialize_weights sets BatchNorm2d.eps = 1e-3)

# Test aid (NOT reference behaviour): when True, the restatement rounds weights and every stored activation to fp16 at
# the points where the HIP engine stores fp16 (fp32 accumulation in between), i.e. it models the engine's numerics.
# Comparing the engine with THIS isolates kernel errors from the unavoidable fp16-storage error of the whole network.
FP16_EMULATION = False


def _q(t):
    return t.half().float() if FP16_EMULATION else t


# --------------------------------------------------------------------------------------------
# Graph tables.  (from, repeats, type, args) -- same information as cfg/models/11/yolo11-seg.yaml:15-47
# (stock YOLO11 backbone+neck) and cfg/models/v8/yolov8-seg.yaml:15-46, written in our own notation.
# HEAD is appended by build_model (Detect or Segment).
# --------------------------------------------------------------------------------------------
SCALES = {
    "yolo11": {"n": (0.50, 0.25, 1024), "s": (0.50, 0.50, 1024), "m": (0.50, 1.00, 512),
               "l": (1.00, 1.00, 512), "x": (1.00, 1.50, 512)},
    "yolov8": {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768),
               "l": (1.00, 1.00, 512), "x": (1.00, 1.25, 512)},
}
GRAPHS = {
    "yolo11": [
        (-1, 1, "Conv", (64, 3, 2)), (-1, 1, "Conv", (128, 3, 2)), (-1, 2, "C3k2", (256, False, 0.25)),
        (-1, 1, "Conv", (256, 3, 2)), (-1, 2, "C3k2", (512, False, 0.25)), (-1, 1, "Conv", (512, 3, 2)),
        (-1, 2, "C3k2", (512, True)), (-1, 1, "Conv", (1024, 3, 2)), (-1, 2, "C3k2", (1024, True)),
        (-1, 1, "SPPF", (1024, 5)), (-1, 2, "C2PSA", (1024,)),
        (-1, 1, "Upsample", ()), ((-1, 6), 1, "Concat", ()), (-1, 2, "C3k2", (512, False)),
        (-1, 1, "Upsample", ()), ((-1, 4), 1, "Concat", ()), (-1, 2, "C3k2", (256, False)),
        (-1, 1, "Conv", (256, 3, 2)), ((-1, 13), 1, "Concat", ()), (-1, 2, "C3k2", (512, False)),
        (-1, 1, "Conv", (512, 3, 2)), ((-1, 10), 1, "Concat", ()), (-1, 2, "C3k2", (1024, True)),
    ],
    "yolov8": [
        (-1, 1, "Conv", (64, 3, 2)), (-1, 1, "Conv", (128, 3, 2)), (-1, 3, "C2f", (128, True)),
        (-1, 1, "Conv", (256, 3, 2)), (-1, 6, "C2f", (256, True)), (-1, 1, "Conv", (512, 3, 2)),
        (-1, 6, "C2f", (512, True)), (-1, 1, "Conv", (1024, 3, 2)), (-1, 3, "C2f", (1024, True)),
        (-1, 1, "SPPF", (1024, 5)),
        (-1, 1, "Upsample", ()), ((-1, 6), 1, "Concat", ()), (-1, 3, "C2f", (512,)),
        (-1, 1, "Upsample", ()), ((-1, 4), 1, "Concat", ()), (-1, 3, "C2f", (256,)),
        (-1, 1, "Conv", (256, 3, 2)), ((-1, 12), 1, "Concat", ()), (-1, 3, "C2f", (512,)),
        (-1, 1, "Conv", (512, 3, 2)), ((-1, 9), 1, "Concat", ()), (-1, 3, "C2f", (1024,)),
    ],
}
# BS-YOLO: cfg/models/11/yolo11.yaml:15-52 of the fork (the repo's namesake graph): C3k2_gai / SCDown / MSCAAttention in the
# backbone, an ELA gate after every neck C3k2; Detect reads the ELA outputs.  Layer 21 concatenates layer 13 -- itself a
# Concat (the yaml kept the stock index after inserting layers) -- so that Concat has three tensors behind it.
GRAPHS["bsyolo11"] = [
    (-1, 1, "Conv", (64, 3, 2)), (-1, 1, "Conv", (128, 3, 2)), (-1, 2, "C3k2_gai", (256, False, 0.25)),
    (-1, 1, "Conv", (256, 3, 2)), (-1, 2, "C3k2_gai", (512, False, 0.25)), (-1, 1, "SCDown", (512, 3, 2)),
    (-1, 2, "C3k2_gai", (512, True)), (-1, 1, "SCDown", (1024, 3, 2)), (-1, 2, "C3k2_gai", (1024, True)),
    (-1, 1, "SPPF", (1024, 5)), (-1, 2, "C2PSA", (1024,)), (-1, 1, "MSCAAttention", ()),
    (-1, 1, "Upsample", ()), ((-1, 6), 1, "Concat", ()), (-1, 2, "C3k2", (512, False)), (-1, 1, "ELA", (512,)),
    (-1, 1, "Upsample", ()), ((-1, 4), 1, "Concat", ()), (-1, 2, "C3k2", (256, False)), (-1, 1, "ELA", (256,)),
    (-1, 1, "Conv", (256, 3, 2)), ((-1, 13), 1, "Concat", ()), (-1, 2, "C3k2", (512, False)), (-1, 1, "ELA", (512,)),
    (-1, 1, "SCDown", (512, 3, 2)), ((-1, 10), 1, "Concat", ()), (-1, 2, "C3k2", (1024, True)), (-1, 1, "ELA", (1024,)),
]
SCALES["bsyolo11"] = SCALES["yolo11"]
# YOLOv5u: cfg/models/v5/yolov5.yaml:14-50 (6x6 stride-2 pad-2 stem, C3 blocks, SPPF, anchor-free Detect head)
GRAPHS["yolov5"] = [
    (-1, 1, "Conv", (64, 6, 2, 2)), (-1, 1, "Conv", (128, 3, 2)), (-1, 3, "C3", (128,)),
    (-1, 1, "Conv", (256, 3, 2)), (-1, 6, "C3", (256,)), (-1, 1, "Conv", (512, 3, 2)),
    (-1, 9, "C3", (512,)), (-1, 1, "Conv", (1024, 3, 2)), (-1, 3, "C3", (1024,)),
    (-1, 1, "SPPF", (1024, 5)),
    (-1, 1, "Conv", (512, 1, 1)), (-1, 1, "Upsample", ()), ((-1, 6), 1, "Concat", ()), (-1, 3, "C3", (512, False)),
    (-1, 1, "Conv", (256, 1, 1)), (-1, 1, "Upsample", ()), ((-1, 4), 1, "Concat", ()), (-1, 3, "C3", (256, False)),
    (-1, 1, "Conv", (256, 3, 2)), ((-1, 14), 1, "Concat", ()), (-1, 3, "C3", (512, False)),
    (-1, 1, "Conv", (512, 3, 2)), ((-1, 10), 1, "Concat", ()), (-1, 3, "C3", (1024, False)),
]
SCALES["yolov5"] = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 1024), "l": (1.00, 1.00, 1024),
                    "x": (1.33, 1.25, 1024)}
HEAD_FROM = {"yolo11": (16, 19, 22), "yolov8": (15, 18, 21), "bsyolo11": (19, 23, 27), "yolov5": (17, 20, 23)}


def make_divisible(x, d):  # utils/ops.py:130-143
    return math.ceil(x / d) * d


def autopad(k, p=None):  # nn/modules/conv.py:29-35 (d == 1 on this path)
    return k // 2 if p is None else p


# --------------------------------------------------------------------------------------------
# Module restatements.  Each holds its parameter-name prefix; forward(P, x) pulls tensors from P.
# --------------------------------------------------------------------------------------------
class Conv:
    """conv -> BN -> SiLU; after fold: SiLU(conv(x) + b).  nn/modules/conv.py:133-151."""

    def __init__(self, name, c1, c2, k=1, s=1, p=None, g=1, act=True):
        self.name, self.c1, self.c2, self.k, self.s, self.g, self.act = name, c1, c2, k, s, g, act
        self.p = autopad(k, p)

    def specs(self):
        n = self.name
        yield (n + ".conv.weight", (self.c2, self.c1 // self.g, self.k, self.k))
        for suf in ("weight", "bias", "running_mean", "running_var"):
            yield (n + ".bn." + suf, (self.c2,))

    def folded(self, P):
        """utils/torch_utils.py:242-269 fuse_conv_and_bn (conv has no bias here)."""
        n = self.name
        if n + ".conv.bias" in P and n + ".bn.weight" not in P:  # already-fused state_dict
            return P[n + ".conv.weight"], P[n + ".conv.bias"]
        w = P[n + ".conv.weight"]
        bw, bb, mu, var = (P[n + ".bn." + s] for s in ("weight", "bias", "running_mean", "running_var"))
        scale = bw.div(torch.sqrt(BN_EPS + var))
        wf = torch.mm(torch.diag(scale), w.view(self.c2, -1)).view(w.shape)
        bf = bb - bw.mul(mu).div(torch.sqrt(var + BN_EPS))
        return wf, bf

    def __call__(self, P, x, res=None):
        w, b = self.folded(P)
        if self.g == 1:
            w = _q(w)  # dense conv weights are packed as fp16 (depthwise weights stay fp32 in the engine)
        y = F.conv2d(_q(x), w, b, self.s, self.p, 1, self.g)
        y = F.silu(y) if self.act else y
        if res is not None:
            # dense convs: the engine's epilogue parks fp16(act(.)) in LDS, then adds the residual and rounds again;
            # depthwise convs add the residual in fp32 before the only rounding
            y = (_q(y) if self.g == 1 else y) + res
        return _q(y)


def DWConv(name, c1, c2, k=1, s=1, act=True):  # nn/modules/conv.py:224-229
    return Conv(name, c1, c2, k, s, g=math.gcd(c1, c2), act=act)


class PlainConv:
    """bare nn.Conv2d with bias (Detect/Segment output convs, head.py:41-43)."""

    def __init__(self, name, c1, c2, k=1):
        self.name, self.c1, self.c2, self.k = name, c1, c2, k

    def specs(self):
        yield (self.name + ".weight", (self.c2, self.c1, self.k, self.k))
        yield (self.name + ".bias", (self.c2,))

    def __call__(self, P, x):
        return F.conv2d(_q(x), _q(P[self.name + ".weight"]), P[self.name + ".bias"])  # fp32 output in the engine too


class Seq:
    def __init__(self, mods):
        self.mods = list(mods)

    def specs(self):
        for m in self.mods:
            yield from m.specs()

    def __call__(self, P, x):
        for m in self.mods:
            x = m(P, x)
        return x


class Bottleneck:
    """nn/modules/block.py:3405-3419."""

    def __init__(self, name, c1, c2, shortcut=True, g=1, k=(3, 3), e=0.5):
        c_ = int(c2 * e)
        self.cv1 = Conv(name + ".cv1", c1, c_, k[0], 1)
        self.cv2 = Conv(name + ".cv2", c_, c2, k[1], 1, g=g)
        self.add = shortcut and c1 == c2

    def specs(self):
        yield from self.cv1.specs()
        yield from self.cv2.specs()

    def __call__(self, P, x):
        if FP16_EMULATION:  # the engine adds the shortcut in the conv epilogue, before the fp16 store
            return self.cv2(P, self.cv1(P, x), res=x if self.add else None)
        y = self.cv2(P, self.cv1(P, x))
        return x + y if self.add else y


class C3k:
    """C3 with k=3 Bottlenecks, e=1.0.  block.py:3320-3334 (C3) + :3807-3815 (C3k)."""

    def __init__(self, name, c1, c2, n=1, shortcut=True, g=1, e=0.5, k=3):
        c_ = int(c2 * e)
        self.cv1 = Conv(name + ".cv1", c1, c_, 1, 1)
        self.cv2 = Conv(name + ".cv2", c1, c_, 1, 1)
        self.cv3 = Conv(name + ".cv3", 2 * c_, c2, 1)
        self.m = Seq(Bottleneck(f"{name}.m.{i}", c_, c_, shortcut, g, k=(k, k), e=1.0) for i in range(n))

    def specs(self):
        for m in (self.cv1, self.cv2, self.cv3, self.m):
            yield from m.specs()

    def __call__(self, P, x):
        return self.cv3(P, torch.cat((self.m(P, self.cv1(P, x)), self.cv2(P, x)), 1))


class C3(C3k):
    """block.py:3320-3334: cv3(cat(m(cv1 x), cv2 x)), m = n x Bottleneck(c_, c_, shortcut, g, k=((1,1),(3,3)), e=1.0)."""

    def __init__(self, name, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__(name, c1, c2, n, shortcut, g, e)
        c_ = int(c2 * e)
        self.m = Seq(Bottleneck(f"{name}.m.{i}", c_, c_, shortcut, g, k=(1, 3), e=1.0) for i in range(n))


class C2f:
    """block.py:3295-3317.  C3k2 (block.py:3796-3804 == :4148-4156) swaps the inner module list."""

    def __init__(self, name, c1, c2, n=1, shortcut=False, g=1, e=0.5, c3k=None):
        self.c = int(c2 * e)
        self.cv1 = Conv(name + ".cv1", c1, 2 * self.c, 1, 1)
        self.cv2 = Conv(name + ".cv2", (2 + n) * self.c, c2, 1)
        if c3k is None:  # plain C2f
            self.m = [Bottleneck(f"{name}.m.{i}", self.c, self.c, shortcut, g, k=(3, 3), e=1.0) for i in range(n)]
        elif c3k:
            self.m = [C3k(f"{name}.m.{i}", self.c, self.c, 2, shortcut, g) for i in range(n)]
        else:
            self.m = [Bottleneck(f"{name}.m.{i}", self.c, self.c, shortcut, g) for i in range(n)]

    def specs(self):
        yield from self.cv1.specs()
        yield from self.cv2.specs()
        for m in self.m:
            yield from m.specs()

    def __call__(self, P, x):
        y = list(self.cv1(P, x).chunk(2, 1))
        y.extend(m(P, y[-1]) for m in self.m)
        return self.cv2(P, torch.cat(y, 1))


def C3k2(name, c1, c2, n=1, c3k=False, e=0.5, g=1, shortcut=True):
    return C2f(name, c1, c2, n, shortcut, g, e, c3k=bool(c3k))


class SPPF:
    """block.py:3114-3149."""

    def __init__(self, name, c1, c2, k=5):
        c_ = c1 // 2
        self.cv1 = Conv(name + ".cv1", c1, c_, 1, 1)
        self.cv2 = Conv(name + ".cv2", c_ * 4, c2, 1, 1)
        self.k = k

    def specs(self):
        yield from self.cv1.specs()
        yield from self.cv2.specs()

    def __call__(self, P, x):
        y = [self.cv1(P, x)]
        y.extend(F.max_pool2d(y[-1], self.k, 1, self.k // 2) for _ in range(3))
        return self.cv2(P, torch.cat(y, 1))


class Attention:
    """block.py:4235-4288."""

    def __init__(self, name, dim, num_heads=8, attn_ratio=0.5):
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.key_dim = int(self.head_dim * attn_ratio)
        self.scale = self.key_dim ** -0.5
        h = dim + self.key_dim * num_heads * 2
        self.qkv = Conv(name + ".qkv", dim, h, 1, act=False)
        self.proj = Conv(name + ".proj", dim, dim, 1, act=False)
        self.pe = Conv(name + ".pe", dim, dim, 3, 1, g=dim, act=False)

    def specs(self):
        for m in (self.qkv, self.proj, self.pe):
            yield from m.specs()

    def __call__(self, P, x, res=None):
        B, C, H, W = x.shape
        N = H * W
        qkv = self.qkv(P, x)
        q, k, v = qkv.view(B, self.num_heads, self.key_dim * 2 + self.head_dim, N).split(
            [self.key_dim, self.key_dim, self.head_dim], dim=2)
        attn = (q.transpose(-2, -1) @ k) * self.scale
        attn = attn.softmax(dim=-1)
        if FP16_EMULATION:  # attention output stored as fp16, then pe(v) + it stored as fp16
            a = _q((v @ attn.transpose(-2, -1)).view(B, C, H, W))
            return self.proj(P, self.pe(P, v.reshape(B, C, H, W), res=a), res=res)
        x = (v @ attn.transpose(-2, -1)).view(B, C, H, W) + self.pe(P, v.reshape(B, C, H, W))
        return self.proj(P, x)


class PSABlock:
    """block.py:4348-4383."""

    def __init__(self, name, c, attn_ratio=0.5, num_heads=4, shortcut=True):
        self.attn = Attention(name + ".attn", c, num_heads=num_heads, attn_ratio=attn_ratio)
        self.ffn = Seq([Conv(name + ".ffn.0", c, c * 2, 1), Conv(name + ".ffn.1", c * 2, c, 1, act=False)])
        self.add = shortcut

    def specs(self):
        yield from self.attn.specs()
        yield from self.ffn.specs()

    def __call__(self, P, x):
        if FP16_EMULATION and self.add:  # shortcut adds fused into the proj / ffn.1 epilogues
            x = self.attn(P, x, res=x)
            return self.ffn.mods[1](P, self.ffn.mods[0](P, x), res=x)
        x = x + self.attn(P, x) if self.add else self.attn(P, x)
        x = x + self.ffn(P, x) if self.add else self.ffn(P, x)
        return x


class C2PSA:
    """block.py:4429-4468."""

    def __init__(self, name, c1, c2, n=1, e=0.5):
        assert c1 == c2
        self.c = int(c1 * e)
        self.cv1 = Conv(name + ".cv1", c1, 2 * self.c, 1, 1)
        self.cv2 = Conv(name + ".cv2", 2 * self.c, c1, 1)
        self.m = Seq(PSABlock(f"{name}.m.{i}", self.c, attn_ratio=0.5, num_heads=self.c // 64) for i in range(n))

    def specs(self):
        for m in (self.cv1, self.cv2, self.m):
            yield from m.specs()

    def __call__(self, P, x):
        a, b = self.cv1(P, x).split((self.c, self.c), dim=1)
        b = self.m(P, b)
        return self.cv2(P, torch.cat((a, b), 1))


class PMSFA:
    """block.py:3035-3054: 3x3 conv -> half of it through a depthwise 5x5 -> half of that through a depthwise 7x7 ->
    cat([7x7 out, other 5x5 half, other 3x3 half]) -> 1x1 conv, + x."""

    def __init__(self, name, inc):
        self.inc = inc
        self.conv1 = Conv(name + ".conv1", inc, inc, 3)
        self.conv2 = Conv(name + ".conv2", inc // 2, inc // 2, 5, g=inc // 2)
        self.conv3 = Conv(name + ".conv3", inc // 4, inc // 4, 7, g=inc // 4)
        self.conv4 = Conv(name + ".conv4", inc, inc, 1)

    def specs(self):
        for m in (self.conv1, self.conv2, self.conv3, self.conv4):
            yield from m.specs()

    def __call__(self, P, x):
        a1, a2 = self.conv1(P, x).chunk(2, 1)
        b1, b2 = self.conv2(P, a1).chunk(2, 1)
        c = self.conv3(P, b1)
        cat = torch.cat([c, b2, a2], 1)
        if FP16_EMULATION:
            return self.conv4(P, cat, res=x)
        return self.conv4(P, cat) + x


class C3k_gai:
    """block.py:3079-3086: C3 (block.py:3320-3334) whose inner chain is n x PMSFA(c_)."""

    def __init__(self, name, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        c_ = int(c2 * e)
        self.cv1 = Conv(name + ".cv1", c1, c_, 1, 1)
        self.cv2 = Conv(name + ".cv2", c1, c_, 1, 1)
        self.cv3 = Conv(name + ".cv3", 2 * c_, c2, 1)
        self.m = Seq(PMSFA(f"{name}.m.{i}", c_) for i in range(n))

    def specs(self):
        for m in (self.cv1, self.cv2, self.cv3, self.m):
            yield from m.specs()

    def __call__(self, P, x):
        return self.cv3(P, torch.cat((self.m(P, self.cv1(P, x)), self.cv2(P, x)), 1))


class C3k2_gai:
    """block.py:3087-3095: C2f whose inner modules are PMSFA(c) (c3k False) or C3k_gai(c, c, 2) (c3k True)."""

    def __init__(self, name, c1, c2, n=1, c3k=False, e=0.5, g=1, shortcut=True):
        self.c = int(c2 * e)
        self.cv1 = Conv(name + ".cv1", c1, 2 * self.c, 1, 1)
        self.cv2 = Conv(name + ".cv2", (2 + n) * self.c, c2, 1)
        self.m = [C3k_gai(f"{name}.m.{i}", self.c, self.c, 2, shortcut, g) if c3k else PMSFA(f"{name}.m.{i}", self.c)
                  for i in range(n)]

    def specs(self):
        yield from self.cv1.specs()
        yield from self.cv2.specs()
        for m in self.m:
            yield from m.specs()

    def __call__(self, P, x):
        y = list(self.cv1(P, x).chunk(2, 1))
        y.extend(m(P, y[-1]) for m in self.m)
        return self.cv2(P, torch.cat(y, 1))


class SCDown:
    """block.py:4503-4535: pointwise Conv (BN + SiLU) then depthwise k x k stride-s Conv (BN, no activation)."""

    def __init__(self, name, c1, c2, k, s):
        self.cv1 = Conv(name + ".cv1", c1, c2, 1, 1)
        self.cv2 = Conv(name + ".cv2", c2, c2, k, s, g=c2, act=False)

    def specs(self):
        yield from self.cv1.specs()
        yield from self.cv2.specs()

    def __call__(self, P, x):
        return self.cv2(P, self.cv1(P, x))


class MSCAAttention:
    """nn/Addmodules/MSCA.py:19-88 (SegNeXt MSCA with learned branch weights): depthwise 5x5, four strip-conv branches
    (1x5/5x1, 1x7/7x1, 1x11/11x1, 1x21/21x1; the first three followed by the SAME depthwise 1x1 `dilconv`), branch weights
    = softmax over the 4 branches of sigmoid(SE_i(GAP(branch_i))), 1x1 conv of the weighted sum, times the input.
    All convs are bare nn.Conv2d with bias (no BN, no activation)."""

    STRIPS = ((5, "conv0"), (7, "conv1"), (11, "conv2"), (21, "conv3"))

    def __init__(self, name, dim):
        self.name, self.dim = name, dim

    def specs(self):
        n, d = self.name, self.dim
        yield (n + ".conv0.weight", (d, 1, 5, 5))
        yield (n + ".conv0.bias", (d,))
        for k, base in self.STRIPS:
            yield (f"{n}.{base}_1.weight", (d, 1, 1, k))
            yield (f"{n}.{base}_1.bias", (d,))
            yield (f"{n}.{base}_2.weight", (d, 1, k, 1))
            yield (f"{n}.{base}_2.bias", (d,))
            if base == "conv0":  # registration order of MSCA.py:26-39
                yield (n + ".dilconv.weight", (d, 1, 1, 1))
                yield (n + ".dilconv.bias", (d,))
        yield (n + ".conv4.weight", (d, d, 1, 1))
        yield (n + ".conv4.bias", (d,))
        for i in range(1, 5):
            yield (f"{n}.SE{i}.conv.0.weight", (d, d, 1, 1))
            yield (f"{n}.SE{i}.conv.0.bias", (d,))

    def __call__(self, P, x):
        n, d = self.name, self.dim
        x = _q(x)
        attn = _q(F.conv2d(x, P[n + ".conv0.weight"], P[n + ".conv0.bias"], 1, 2, 1, d))
        branches = []
        for i, (k, base) in enumerate(self.STRIPS):
            a = _q(F.conv2d(attn, P[f"{n}.{base}_1.weight"], P[f"{n}.{base}_1.bias"], 1, (0, k // 2), 1, d))
            w2, b2 = P[f"{n}.{base}_2.weight"], P[f"{n}.{base}_2.bias"]
            if i < 3 and FP16_EMULATION:  # the engine folds the depthwise 1x1 into the column conv (one rounding)
                dw, db = P[n + ".dilconv.weight"].view(d), P[n + ".dilconv.bias"]
                a = F.conv2d(a, w2 * dw.view(d, 1, 1, 1), b2 * dw + db, 1, (k // 2, 0), 1, d)
            else:
                a = F.conv2d(a, w2, b2, 1, (k // 2, 0), 1, d)
                if i < 3:
                    a = F.conv2d(a, P[n + ".dilconv.weight"], P[n + ".dilconv.bias"], 1, 0, 2, d)  # k = 1: dilation is moot
            branches.append(_q(a))
        ws = [F.conv2d(_q(b.mean((2, 3), keepdim=True)), _q(P[f"{n}.SE{i + 1}.conv.0.weight"]), P[f"{n}.SE{i + 1}.conv.0.bias"])
              for i, b in enumerate(branches)]                               # (B, d, 1, 1) each
        weight = torch.softmax(torch.sigmoid(torch.cat(ws, 2)), 2)           # (B, d, 4, 1): softmax over the branches
        x_att = sum(weight[:, :, i:i + 1] * b for i, b in enumerate(branches))
        out = F.conv2d(_q(x_att), _q(P[n + ".conv4.weight"]), P[n + ".conv4.bias"])
        return _q(_q(out) * x)


class ELA:
    """nn/Addmodules/ELA.py:33-101: x * (sig(ch_w) * channel gate + sig(sp_w) * row gate * column gate) + sig(res_w) * x.
    Channel gate: sigmoid of a depthwise Conv1d applied to the length-1 sequence of the channel's global mean (only the
    centre tap ever meets data).  Row / column gates: mean over W / H -> depthwise Conv1d (dilation 2) -> GroupNorm
    (channel // 16 groups) -> sigmoid; one conv and one norm shared by both directions."""

    def __init__(self, name, channel, b=1, gamma=2):
        self.name, self.c = name, channel
        k = int(abs((math.log(channel, 2) + b) / gamma))
        self.k = k if k % 2 else k + 1
        self.groups = max(1, channel // 16)

    def specs(self):
        n = self.name
        yield (n + ".ch_weight", (1,))
        yield (n + ".sp_weight", (1,))
        yield (n + ".res_weight", (1,))
        yield (n + ".ch_att.2.weight", (self.c, 1, self.k))
        yield (n + ".spatial_conv.weight", (self.c, 1, self.k))
        yield (n + ".gn.weight", (self.c,))
        yield (n + ".gn.bias", (self.c,))

    def __call__(self, P, x):
        n, k = self.name, self.k
        x = _q(x)
        B, C, H, W = x.shape
        ch = torch.sigmoid(F.conv1d(x.mean((2, 3)).view(B, C, 1), P[n + ".ch_att.2.weight"], None, 1, (k - 1) // 2, 1, C))
        ch = ch.view(B, C, 1, 1)

        def gate(v):  # v (B, C, L)
            v = F.conv1d(v, P[n + ".spatial_conv.weight"], None, 1, (k - 1) * 2 // 2, 2, C)
            return torch.sigmoid(F.group_norm(v, self.groups, P[n + ".gn.weight"], P[n + ".gn.bias"], 1e-5))

        h_att = gate(x.mean(3)).view(B, C, H, 1)
        w_att = gate(x.mean(2)).view(B, C, 1, W)
        mask = torch.sigmoid(P[n + ".ch_weight"]) * ch + torch.sigmoid(P[n + ".sp_weight"]) * (h_att * w_att)
        return _q(x * mask + torch.sigmoid(P[n + ".res_weight"]) * x)


def make_anchors(feats, strides, offset=0.5):  # utils/tal.py:371-383
    pts, st = [], []
    for f, s in zip(feats, strides):
        h, w = f.shape[2:]
        sx = torch.arange(w, dtype=f.dtype) + offset
        sy = torch.arange(h, dtype=f.dtype) + offset
        sy, sx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((sx, sy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(s), dtype=f.dtype))
    return torch.cat(pts), torch.cat(st)


def dist2bbox(distance, anchor_points, xywh=True, dim=-1):  # utils/tal.py:386-395
    lt, rb = distance.chunk(2, dim)
    x1y1 = anchor_points - lt
    x2y2 = anchor_points + rb
    if xywh:
        return torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), dim)
    return torch.cat((x1y1, x2y2), dim)


class Detect:
    """head.py:21-148 (inference, non-export, non-end2end path)."""

    def __init__(self, name, nc, ch, legacy, strides):
        self.name, self.nc, self.nl, self.reg_max = name, nc, len(ch), 16
        self.no = nc + 64
        self.stride = strides
        c2, c3 = max(16, ch[0] // 4, 64), max(ch[0], min(nc, 100))
        self.cv2 = [Seq([Conv(f"{name}.cv2.{i}.0", x, c2, 3), Conv(f"{name}.cv2.{i}.1", c2, c2, 3),
                         PlainConv(f"{name}.cv2.{i}.2", c2, 64)]) for i, x in enumerate(ch)]
        if legacy:
            self.cv3 = [Seq([Conv(f"{name}.cv3.{i}.0", x, c3, 3), Conv(f"{name}.cv3.{i}.1", c3, c3, 3),
                             PlainConv(f"{name}.cv3.{i}.2", c3, nc)]) for i, x in enumerate(ch)]
        else:
            self.cv3 = [Seq([DWConv(f"{name}.cv3.{i}.0.0", x, x, 3), Conv(f"{name}.cv3.{i}.0.1", x, c3, 1),
                             DWConv(f"{name}.cv3.{i}.1.0", c3, c3, 3), Conv(f"{name}.cv3.{i}.1.1", c3, c3, 1),
                             PlainConv(f"{name}.cv3.{i}.2", c3, nc)]) for i, x in enumerate(ch)]

    def specs(self):
        for s in self.cv2:
            yield from s.specs()
        for s in self.cv3:
            yield from s.specs()
        yield (self.name + ".dfl.conv.weight", (1, 16, 1, 1))

    def raw(self, P, xs):
        return [torch.cat((self.cv2[i](P, x), self.cv3[i](P, x)), 1) for i, x in enumerate(xs)]

    def decode(self, raw):
        """head.py:100-131 _inference + block.py:58-77 DFL."""
        b = raw[0].shape[0]
        x_cat = torch.cat([xi.view(b, self.no, -1) for xi in raw], 2)
        anchors, strides = (t.transpose(0, 1) for t in make_anchors(raw, self.stride, 0.5))
        box, cls = x_cat.split((64, self.nc), 1)
        a = box.shape[-1]
        proj = torch.arange(16, dtype=box.dtype).view(1, 16, 1, 1)
        dfl = F.conv2d(box.view(b, 4, 16, a).transpose(2, 1).softmax(1), proj).view(b, 4, a)
        dbox = dist2bbox(dfl, anchors.unsqueeze(0), xywh=True, dim=1) * strides
        return torch.cat((dbox, cls.sigmoid()), 1)

    def __call__(self, P, xs):
        raw = self.raw(P, xs)
        return self.decode(raw), raw


class Proto:
    """block.py:80-97."""

    def __init__(self, name, c1, c_=256, c2=32):
        self.name = name
        self.cv1 = Conv(name + ".cv1", c1, c_, 3)
        self.cv2 = Conv(name + ".cv2", c_, c_, 3)
        self.cv3 = Conv(name + ".cv3", c_, c2, 1)
        self.c_ = c_

    def specs(self):
        yield from self.cv1.specs()
        yield (self.name + ".upsample.weight", (self.c_, self.c_, 2, 2))
        yield (self.name + ".upsample.bias", (self.c_,))
        yield from self.cv2.specs()
        yield from self.cv3.specs()

    def __call__(self, P, x):
        x = self.cv1(P, x)
        x = _q(F.conv_transpose2d(x, _q(P[self.name + ".upsample.weight"]), P[self.name + ".upsample.bias"], 2, 0))
        return self.cv3(P, self.cv2(P, x))


class Segment(Detect):
    """head.py:175-197."""

    def __init__(self, name, nc, nm, npr, ch, legacy, strides):
        super().__init__(name, nc, ch, legacy, strides)
        self.nm = nm
        self.proto = Proto(name + ".proto", ch[0], npr, nm)
        c4 = max(ch[0] // 4, nm)
        self.cv4 = [Seq([Conv(f"{name}.cv4.{i}.0", x, c4, 3), Conv(f"{name}.cv4.{i}.1", c4, c4, 3),
                         PlainConv(f"{name}.cv4.{i}.2", c4, nm)]) for i, x in enumerate(ch)]

    def specs(self):
        yield from Detect.specs(self)
        yield from self.proto.specs()
        for s in self.cv4:
            yield from s.specs()

    def __call__(self, P, xs):
        p = self.proto(P, xs[0])
        bs = p.shape[0]
        mc = torch.cat([self.cv4[i](P, x).view(bs, self.nm, -1) for i, x in enumerate(xs)], 2)
        y, raw = Detect.__call__(self, P, xs)
        return torch.cat([y, mc], 1), (raw, mc, p)


# --------------------------------------------------------------------------------------------
# Model: graph walk as nn/tasks.py:138-165 (_predict_once) over layers built as parse_model :940-1105
# --------------------------------------------------------------------------------------------
class Model:
    def __init__(self, family="yolo11", scale="s", nc=80, task="detect", ch=3):
        depth, width, max_ch = SCALES[family][scale]
        self.family, self.scale, self.nc, self.task = family, scale, nc, task
        legacy = True
        chans: List[int] = []
        self.layers = []
        for i, (f, n, t, args) in enumerate(GRAPHS[family]):
            name = f"model.{i}"
            n = max(round(n * depth), 1) if n > 1 else n  # tasks.py:972
            if t in ("Conv", "C3k2", "C2f", "SPPF", "C2PSA", "C3k2_gai", "SCDown", "C3", "DWConv"):
                c1 = chans[f] if chans else ch  # tasks.py:1014 (ch[f]; the first layer sees the image)
                c2 = make_divisible(min(args[0], max_ch) * width, 8)  # tasks.py:1016
                if t == "Conv":
                    m = Conv(name, c1, c2, *args[1:])
                elif t == "C3k2":
                    legacy = False  # tasks.py:1046-1049
                    a = list(args[1:])
                    if scale in "mlx":
                        a[0] = True
                    m = C3k2(name, c1, c2, n, *a)
                elif t == "C3k2_gai":  # tasks.py:1038 (repeat count inserted; no m/l/x override, no legacy switch)
                    m = C3k2_gai(name, c1, c2, n, *args[1:])
                elif t == "SCDown":
                    m = SCDown(name, c1, c2, *args[1:])
                elif t == "C2f":
                    m = C2f(name, c1, c2, n, *args[1:])
                elif t == "C3":
                    m = C3(name, c1, c2, n, *args[1:])
                elif t == "DWConv":
                    m = DWConv(name, c1, c2, *args[1:])
                elif t == "SPPF":
                    m = SPPF(name, c1, c2, *args[1:])
                else:
                    m = C2PSA(name, c1, c2, n)
            elif t == "MSCAAttention":  # tasks.py:1052-1054
                c2 = chans[f]
                m = MSCAAttention(name, c2)
            elif t == "ELA":  # tasks.py:1066-1070: built on the INPUT channels; args[0] only feeds the channel bookkeeping
                c2 = make_divisible(min(args[0], max_ch) * width, 8)
                m = ELA(name, chans[f])
            elif t == "Upsample":
                m, c2 = "up", chans[f]
            elif t == "Concat":
                m, c2 = "cat", sum(chans[x] for x in f)
            else:
                raise ValueError(t)
            self.layers.append((f, m))
            chans.append(c2)
        hf = HEAD_FROM[family]
        hch = [chans[x] for x in hf]
        self.strides = self._probe_strides(hf)
        hname = f"model.{len(self.layers)}"
        if task == "detect":
            head = Detect(hname, nc, hch, legacy, self.strides)
        else:
            npr = make_divisible(min(256, max_ch) * width, 8)  # tasks.py:1082-1083
            head = Segment(hname, nc, 32, npr, hch, legacy, self.strides)
        self.layers.append((hf, head))
        self.head = head
        self.save = sorted({x % len(self.layers) for f, _ in self.layers for x in ([f] if isinstance(f, int) else f)
                            if x != -1})

    def _probe_strides(self, hf, s=256):
        """tasks.py:333-344: the reference sends a 256 x 256 zero image through the graph and sets stride = 256 / height of every
        Detect input.  Heights follow from the layers' conv arithmetic alone (8, 16, 32 for the stock graphs)."""
        hs: List[int] = []
        for f, m in self.layers:
            h_in = s if not hs else hs[f if isinstance(f, int) else f[0]]
            conv = m.cv2 if isinstance(m, SCDown) else m  # SCDown: its depthwise conv carries the stride
            if m == "up":
                h = 2 * h_in
            elif isinstance(conv, Conv):
                h = (h_in + 2 * conv.p - conv.k) // conv.s + 1
            else:
                h = h_in
            hs.append(h)
        return [float(s) / hs[x] for x in hf]

    def param_specs(self) -> List[Tuple[str, Tuple[int, ...]]]:
        out = []
        for _, m in self.layers:
            if not isinstance(m, str):
                out.extend(m.specs())
        return out

    def num_params(self):
        """Count as the reference does (Parameters only: BN running stats are buffers)."""
        return sum(int(np.prod(s)) for n, s in self.param_specs() if "running_" not in n)

    def forward(self, P: Dict[str, torch.Tensor], x: torch.Tensor, return_layers: bool = False):
        """nn/tasks.py:138-165.  return_layers: also return the list of every top-level layer's output (what the fixtures'
        `layer0_i` entries hold for the reference itself)."""
        y: List = []
        for f, m in self.layers:
            if f != -1:
                x = y[f] if isinstance(f, int) else [x if j == -1 else y[j] for j in f]
            if m == "up":
                x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            elif m == "cat":
                x = torch.cat(x, 1)
            else:
                x = m(P, x)
            y.append(x)
        return (x, y) if return_layers else x


# --------------------------------------------------------------------------------------------
# Deterministic synthetic parameters, addressed by state_dict name.  Shared by
# tests/golden/make_fixtures.py (which writes them INTO the reference model) and the tests.
# --------------------------------------------------------------------------------------------
CLS_BIAS = -6.0  # shifts the class logits so that O(1 %) of anchors pass conf 0.25


def synth_param(name: str, shape: Sequence[int], seed: int = 0) -> torch.Tensor:
    rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
    shape = tuple(int(s) for s in shape)
    if name.endswith("dfl.conv.weight"):
        return torch.arange(16, dtype=torch.float32).view(shape)
    if name.endswith("bn.weight") or name.endswith("gn.weight"):
        a = rng.uniform(0.7, 1.3, shape)
    elif name.endswith("bn.bias") or name.endswith("running_mean"):
        a = rng.uniform(-0.3, 0.3, shape)
    elif name.endswith("running_var"):
        a = rng.uniform(0.5, 1.5, shape)
    elif name.endswith(".bias"):
        if ".cv3." in name:
            a = rng.uniform(CLS_BIAS - 0.5, CLS_BIAS + 0.5, shape)
        elif ".cv2." in name:
            a = rng.uniform(0.5, 1.5, shape)
        else:
            a = rng.uniform(-0.2, 0.2, shape)
    else:
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        if "upsample" in name:  # ConvTranspose2d weight (cin, cout, 2, 2): one tap per output pixel
            fan_in = shape[0]
        a = rng.standard_normal(shape) * math.sqrt(2.0 / max(fan_in, 1))
    return torch.from_numpy(np.asarray(a, dtype=np.float32))


def synth_params(model: Model, seed: int = 0) -> Dict[str, torch.Tensor]:
    return {n: synth_param(n, s, seed) for n, s in model.param_specs()}
