"""Weight preparation: BN folding + packing into the engine's device blob.

  * fold_conv_bn  -- the arithmetic of utils/torch_utils.py:242-269 fuse_conv_and_bn (called per Conv by
                     BaseModel.fuse, nn/tasks.py:209-215): W' = diag(g / sqrt(var + eps)) W,  b' = beta - g mu / sqrt(var + eps).
                     Accepts either an un-fused state_dict (conv.weight + bn.*) or an already fused one
                     (conv.weight + conv.bias), which is what a reference model holds after AutoBackend's fuse().
  * pack layouts  -- conv   : fp16 [CoutPad128][Kpad32], K order (kh, kw, cin)  (conv_mfma.hip), bias fp32 [CoutPad128]
                     first  : same as conv with cin = 3 (K = 27 -> 32)  (conv_first.hip)
                     dw     : fp32 [9][C] (elementwise.hip)
"""
from __future__ import annotations

from typing import Dict, Mapping

import numpy as np
import torch

from .plan import Plan, WRec

BN_EPS = 1e-3  # utils/torch_utils.py:424


def _f32(t) -> torch.Tensor:
    return torch.as_tensor(t).detach().to("cpu", torch.float32)


def fold_conv_bn(sd: Mapping[str, torch.Tensor], prefix: str, eps: float = BN_EPS):
    """(weight, bias) fp32 of `prefix` (a reference Conv module) with BatchNorm folded in."""
    w = _f32(sd[prefix + ".conv.weight"])
    if prefix + ".bn.weight" in sd:
        g, beta = _f32(sd[prefix + ".bn.weight"]), _f32(sd[prefix + ".bn.bias"])
        mu, var = _f32(sd[prefix + ".bn.running_mean"]), _f32(sd[prefix + ".bn.running_var"])
        scale = g / torch.sqrt(var + eps)
        w = w * scale.view(-1, 1, 1, 1)
        b = beta - g * mu / torch.sqrt(var + eps)
        if prefix + ".conv.bias" in sd:
            b = b + scale * _f32(sd[prefix + ".conv.bias"])
    else:
        b = _f32(sd[prefix + ".conv.bias"])
    return w, b


def _align(n: int, a: int = 256) -> int:
    return (n + a - 1) // a * a


def split_f16_planes(w2d: torch.Tensor):
    """(Cout, K) f32 -> (hi, lo) f16 planes [Cout][K rounded up to 32] with w ~= hi + lo (csrc/conv32x_mfma.hip): hi = f16(w),
    lo = f16(w - hi); the subtraction is exact in f32, so the pair carries w to ~22 bits (less where lo falls into f16's
    subnormal range: |w| < 2^-3 keeps an absolute 2^-25).  Values beyond the f16 range saturate at +-65504 like the kernel's
    pixel split (never an infinity)."""
    cout, K = w2d.shape
    kp = (K + 31) // 32 * 32
    w = torch.zeros(cout, kp, dtype=torch.float32)
    w[:, :K] = w2d
    hi = w.clamp(-65504.0, 65504.0).to(torch.float16)
    lo = (w - hi.float()).clamp(-65504.0, 65504.0).to(torch.float16)
    return hi, lo


def pack_record(sd: Mapping[str, torch.Tensor], r: WRec, eps: float = BN_EPS, f32: bool = False, split: bool = False):
    """-> (weight bytes, bias bytes) for one op.  f32: the fp32 correctness mode's layout for dense convs -- f32
    [k*k*cin][cout] (K order (kh, kw, cin) as in the fp16 layout, cout fastest), bias f32 [cout], no padding
    (csrc/ref32.hip); depthwise / ELA records are f32 in both modes.  split (fp32x mode, with f32): the f32 matrix is followed
    by the two f16 planes of split_f16_planes, each part padded to 256 bytes (csrc/engine.hip run_op_f32 computes the same
    offsets)."""
    if r.kind == "ela":
        # ELA (nn/Addmodules/ELA.py:36-72): [spatial_conv (C,k)][ch_att.2 (C,k)][gn.weight][gn.bias]; the three scalar mixing
        # weights enter the op record as their sigmoids
        c, k = r.cout, r.k
        wsp, wch = _f32(sd[r.name + ".spatial_conv.weight"]).reshape(c, k), _f32(sd[r.name + ".ch_att.2.weight"]).reshape(c, k)
        gw, gb = _f32(sd[r.name + ".gn.weight"]), _f32(sd[r.name + ".gn.bias"])
        r.coef = tuple(float(torch.sigmoid(_f32(sd[r.name + "." + s]).view(-1)[0])) for s in ("ch_weight", "sp_weight", "res_weight"))
        return torch.cat([wsp.reshape(-1), wch.reshape(-1), gw, gb]).contiguous().numpy().tobytes(), b""
    if r.kind in ("plain", "dwg_plain"):
        w, b = _f32(sd[r.name + ".weight"]), _f32(sd[r.name + ".bias"])
    elif r.kind == "deconv":
        # nn.ConvTranspose2d(c, c, 2, 2, 0) (block.py:91): out[2i+dy, 2j+dx] = W[:, :, dy, dx]^T x[i, j] + b, i.e. one
        # 1x1 conv per output phase; weight layout (cin, cout, 2, 2)
        wt, b = _f32(sd[r.name + ".weight"]), _f32(sd[r.name + ".bias"])
        dy, dx = r.tap
        w = wt[:, :, dy, dx].t().contiguous().view(r.cout, r.cin, 1, 1)
    elif r.kind == "conv2":  # two Conv modules on the same input, stacked along cout (plan.py conv(name2=...))
        (w, b), (w2, b2) = fold_conv_bn(sd, r.name, eps), fold_conv_bn(sd, r.post, eps)
        w, b = torch.cat([w, w2]), torch.cat([b, b2])
    else:
        w, b = fold_conv_bn(sd, r.name, eps)
    if r.rows is not None or r.cols is not None:
        # the op's channels are a re-laid-out, zero-padded copy of the module's (plan.py _pmsfa_padded)
        dw = w.shape[1] == 1 and r.kind in ("dw", "dwg", "dwg_plain", "dwg_ext")
        if r.rows is not None:
            idx = torch.as_tensor(r.rows, dtype=torch.long)
            assert int(idx.max()) < w.shape[0], (r.name, tuple(w.shape), int(idx.max()))
            w2, b2 = torch.zeros(len(idx), *w.shape[1:]), torch.zeros(len(idx))
            w2[idx >= 0], b2[idx >= 0] = w[idx[idx >= 0]], b[idx[idx >= 0]]
            w, b = w2, b2
        if r.cols is not None and not dw:
            idx = torch.as_tensor(r.cols, dtype=torch.long)
            assert int(idx.max()) < w.shape[1], (r.name, tuple(w.shape), int(idx.max()))
            w2 = torch.zeros(w.shape[0], len(idx), *w.shape[2:])
            w2[:, idx >= 0] = w[:, idx[idx >= 0]]
            w = w2
    elif r.real_cout or r.real_cin:
        # the op runs wider than the module (plan.py detect(): widths padded to a multiple of 8): zero weights and biases in the padding
        dw = w.shape[1] == 1 and r.kind in ("dw", "dwg", "dwg_plain", "dwg_ext")
        co, ci = r.real_cout or r.cout, (1 if dw else (r.real_cin or r.cin))
        assert tuple(w.shape[:2]) == (co, ci), (r.name, tuple(w.shape), (co, ci))
        w2 = torch.zeros(r.cout, 1 if dw else r.cin, *w.shape[2:])
        w2[:co, :ci] = w
        b2 = torch.zeros(r.cout)
        b2[:co] = b
        w, b = w2, b2
    if r.kind == "first_s2d" and not f32:
        # 6x6 stride-2 pad-2 image conv (YOLOv5u's stem, cfg/models/v5/yolov5.yaml:16) = 3x3 stride-1 pad-1 conv over the
        # space-to-depth image [Y][X][(dy, dx, c)] (csrc/elementwise.hip s2d_kernel): input row 2 oy - 2 + kh with kh = 2 a + dy is
        # s2d row oy - 1 + a, so w3[co][(dy, dx, c)][a][b] = w[co][c][2a + dy][2b + dx]; channels 12 .. 15 are zero padding
        cout = r.cout
        assert tuple(w.shape) == (cout, 3, 6, 6), (r.name, tuple(w.shape))
        w3 = torch.zeros(cout, 16, 3, 3)
        for dy in (0, 1):
            for dx in (0, 1):
                w3[:, (dy * 2 + dx) * 3:(dy * 2 + dx) * 3 + 3] = w[:, :, dy::2, dx::2]
        w = w3
        r = WRec(name=r.name, kind="conv", cout=cout, cin=16, k=3)
    if r.kind in ("conv", "conv2", "plain", "first", "first_s2d", "deconv"):
        cout, cin, k = r.cout, r.cin, r.k
        assert tuple(w.shape) == (cout, cin, k, k), (r.name, tuple(w.shape), (cout, cin, k, k))
        if r.perm is not None:
            idx = torch.as_tensor(r.perm, dtype=torch.long)
            w, b = w[idx], b[idx]
        if f32:
            wb = w.permute(2, 3, 1, 0).reshape(k * k * cin, cout).contiguous().numpy().tobytes()
            if split:
                hi, lo = split_f16_planes(w.permute(0, 2, 3, 1).reshape(cout, k * k * cin))
                hb = hi.numpy().tobytes()
                wb = wb + b"\0" * (_align(len(wb)) - len(wb)) + hb + b"\0" * (_align(len(hb)) - len(hb)) + lo.numpy().tobytes()
            return wb, b.contiguous().numpy().tobytes()
        if r.kind == "first":  # image conv: zero 4th input channel, k = (kh, kw, c4)  (csrc/image_conv.h)
            w = torch.cat([w, torch.zeros(cout, 4 - cin, k, k)], 1)
            cin = 4
        K = k * k * cin
        cp, kp = (cout + 127) // 128 * 128, (K + 31) // 32 * 32
        wp = torch.zeros(cp, kp, dtype=torch.float16)
        wp[:cout, :K] = w.permute(0, 2, 3, 1).reshape(cout, K).to(torch.float16)
        bp = torch.zeros(cp, dtype=torch.float32)
        bp[:cout] = b
        return wp.numpy().tobytes(), bp.numpy().tobytes()
    if r.kind in ("dwg", "dwg_plain", "dwg_ext"):
        # generic depthwise (csrc/bsyolo_ops.hip): f32 [kh*kw][C] + bias [C]
        kh, kw = r.k, r.kw or r.k
        if r.kind == "dwg_ext":
            # PMSFA.conv3 (block.py:3042): a depthwise 7x7 over the first half of its input, extended with an identity
            # kernel (centre tap 1, bias 0, no activation) so the second half passes through untouched
            half = r.cout // 2
            assert tuple(w.shape) == (half, 1, kh, kw), (r.name, tuple(w.shape))
            ident = torch.zeros(r.cout - half, 1, kh, kw)
            ident[:, 0, kh // 2, kw // 2] = 1.0
            w, b = torch.cat([w, ident]), torch.cat([b, torch.zeros(r.cout - half)])
        assert tuple(w.shape) == (r.cout, 1, kh, kw), (r.name, tuple(w.shape), (r.cout, kh, kw))
        if r.post:  # a following depthwise 1x1 conv (MSCAAttention.dilconv, MSCA.py:31): y -> a*y + d, folded in
            a, d = _f32(sd[r.post + ".weight"]).view(-1), _f32(sd[r.post + ".bias"])
            w, b = w * a.view(-1, 1, 1, 1), a * b + d
        wp = w.view(r.cout, kh * kw).t().contiguous()
        return wp.numpy().tobytes(), b.contiguous().numpy().tobytes()
    if r.kind == "dw":
        assert tuple(w.shape) == (r.cout, 1, 3, 3), (r.name, tuple(w.shape))
        wp = w.view(r.cout, 9).t().contiguous()
        return wp.numpy().tobytes(), b.contiguous().numpy().tobytes()
    raise ValueError(r.kind)


def pack_plan_weights(plan: Plan, sd: Mapping[str, torch.Tensor], eps: float = BN_EPS) -> bytes:
    """Packs every record of `plan` (sets w_off / b_off on the records) -> host blob (fp32 layouts for an fp32-mode plan)."""
    chunks, off = [], 0
    for r in plan.wrecs.values():
        wb, bb = pack_record(sd, r, eps, f32=getattr(plan, "f32_mode", False), split=getattr(plan, "split_f16", False))
        r.w_off = off
        chunks.append(wb)
        pad = _align(len(wb)) - len(wb)
        chunks.append(b"\0" * pad)
        off += len(wb) + pad
        r.b_off = off
        chunks.append(bb)
        pad = _align(len(bb)) - len(bb)
        chunks.append(b"\0" * pad)
        off += len(bb) + pad
    return b"".join(chunks)


def adopt_offsets(plan: Plan, packed: Plan) -> None:
    """Copy blob offsets from an already packed plan of the same model (other input shape)."""
    for k, r in plan.wrecs.items():
        src = packed.wrecs[k]
        assert (src.kind, src.cout, src.cin, src.k) == (r.kind, r.cout, r.cin, r.k)
        r.w_off, r.b_off, r.coef = src.w_off, src.b_off, src.coef


def synth_state_dict(plan: Plan, seed: int = 0, cls_gain: float = 0.1, cls_bias: float = -4.21) -> Dict[str, torch.Tensor]:
    """Seeded random parameters for benchmarking without a checkpoint (no weights ship with the reference:
    .MISSING_LARGE_BLOBS).  He-normal conv weights, randomised BN statistics.  The class head is damped (cls_gain) and
    shifted (cls_bias) so that a realistic share of anchors passes conf 0.25: the defaults put 1.5 % of the anchors
    of YOLO11s @640 (seed 0) above 0.25 (SURVEY 8d asks for 1-2 %), i.e. ~126 NMS candidates per image."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for r in plan.wrecs.values():
        if r.kind == "deconv":
            if r.name + ".weight" not in sd:
                sd[r.name + ".weight"] = torch.randn(r.cin, r.cout, 2, 2, generator=g) * (2.0 / r.cin) ** 0.5
                sd[r.name + ".bias"] = torch.rand(r.cout, generator=g) * 0.4 - 0.2
            continue
        if r.kind == "ela":
            for s_ in ("ch_weight", "sp_weight", "res_weight"):
                sd[f"{r.name}.{s_}"] = torch.randn(1, generator=g)
            sd[r.name + ".spatial_conv.weight"] = torch.randn(r.cout, 1, r.k, generator=g) * (2.0 / r.k) ** 0.5
            sd[r.name + ".ch_att.2.weight"] = torch.randn(r.cout, 1, r.k, generator=g) * (2.0 / r.k) ** 0.5
            sd[r.name + ".gn.weight"] = torch.rand(r.cout, generator=g) * 0.6 + 0.7
            sd[r.name + ".gn.bias"] = torch.rand(r.cout, generator=g) * 0.6 - 0.3
            continue
        if r.kind == "dwg_plain":
            kh, kw = r.k, r.kw or r.k
            sd[r.name + ".weight"] = torch.randn(r.cout, 1, kh, kw, generator=g) * (2.0 / (kh * kw)) ** 0.5
            sd[r.name + ".bias"] = torch.rand(r.cout, generator=g) * 0.4 - 0.2
            if r.post and r.post + ".weight" not in sd:
                sd[r.post + ".weight"] = torch.rand(r.cout, 1, 1, 1, generator=g) + 0.5
                sd[r.post + ".bias"] = torch.rand(r.cout, generator=g) * 0.4 - 0.2
            continue
        if r.kind in ("dwg", "dwg_ext"):
            kh, kw = r.k, r.kw or r.k
            co = r.real_cout or (r.cout // 2 if r.kind == "dwg_ext" else r.cout)
            sd[r.name + ".conv.weight"] = torch.randn(co, 1, kh, kw, generator=g) * (2.0 / (kh * kw)) ** 0.5
            sd[r.name + ".bn.weight"] = torch.rand(co, generator=g) * 0.6 + 0.7
            sd[r.name + ".bn.bias"] = torch.rand(co, generator=g) * 0.6 - 0.3
            sd[r.name + ".bn.running_mean"] = torch.rand(co, generator=g) * 0.6 - 0.3
            sd[r.name + ".bn.running_var"] = torch.rand(co, generator=g) + 0.5
            continue
        if r.kind == "plain":
            rci = r.real_cin or r.cin
            fan = rci * r.k * r.k
            sd[r.name + ".weight"] = torch.randn(r.real_cout or r.cout, rci, r.k, r.k, generator=g) * (2.0 / fan) ** 0.5
            if ".cv3." in r.name:
                sd[r.name + ".weight"] *= cls_gain
                sd[r.name + ".bias"] = torch.full((r.real_cout or r.cout,), float(cls_bias))
            elif ".cv2." in r.name:
                sd[r.name + ".bias"] = torch.rand(r.cout, generator=g) + 0.5
            else:
                sd[r.name + ".bias"] = torch.rand(r.cout, generator=g) * 0.4 - 0.2
            continue
        cin = 1 if r.kind == "dw" else (r.real_cin or r.cin)
        fan = cin * r.k * r.k
        # kind "conv2": two modules of half the width each, drawn in the order the unmerged plan draws them
        for nm, co in (((r.name, (r.real_cout or r.cout) // 2), (r.post, (r.real_cout or r.cout) // 2)) if r.kind == "conv2" else ((r.name, r.real_cout or r.cout),)):
            sd[nm + ".conv.weight"] = torch.randn(co, cin, r.k, r.k, generator=g) * (2.0 / fan) ** 0.5
            sd[nm + ".bn.weight"] = torch.rand(co, generator=g) * 0.6 + 0.7
            sd[nm + ".bn.bias"] = torch.rand(co, generator=g) * 0.6 - 0.3
            sd[nm + ".bn.running_mean"] = torch.rand(co, generator=g) * 0.6 - 0.3
            sd[nm + ".bn.running_var"] = torch.rand(co, generator=g) + 0.5
    return sd
