"""Tensor-level wrappers over the stand-alone C entry points (same kernels the engine launches).

Used by the per-module hooks (plugin.py) and by the parity tests.  Activations are NHWC fp16 torch tensors
(``x.permute(0, 2, 3, 1).contiguous()`` of the reference's BCHW tensors); every call runs on the current stream.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import lib as L


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def pack_conv_weight(w: torch.Tensor, b: Optional[torch.Tensor], device) -> tuple:
    """(Cout, Cin, k, k) fp32 [+ bias] -> packed fp16 [CoutPad][Kpad] + fp32 [CoutPad] on `device`."""
    cout, cin, k, _ = w.shape
    cp, kp = C.c_int(), C.c_int()
    L.check(L.lib.bsy_conv_packed_dims(cout, cin, k, C.byref(cp), C.byref(kp)))
    w = w.detach().float().cpu()
    if cin == 3:  # image conv: zero 4th input channel (csrc/image_conv.h)
        w = torch.cat([w, torch.zeros(cout, 1, k, k)], 1)
        cin = 4
    K = k * k * cin
    wp = torch.zeros(cp.value, kp.value, dtype=torch.float16)
    wp[:cout, :K] = w.permute(0, 2, 3, 1).reshape(cout, K).half()
    bp = torch.zeros(cp.value, dtype=torch.float32)
    if b is not None:
        bp[:cout] = b.detach().float().cpu()
    return wp.to(device), bp.to(device)


def conv2d_nhwc(x: torch.Tensor, wp: torch.Tensor, bp: torch.Tensor, cout: int, k: int, s: int = 1, act: bool = True,
                res: Optional[torch.Tensor] = None, out_f32: bool = False, cin: Optional[int] = None,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x (B,H,W,ld) fp16 (uses the first `cin` channels) -> (B,OH,OW,cout).  Conv.forward_fuse (conv.py:149-151)."""
    assert x.dtype == torch.float16 and x.is_contiguous() and x.dim() == 4
    B, H, W, ld = x.shape
    cin = ld if cin is None else cin
    p = k // 2
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    if out is None:
        ldy = (cout + 3) // 4 * 4 if out_f32 else (cout + 7) // 8 * 8
        out = torch.zeros((B, OH, OW, ldy), dtype=torch.float32 if out_f32 else torch.float16, device=x.device)
    L.check(L.lib.bsy_conv2d(_p(x), ld, B, H, W, cin, _p(wp), _p(bp), _p(out), out.shape[-1], cout, k, s, int(act),
                             _p(res), res.shape[-1] if res is not None else 0, int(out_f32), _stream(x)))
    return out


def conv2d_nhwc_f32(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, k: int, s: int = 1, act: bool = True,
                    res: Optional[torch.Tensor] = None, cin: Optional[int] = None, impl: int = 0) -> torch.Tensor:
    """fp32 engine mode's conv: x (B,H,W,ld) f32 (first `cin` channels), w (Cout,Cin,k,k) f32, b (Cout) -> (B,OH,OW,Cout) f32.
    impl 0 = the engine's routing, 1 = scalar kernel, 2 = fp32 MFMA kernel (bsy_conv2d_f32)."""
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    B, H, W, ld = x.shape
    cin = ld if cin is None else cin
    cout = w.shape[0]
    assert tuple(w.shape) == (cout, cin, k, k)
    wk = w.detach().float().permute(2, 3, 1, 0).reshape(k * k * cin, cout).contiguous().to(x.device)
    bk = b.detach().float().contiguous().to(x.device)
    p = k // 2
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    ldy = (cout + 3) // 4 * 4
    out = torch.zeros((B, OH, OW, ldy), dtype=torch.float32, device=x.device)
    L.check(L.lib.bsy_conv2d_f32(_p(x), ld, B, H, W, cin, _p(wk), _p(bk), _p(out), ldy, cout, k, s, int(act), _p(res),
                                 res.shape[-1] if res is not None else 0, impl, _stream(x)))
    return out


def conv2d_nhwc_f32x(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, k: int, s: int = 1, act: bool = True,
                     res: Optional[torch.Tensor] = None, cin: Optional[int] = None) -> torch.Tensor:
    """fp32x engine mode's conv (bsy_conv2d_f32x): fp32 NHWC in / out, operands split into f16 pairs, three f16 MFMAs per product.
    x (B,H,W,ld) f32 (first `cin` channels), w (Cout,Cin,k,k) f32, b (Cout) -> (B,OH,OW,Cout up 4) f32."""
    from .weights import split_f16_planes
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    B, H, W, ld = x.shape
    cin = ld if cin is None else cin
    cout = w.shape[0]
    assert tuple(w.shape) == (cout, cin, k, k)
    hi, lo = split_f16_planes(w.detach().float().cpu().permute(0, 2, 3, 1).reshape(cout, k * k * cin))
    hi, lo = hi.contiguous().to(x.device), lo.contiguous().to(x.device)
    bk = b.detach().float().contiguous().to(x.device)
    p = k // 2
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    ldy = (cout + 3) // 4 * 4
    out = torch.zeros((B, OH, OW, ldy), dtype=torch.float32, device=x.device)
    L.check(L.lib.bsy_conv2d_f32x(_p(x), ld, B, H, W, cin, _p(hi), _p(lo), hi.shape[1], _p(bk), _p(out), ldy, cout, k, s, int(act),
                                  _p(res), res.shape[-1] if res is not None else 0, _stream(x)))
    return out


def conv_first(img: torch.Tensor, w: torch.Tensor, b: torch.Tensor, k: int = 3, s: int = 2, act: bool = True):
    """img BCHW fp16/fp32; w (Cout,3,3,3) fp32; -> NHWC fp16."""
    assert img.is_contiguous() and img.shape[1] == 3
    B, _, H, W = img.shape
    cout = w.shape[0]
    wp, bp = pack_conv_weight(w, b, img.device)
    p = k // 2
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    out = torch.empty((B, OH, OW, cout), dtype=torch.float16, device=img.device)
    L.check(L.lib.bsy_conv_first(_p(img), L.dtype_code(img.dtype), B, H, W, _p(wp), _p(bp), _p(out), cout, cout, k, s,
                                 int(act), _stream(img)))
    return out


def stem_fused(img: torch.Tensor, w0: torch.Tensor, b0: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor,
               act: bool = True) -> torch.Tensor:
    """Layers 0 + 1 in one launch: img BCHW fp16/fp32; w0 (C0,3,3,3), w1 (C1,C0,3,3) fp32 -> NHWC fp16 (B,H/4,W/4,C1)."""
    assert img.is_contiguous() and img.shape[1] == 3
    B, _, H, W = img.shape
    c0, c1 = w0.shape[0], w1.shape[0]
    w0p, b0p = pack_conv_weight(w0, b0, img.device)
    w1p, b1p = pack_conv_weight(w1, b1, img.device)
    oh0, ow0 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((B, (oh0 - 1) // 2 + 1, (ow0 - 1) // 2 + 1, c1), dtype=torch.float16, device=img.device)
    L.check(L.lib.bsy_stem_fused(_p(img), L.dtype_code(img.dtype), B, H, W, _p(w0p), _p(b0p), c0, _p(w1p), _p(b1p), c1,
                                 _p(out), c1, int(act), _stream(img)))
    return out


def bottleneck_fused(x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x + cv2(cv1(x)) in one launch.  x, out: (B,H,W,C) fp16 NHWC, possibly channel slices of wider buffers
    (stride(2) = leading dimension); w1 (CH,C,3,3), w2 (C,CH,3,3) fp32 (BN already folded)."""
    B, H, W, c = x.shape
    assert x.stride(3) == 1 and x.stride(1) == W * x.stride(2) and x.stride(0) == H * W * x.stride(2)
    ch = w1.shape[0]
    w1p, b1p = pack_conv_weight(w1, b1, x.device)
    w2p, b2p = pack_conv_weight(w2, b2, x.device)
    if out is None:
        out = torch.empty((B, H, W, c), dtype=torch.float16, device=x.device)
    assert out.shape == x.shape and out.stride(3) == 1 and out.stride(1) == W * out.stride(2)
    L.check(L.lib.bsy_bottleneck_fused(_p(x), x.stride(2), B, H, W, c, ch, _p(w1p), _p(b1p), _p(w2p), _p(b2p), _p(out),
                                       out.stride(2), 1, _stream(x)))
    return out


def c3k2_fused(x: torch.Tensor, w1, b1, wa, ba, wb, bb, w4, b4, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Whole C3k2 block (c3k = False, one Bottleneck; block.py:3796-3804) in one launch: x (B,H,W,Cin) fp16 NHWC view ->
    (B,H,W,C2).  w1 (2c,Cin,1,1), wa (c/2,c,3,3), wb (c,c/2,3,3), w4 (C2,3c,1,1) fp32 with BN folded, biases fp32."""
    B, H, W, cin = x.shape
    assert x.stride(3) == 1 and x.stride(1) == W * x.stride(2) and x.stride(0) == H * W * x.stride(2)
    c, c2 = wb.shape[0], w4.shape[0]
    packs = [pack_conv_weight(w, b, x.device) for w, b in ((w1, b1), (wa, ba), (wb, bb), (w4, b4))]
    if out is None:
        out = torch.empty((B, H, W, c2), dtype=torch.float16, device=x.device)
    assert out.shape == (B, H, W, c2) and out.stride(3) == 1 and out.stride(1) == W * out.stride(2)
    L.check(L.lib.bsy_c3k2_fused(_p(x), x.stride(2), B, H, W, cin, c, c2, _p(packs[0][0]), _p(packs[0][1]), _p(packs[1][0]),
                                 _p(packs[1][1]), _p(packs[2][0]), _p(packs[2][1]), _p(packs[3][0]), _p(packs[3][1]), _p(out),
                                 out.stride(2), _stream(x)))
    return out


def dwpw_fused(x: torch.Tensor, wd: torch.Tensor, bd: torch.Tensor, w: torch.Tensor, b: torch.Tensor, act: bool = True) -> torch.Tensor:
    """DWConv 3x3 + SiLU -> Conv 1x1 (+act) in one launch.  x (B,H,W,C) fp16; wd (C,1,3,3), bd (C); w (C2,C,1,1), b (C2)."""
    B, H, W, Cc = x.shape
    c2 = w.shape[0]
    wdp = wd.detach().float().cpu().view(Cc, 9).t().contiguous().to(x.device)
    bdp = bd.detach().float().contiguous().to(x.device)
    wp, bp = pack_conv_weight(w, b, x.device)
    out = torch.empty((B, H, W, c2), dtype=torch.float16, device=x.device)
    L.check(L.lib.bsy_dwpw_fused(_p(x), Cc, B, H, W, Cc, _p(wdp), _p(bdp), _p(wp), _p(bp), _p(out), c2, c2, int(act), _stream(x)))
    return out


def dwconv_nhwc(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, stride: int = 1, act: bool = False) -> torch.Tensor:
    """Depthwise kh x kw conv, padding k/2 (csrc/bsyolo_ops.hip).  x (B,H,W,C) fp16; w (C,1,kh,kw) fp32; b (C)."""
    B, H, W, Cc = x.shape
    kh, kw = int(w.shape[2]), int(w.shape[3])
    wp = w.detach().float().cpu().view(Cc, kh * kw).t().contiguous().to(x.device)
    bp = b.detach().float().contiguous().to(x.device)
    OH, OW = (H + 2 * (kh // 2) - kh) // stride + 1, (W + 2 * (kw // 2) - kw) // stride + 1
    out = torch.empty((B, OH, OW, Cc), dtype=torch.float16, device=x.device)
    L.check(L.lib.bsy_dwconv(_p(x), Cc, B, H, W, Cc, kh, kw, stride, _p(wp), Cc, _p(bp), _p(out), Cc, int(act), _stream(x)))
    return out


def ela_nhwc(x: torch.Tensor, wsp: torch.Tensor, wch: torch.Tensor, gnw: torch.Tensor, gnb: torch.Tensor, coef) -> torch.Tensor:
    """ELA.forward (nn/Addmodules/ELA.py:77-101).  x (B,H,W,C) fp16; wsp / wch (C,1,k) Conv1d weights; coef = the three
    sigmoids (ch, sp, res)."""
    B, H, W, Cc = x.shape
    k = int(wsp.shape[-1])
    dev = x.device
    f = lambda t: t.detach().float().reshape(-1).contiguous().to(dev)  # noqa: E731
    a, b_, c, d = f(wsp), f(wch), f(gnw), f(gnb)
    scratch = torch.empty(int(L.lib.bsy_ela_scratch_bytes(B, H, W, Cc)) // 4, dtype=torch.float32, device=dev)
    out = torch.empty_like(x)
    cf = (C.c_float * 3)(*[float(v) for v in coef])
    L.check(L.lib.bsy_ela(_p(x), Cc, B, H, W, Cc, k, _p(a), _p(b_), _p(c), _p(d), cf, _p(scratch), _p(out), Cc, _stream(x)))
    return out


def dwconv3x3_nhwc(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, act: bool = True,
                   res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x (B,H,W,C) fp16; w (C,1,3,3) fp32.  DWConv (conv.py:224-229)."""
    B, H, W, Cc = x.shape
    wp = w.detach().float().cpu().view(Cc, 9).t().contiguous().to(x.device)
    bp = b.detach().float().contiguous().to(x.device)
    out = torch.empty_like(x)
    L.check(L.lib.bsy_dwconv3x3(_p(x), Cc, B, H, W, Cc, _p(wp), _p(bp), _p(out), Cc, int(act), _p(res),
                                res.shape[-1] if res is not None else 0, _stream(x)))
    return out


def sppf_pool_nhwc(x1: torch.Tensor) -> torch.Tensor:
    """x1 (B,H,W,C) -> (B,H,W,4C) = cat[x1, m(x1), m(m(x1)), m(m(m(x1)))], m = MaxPool2d(5,1,2) (block.py:3145-3149)."""
    B, H, W, Cc = x1.shape
    buf = torch.zeros((B, H, W, 4 * Cc), dtype=torch.float16, device=x1.device)
    buf[..., :Cc] = x1
    L.check(L.lib.bsy_sppf_pool(_p(buf), 4 * Cc, B, H, W, Cc, _stream(x1)))
    return buf


def attention_nhwc(qkv: torch.Tensor, heads: int, key_dim: int, head_dim: int, scale: float) -> torch.Tensor:
    """qkv (B,N,heads*(2kd+hd)) fp16 in [q|k|v] (all-heads-contiguous) order -> (B,N,heads*hd) (block.py:4279-4286)."""
    B, N, ld = qkv.shape
    out = torch.empty((B, N, heads * head_dim), dtype=torch.float16, device=qkv.device)
    L.check(L.lib.bsy_attention(_p(qkv), ld, B, N, heads, key_dim, head_dim, float(scale), _p(out), heads * head_dim,
                                _stream(qkv)))
    return out


def detect_decode(box: Sequence[torch.Tensor], cls: Sequence[torch.Tensor], hw: Sequence[tuple],
                  strides: Sequence[float], nc: int, out_dtype=torch.float32,
                  msk: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
    """box[l] (B*h*w, 64) f32, cls[l] (B*h*w, ldc) f32 -> (B, 4+nc+nm, A) (head.py:100-131)."""
    nl = len(box)
    B = box[0].shape[0] // (hw[0][0] * hw[0][1])
    nm = msk[0].shape[-1] if msk else 0
    A = sum(h * w for h, w in hw)
    y = torch.empty((B, 4 + nc + nm, A), dtype=out_dtype, device=box[0].device)
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    bp = (vp * nl)(*[t.data_ptr() for t in box])
    cp = (vp * nl)(*[t.data_ptr() for t in cls])
    mp = (vp * nl)(*[t.data_ptr() for t in msk]) if msk else None
    ldb = (i32 * nl)(*[t.shape[-1] for t in box])
    ldc = (i32 * nl)(*[t.shape[-1] for t in cls])
    ldm = (i32 * nl)(*[t.shape[-1] for t in msk]) if msk else None
    hh = (i32 * nl)(*[h for h, _ in hw])
    ww = (i32 * nl)(*[w for _, w in hw])
    st = (f32 * nl)(*[float(s) for s in strides])
    L.check(L.lib.bsy_detect_decode(bp, ldb, cp, ldc, mp, ldm, hh, ww, st, nl, B, nc, nm, _p(y), L.dtype_code(out_dtype),
                                    _stream(y)))
    return y
