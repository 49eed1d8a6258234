"""GPU parity: every HIP kernel, called through the C ABI (libbsyolo_hip.so), against the CPU oracle.

Run on the MI355X box with ``pytest -m gpu``.  Float kernels: tolerance stated per test (fp16 storage, fp32
accumulation).  Integer / control-flow work (NMS decisions, letterbox pixels) is compared bit-exactly.
"""
import json
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from bs_yolo_amd import lib as L  # noqa: F401  (fails loudly when the .so is missing)
    from bs_yolo_amd import letterbox as HLB
    from bs_yolo_amd import nms as HN
    from bs_yolo_amd import ops as O
    from bs_yolo_amd.engine import YoloEngine
    from bs_yolo_amd.graphs import stock_cfg

from oracle import letterbox_ref as LB
from oracle import postproc_ref as PP
from oracle import yolo_ref as R

DEV = "cuda:0"


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def h16(t):
    """fp16-round a fp32 tensor (what the kernels see)."""
    return t.half().float()


# ------------------------------------------------------------------------------------------------------------
# implicit-GEMM conv (conv_mfma.hip)
# ------------------------------------------------------------------------------------------------------------
CONV_CASES = [
    # B, H, W, cin, cout, k, s, act, res, f32
    (2, 12, 10, 16, 32, 1, 1, True, False, False),
    (2, 12, 10, 16, 24, 3, 1, True, False, False),
    (1, 13, 11, 8, 16, 3, 2, True, False, False),      # odd extents, thin K (Cin=8 -> K=72)
    (2, 20, 20, 64, 64, 3, 1, True, True, False),      # Bottleneck.cv2 with shortcut
    (2, 16, 16, 96, 128, 1, 1, True, False, False),    # C3k2.cv2 over a 3-way cat
    (1, 40, 40, 128, 256, 3, 2, True, False, False),   # stride-2 downsample, 2 cout tiles
    (3, 9, 7, 32, 80, 1, 1, False, False, True),       # Detect cls conv: nc=80, bias, fp32 out, no act
    (2, 8, 8, 64, 5, 1, 1, False, False, True),        # nc=5: ragged cout (not a multiple of 4)
    (1, 7, 5, 256, 512, 1, 1, True, False, False),     # 4 cout tiles, tiny M
    (2, 33, 31, 32, 64, 3, 1, True, False, False),     # M tail inside a 256-pixel tile
    (1, 20, 20, 512, 256, 3, 1, True, True, False),    # deep K = 4608
    (2, 10, 10, 16, 16, 3, 1, True, True, False),      # yolo11n bottleneck widths
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_matches_oracle(case):
    B, H, W, cin, cout, k, s, act, use_res, f32 = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = h16(torch.randn(B, cin, H, W, generator=g))
    w = h16(torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.5
    y = F.conv2d(x, w, b, s, k // 2)
    if act:
        y = F.silu(y)
    res = None
    if use_res:
        res = h16(torch.randn(y.shape, generator=g))
        y = y + res
    wp, bp = O.pack_conv_weight(w, b, DEV)
    out = O.conv2d_nhwc(nhwc(x).half().to(DEV), wp, bp, cout, k, s, act,
                        res=nhwc(res).half().to(DEV) if use_res else None, out_f32=f32)
    torch.cuda.synchronize()
    got = nchw(out[..., :cout].float().cpu())
    # fp16 operands are exact on both sides; differences = fp32 summation order (+ one fp16 rounding of the output)
    tol = dict(rtol=1e-4, atol=1e-4) if f32 else dict(rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(got.numpy(), y.numpy(), **tol)


CONV32_CASES = [
    # B, H, W, cin, cout, k, s, act, res, ld extra
    (2, 12, 10, 16, 32, 1, 1, True, False, 0),
    (2, 12, 10, 16, 24, 3, 1, True, False, 8),       # view of a wider buffer
    (1, 13, 11, 8, 16, 3, 2, True, False, 0),        # odd extents, K = 72 (K tail inside a 32-wide step)
    (2, 20, 20, 64, 64, 3, 1, True, True, 0),        # Bottleneck.cv2 with shortcut
    (2, 16, 16, 96, 128, 1, 1, True, False, 0),
    (1, 40, 40, 128, 256, 3, 2, True, False, 0),     # two 128-cout tiles
    (3, 9, 7, 32, 80, 1, 1, False, False, 0),        # Detect cls conv: ragged cout inside a 128-wide tile
    (1, 7, 5, 256, 512, 1, 1, True, False, 0),       # M = 35: one partial pixel tile
    (2, 33, 31, 40, 68, 3, 1, True, True, 0),        # Cin = 40 (a K-step straddles taps), cout 68
    (1, 20, 20, 512, 256, 3, 1, True, True, 0),      # K = 4608
    (3, 33, 47, 64, 32, 3, 1, True, True, 0),        # 32-cout (thin) tile with shortcut, ragged last pixel tile
    (2, 24, 24, 32, 16, 3, 1, True, False, 0),       # 16 couts on the thin tile
]


@pytest.mark.parametrize("case", CONV32_CASES)
def test_conv32_mfma_equals_scalar(case):
    """fp32 engine mode: the fp32-MFMA conv (conv32_mfma.hip, v_mfma_f32_32x32x2_f32) against the scalar kernel it replaces
    (ref32.hip, one sequential fmaf chain per output) -- the SAME chain, so bit for bit -- and against torch's fp32 conv to
    summation-order tolerance (1e-4 of the output range, the fp32 mode's per-layer bound)."""
    B, H, W, cin, cout, k, s, act, use_res, ldx = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.5
    y = F.conv2d(x, w, b, s, k // 2)
    if act:
        y = F.silu(y)
    res = torch.randn(y.shape, generator=g) if use_res else None
    if use_res:
        y = y + res
    xd = torch.zeros(B, H, W, cin + ldx)
    xd[..., :cin] = nhwc(x)
    xd[..., cin:] = float("nan")  # channels outside the view must never be read into a result
    xd = xd.to(DEV)
    rd = nhwc(res).to(DEV) if use_res else None
    got = {impl: O.conv2d_nhwc_f32(xd, w, b, k, s, act, res=rd, cin=cin, impl=impl)[..., :cout].cpu() for impl in (1, 2, 0)}
    torch.cuda.synchronize()
    assert torch.equal(got[2], got[1]), (got[2] - got[1]).abs().max()
    assert torch.equal(got[0], got[2])  # the routing picks the MFMA kernel for these shapes
    rng = float(y.abs().max())
    assert float((nchw(got[2]) - y).abs().max()) <= 1e-4 * max(rng, 1.0)


def test_conv32_routing_falls_back_to_scalar():
    """Shapes the MFMA kernel does not take (Cout % 4 != 0, Cin % 8 != 0) run on the scalar kernel through the same entry point."""
    g = torch.Generator().manual_seed(5)
    for cin, cout in ((16, 5), (12, 16)):
        x = torch.randn(2, cin, 9, 9, generator=g)
        w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
        b = torch.randn(cout, generator=g)
        y = F.silu(F.conv2d(x, w, b, 1, 1))
        got = O.conv2d_nhwc_f32(nhwc(x).to(DEV), w, b, 3, 1, True)[..., :cout].cpu()
        assert float((nchw(got) - y).abs().max()) <= 1e-4 * max(float(y.abs().max()), 1.0)
        with pytest.raises(L.BsyError):
            O.conv2d_nhwc_f32(nhwc(x).to(DEV), w, b, 3, 1, True, impl=2)


CONV32X_CASES = CONV32_CASES + [
    (4, 40, 40, 256, 256, 3, 2, True, False, 0),     # 256 x 256 tile (two pixel tiles, K = 2304), stride 2
    (2, 80, 80, 192, 256, 1, 1, True, True, 0),      # 256 x 256 tile, 1x1, shortcut, 50 pixel tiles
    (3, 47, 33, 128, 136, 3, 1, True, True, 8),      # 256 x 128 tile: ragged last pixel tile, couts past the second tile's range
    (2, 64, 64, 64, 128, 1, 1, False, False, 0),     # 256 x 128 tile, thin K = 64 (two K-steps)
    (2, 80, 80, 128, 64, 3, 1, True, False, 0),      # patch kernel: Detect cv2.0.0's shape (four 32-channel chunks, 64 couts)
    (3, 40, 40, 64, 160, 3, 1, True, True, 0),       # patch kernel: 128-cout tiles, the second one ragged; 40 x 40 = 5 x 3 tiles, last column half outside
    (2, 37, 21, 48, 16, 3, 1, True, False, 8),       # patch kernel: 16-channel chunks (Cin = 48), thin tile, ragged map, padded row stride
    (2, 160, 160, 16, 32, 3, 1, True, True, 0),      # patch kernel: ONE 16-channel chunk (model.2.m.0.cv2 of YOLO11s)
    (3, 20, 20, 128, 128, 3, 1, True, True, 0),      # patch kernel, 6 x 20 tiles (W % 20 == 0): 4 tile rows, the last with 2 of 6 rows inside
    (2, 17, 40, 64, 48, 3, 1, True, False, 0),       # 6 x 20 tiles on a 40-wide map, ragged height, ragged couts
    (2, 64, 96, 32, 64, 3, 2, True, False, 0),       # stride 2, 64 couts: model.1's shape class
    (2, 80, 80, 128, 128, 3, 2, True, False, 0),     # stride 2, 128 couts: model.3 / model.17
    (3, 37, 21, 48, 96, 3, 2, True, True, 8),        # stride 2 on an odd map (19 x 11 outputs), ragged couts, shortcut, padded rows
    (1, 16, 16, 16, 40, 3, 2, False, False, 0),      # stride 2, one 16-channel K piece, no activation, 40 couts
]


@pytest.mark.parametrize("case", CONV32X_CASES)
def test_conv32x_matches_fp64_reference(case, monkeypatch):
    """fp32x mode's conv (conv32x_mfma.hip: fp32 storage, operands split into f16 pairs, three fp16 MFMAs per product) against an
    fp64 torch conv: 2e-6 of the output range -- the exact fp32 kernel's own distance to fp64 on these cases is 1-2e-6 (one rounding
    per product), the split-f16 form measured 0.4-1.6e-6 -- on every tile the launcher can pick (128-pixel tiles, 256 x 128, 256 x 256;
    K-steps of 16 and 32).  Activations include tiny values (lo parts in f16's subnormal range) and channels outside the view hold
    NaNs."""
    B, H, W, cin, cout, k, s, act, use_res, ldx = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, cin, H, W, generator=g)
    x[:, ::3] *= 1e-3
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.5
    y = F.conv2d(x.double(), w.double(), b.double(), s, k // 2)
    if act:
        y = F.silu(y)
    res = torch.randn(y.shape, generator=g) if use_res else None
    if use_res:
        y = y + res.double()
    xd = torch.zeros(B, H, W, cin + ldx)
    xd[..., :cin] = nhwc(x)
    xd[..., cin:] = float("nan")
    xd = xd.to(DEV)
    rd = nhwc(res).to(DEV) if use_res else None
    rng = max(float(y.abs().max()), 1.0)
    for env in ({}, {"BSY_CONV32X_PATCH": "1"}, {"BSY_CONV32X_PATCH": "0"}, {"BSY_CONV32X_TILE": "0"}, {"BSY_CONV32X_TILE": "1", "BSY_CONV32X_BK": "16"},
                {"BSY_CONV32X_TILE": "1"}, {"BSY_CONV32X_TILE": "2", "BSY_CONV32X_BK": "16"}, {"BSY_CONV32X_TILE": "2"}):
        for kname in ("BSY_CONV32X_PATCH", "BSY_CONV32X_TILE", "BSY_CONV32X_BK"):
            monkeypatch.delenv(kname, raising=False)
        for kname, v in env.items():
            monkeypatch.setenv(kname, v)
        got = O.conv2d_nhwc_f32x(xd, w, b, k, s, act, res=rd, cin=cin)[..., :cout].cpu()
        err = float((nchw(got).double() - y).abs().max())
        assert err <= 2e-6 * rng, (env, err, rng)


def test_conv32x_rejects_what_it_does_not_take():
    """Shapes outside the kernel's rules (Cout % 4, Cin % 8) are BSY_ERR_ARG on the stand-alone entry point; the engine routes them
    to the exact kernels (test_engine_on_custom_width_multiples runs such graphs in this mode)."""
    g = torch.Generator().manual_seed(5)
    for cin, cout in ((16, 6), (12, 16)):
        x = torch.randn(2, 9, 9, cin, generator=g).to(DEV)
        w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
        with pytest.raises(L.BsyError):
            O.conv2d_nhwc_f32x(x, w, torch.zeros(cout), 3, 1, True)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
def test_conv32_first_mfma_equals_scalar(dt):
    """fp32 mode's image conv on the fp32 MFMA kernel (taps widened to 8 k values, 5 of them zero) = the scalar kernel, bit for bit."""
    import ctypes as C
    g = torch.Generator().manual_seed(3)
    B, H, W, cout = 3, 46, 38, 16
    x = torch.rand(B, 3, H, W, generator=g).to(dt)
    w = torch.randn(cout, 3, 3, 3, generator=g) * 0.3
    b = torch.randn(cout, generator=g) * 0.5
    y = F.silu(F.conv2d(x.float(), w, b, 2, 1))
    xd = x.to(DEV).contiguous()
    wk = w.permute(2, 3, 1, 0).reshape(27, cout).contiguous().to(DEV)
    bk = b.to(DEV)
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    got = {}
    for impl in (1, 2, 0):
        out = torch.zeros(B, OH, OW, cout, device=DEV)
        L.check(L.lib.bsy_conv_first_f32(O._p(xd), L.dtype_code(dt), B, H, W, O._p(wk), O._p(bk), O._p(out), cout, cout, 3, 2, 1, impl,
                                         O._stream(xd)))
        got[impl] = out.cpu()
    assert torch.equal(got[2], got[1]) and torch.equal(got[0], got[2])
    assert float((nchw(got[2]) - y).abs().max()) <= 1e-4 * max(float(y.abs().max()), 1.0)


@pytest.mark.parametrize("N", [100, 400, 1600])
def test_attention32_tiled_equals_generic(N):
    """fp32 mode's attention: the LDS-tiled kernel (key_dim 32, head_dim 64) = the generic one bit for bit, and both = softmax
    attention in torch fp32 to 1e-5."""
    g = torch.Generator().manual_seed(N)
    B, heads, kd, hd = 2, 2, 32, 64
    ld = heads * (2 * kd + hd)
    qkv = torch.randn(B, N, ld, generator=g)
    q = qkv[..., :heads * kd].view(B, N, heads, kd).permute(0, 2, 1, 3)
    k = qkv[..., heads * kd:2 * heads * kd].view(B, N, heads, kd).permute(0, 2, 1, 3)
    v = qkv[..., 2 * heads * kd:].view(B, N, heads, hd).permute(0, 2, 1, 3)
    ref = (torch.softmax((q @ k.transpose(-1, -2)) * kd ** -0.5, -1) @ v).permute(0, 2, 1, 3).reshape(B, N, heads * hd)
    qd = qkv.to(DEV)
    got = {}
    for impl in (1, 2, 0):
        out = torch.zeros(B, N, heads * hd, device=DEV)
        L.check(L.lib.bsy_attention_f32(O._p(qd), ld, B, N, heads, kd, hd, kd ** -0.5, O._p(out), heads * hd, impl, O._stream(qd)))
        got[impl] = out.cpu()
    assert torch.equal(got[2], got[1]) and torch.equal(got[0], got[2])
    assert float((got[2] - ref).abs().max()) <= 1e-5 * max(float(ref.abs().max()), 1.0)


@pytest.mark.parametrize("N", [100, 400, 1600, 37])
def test_attention32x_matches_fp64_softmax_attention(N):
    """fp32x mode's attention (attention32x.hip: flash-style on the fp16 matrix pipe, q / k / v / p as f16 pairs) against softmax
    attention in torch fp64: 3e-6 of the output range (the exact kernels above hold 1e-5 against torch fp32), on ragged key counts
    (100 = 3 tiles + 4 keys, 37) and with large logits (q scaled so that the softmax is peaky: the online rescale path runs)."""
    g = torch.Generator().manual_seed(N)
    B, heads, kd, hd = 2, 2, 32, 64
    ld = heads * (2 * kd + hd)
    qkv = torch.randn(B, N, ld, generator=g)
    qkv[..., :heads * kd] *= 3.0
    q = qkv[..., :heads * kd].view(B, N, heads, kd).permute(0, 2, 1, 3).double()
    k = qkv[..., heads * kd:2 * heads * kd].view(B, N, heads, kd).permute(0, 2, 1, 3).double()
    v = qkv[..., 2 * heads * kd:].view(B, N, heads, hd).permute(0, 2, 1, 3).double()
    ref = (torch.softmax((q @ k.transpose(-1, -2)) * kd ** -0.5, -1) @ v).permute(0, 2, 1, 3).reshape(B, N, heads * hd)
    qd = qkv.to(DEV)
    out = torch.zeros(B, N, heads * hd, device=DEV)
    L.check(L.lib.bsy_attention_f32(O._p(qd), ld, B, N, heads, kd, hd, kd ** -0.5, O._p(out), heads * hd, 3, O._stream(qd)))
    err = float((out.cpu().double() - ref).abs().max())
    assert err <= 3e-6 * max(float(ref.abs().max()), 1.0), err


PATCH_CASES = [
    # B, H, W, cin, cout, cfg, res      cfg = tile << 4 | variant: 161 = patch kernel TN 128, 177 = TN 64
    (2, 24, 40, 64, 64, 177, False),      # 40-wide map: 2.5 tiles per row
    (1, 17, 21, 32, 128, 161, False),     # ragged on both axes, one chunk
    (2, 16, 16, 128, 72, 161, True),      # ragged cout inside a 128-wide tile + residual
    (1, 80, 80, 128, 64, 177, False),     # Detect cv2[0][0] of YOLO11s
    (3, 20, 20, 256, 256, 161, True),     # 8 chunks, 2 cout tiles, residual
    (2, 33, 31, 64, 40, 177, False),
    (2, 24, 40, 96, 64, 178, True),       # 2-stage weight ring, 3 chunks
    (1, 33, 47, 128, 136, 161, False),    # ragged everywhere, 2 cout tiles of 128
    # 6 x 20-pixel tiles (tiles 12 / 13; 20 x 20 and 40 x 40 maps: fewer tiles than 8 x 16), exact and ragged extents
    (3, 20, 20, 128, 128, 193, True),
    (2, 20, 20, 64, 64, 209, False),
    (2, 40, 40, 64, 128, 210, True),
    (1, 13, 27, 32, 24, 210, False),      # ragged: last tile row has 1 of 6 rows, last tile column 7 of 20 pixels
    (5, 6, 20, 96, 64, 209, False),       # exactly one tile per image
]


@pytest.mark.parametrize("case", PATCH_CASES)
def test_conv_patch_kernel_matches_oracle(case, monkeypatch):
    """3x3 stride-1 patch kernel (conv_mfma.hip, tiles 10/11 = 8x16-pixel tiles, 12/13 = 6x20) against the fp32 reference and against the implicit-GEMM
    kernel, which walks K in the same order for these layers (conv_korder: chunk-major over 32-channel chunks): bit for bit."""
    B, H, W, cin, cout, cfg, use_res = case
    g = torch.Generator().manual_seed(cfg * 1000 + cin)
    x = h16(torch.randn(B, cin, H, W, generator=g))
    w = h16(torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.5
    y = F.silu(F.conv2d(x, w, b, 1, 1))
    res = h16(torch.randn(y.shape, generator=g)) if use_res else None
    if use_res:
        y = y + res
    wp, bp = O.pack_conv_weight(w, b, DEV)
    xd = nhwc(x).half().to(DEV)
    rd = nhwc(res).half().to(DEV) if use_res else None
    monkeypatch.setenv("BSY_CONV_CFG", str(0x35))  # 128 x 64 implicit-GEMM tile, BK 32, the layer's chunk-major walk
    base = O.conv2d_nhwc(xd, wp, bp, cout, 3, 1, True, res=rd)
    monkeypatch.setenv("BSY_CONV_CFG", str(cfg))
    out = O.conv2d_nhwc(xd, wp, bp, cout, 3, 1, True, res=rd)
    torch.cuda.synchronize()
    monkeypatch.delenv("BSY_CONV_CFG")
    np.testing.assert_allclose(nchw(out.float().cpu()).numpy(), y.numpy(), rtol=2e-3, atol=2e-3)
    assert torch.equal(out, base)


PERSIST_CASES = [
    # B, H, W, cin, cout, cfg      tiles 8 / 9 = persistent 1x1 kernel (128 x 128, 128 x 64 tiles), variants 1-3
    (70, 40, 40, 64, 128, 0x81), (70, 40, 40, 64, 128, 0x82), (9, 80, 80, 128, 128, 0x83),
    (70, 40, 40, 96, 64, 0x91), (33, 37, 41, 32, 72, 0x92), (9, 80, 80, 192, 64, 0x93),
    (17, 33, 47, 96, 512, 0x82),                                                             # 4 cout tiles per pixel tile
    # tiles 14 / 15 = weights-resident streaming 1x1 kernel (128 x 128 / 128 x 64 tiles; variant 1: 4-stage pixel ring, 2: 3 stages)
    (70, 40, 40, 128, 128, 0xe1), (70, 40, 40, 192, 256, 0xe2), (33, 37, 41, 96, 72, 0xe1), (9, 80, 80, 256, 256, 0xf1),           # (K = 256 on the 128-cout tile would need 32 fragments: not built)
    (70, 40, 40, 512, 128, 0xf1), (33, 37, 41, 128, 72, 0xf2), (3, 7, 5, 128, 64, 0xf1), (300, 16, 16, 160, 136, 0xe2),
]


@pytest.mark.parametrize("case", PERSIST_CASES)
def test_conv_persistent_1x1_matches_one_tile_per_workgroup(case, monkeypatch):
    """conv1x1_persist_kernel (tiles walked by persistent workgroups, dedicated store waves) and conv1x1_wres_kernel (weights in
    registers, pixels streamed, stores counted in the DMA waits) against the one-tile-per-workgroup implicit-GEMM kernel -- same K
    walk, so bit for bit -- and the fp32 reference; more tiles than workgroups, ragged last tiles, fewer tiles than XCDs."""
    B, H, W, cin, cout, cfg = case
    g = torch.Generator().manual_seed(cfg * 7 + cin)
    x = h16(torch.randn(B, cin, H, W, generator=g))
    w = h16(torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.5
    wp, bp = O.pack_conv_weight(w, b, DEV)
    xd = nhwc(x).half().to(DEV)
    monkeypatch.setenv("BSY_CONV_CFG", str(0x21))
    base = O.conv2d_nhwc(xd, wp, bp, cout, 1, 1, True)
    monkeypatch.setenv("BSY_CONV_CFG", str(cfg))
    out = O.conv2d_nhwc(xd, wp, bp, cout, 1, 1, True)
    torch.cuda.synchronize()
    monkeypatch.delenv("BSY_CONV_CFG")
    assert torch.equal(out, base)
    y = F.silu(F.conv2d(x[:2], w, b))
    np.testing.assert_allclose(nchw(out[:2].float().cpu()).numpy(), y.numpy(), rtol=2e-3, atol=2e-3)


KORDER_CASES = [
    # B, H, W, cin, cout, k, stride, K walk (conv_mfma.hip conv_korder: 0 packed order, 1 / 2 chunk-major over 32- / 64-channel chunks)
    (2, 40, 40, 128, 128, 3, 2, 2),       # model.3's shape class
    (1, 33, 31, 64, 256, 3, 2, 2),        # odd extents; the 256 x 256 tile applies
    (2, 20, 20, 256, 256, 3, 2, 2),       # 4 chunks x 9 taps
    (3, 17, 23, 32, 48, 3, 2, 1),         # one 32-channel chunk, ragged cout
    (2, 24, 24, 96, 64, 3, 1, 1),         # stride 1: the patch kernel's order, three chunks
    (2, 24, 40, 128, 128, 3, 1, 1),       # stride 1, Cin % 64 == 0: still 32-channel chunks (no 64-deep K-steps)
    (2, 16, 16, 96, 128, 1, 1, 0),        # 1x1: one tap, one order
    (2, 12, 10, 24, 32, 3, 1, 0),         # unaligned channel count: generic variant only
]


@pytest.mark.parametrize("case", KORDER_CASES)
def test_conv_every_configuration_of_a_layer_is_bit_identical(case, monkeypatch):
    """The K walk of a conv is a function of its shape (conv_mfma.hip conv_korder), not of the kernel configuration: every id the
    library accepts for a layer -- implicit-GEMM tiles with BK 32 / 64 and 2 / 3 stages, patch tiles, persistent tiles -- returns
    the same bits (round-2 VERDICT: results must not depend on what the autotuner timed), and ids of another walk (recorded by an
    older tune cache) are rejected."""
    B, H, W, cin, cout, k, s, ko = case
    g = torch.Generator().manual_seed(1000 * cin + cout + k)
    x = h16(torch.randn(B, cin, H, W, generator=g))
    w = h16(torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.5
    y = F.silu(F.conv2d(x, w, b, s, k // 2))
    wp, bp = O.pack_conv_weight(w, b, DEV)
    xd = nhwc(x).half().to(DEV)
    monkeypatch.delenv("BSY_CONV_CFG", raising=False)
    base = O.conv2d_nhwc(xd, wp, bp, cout, k, s, True)  # the heuristic configuration
    np.testing.assert_allclose(nchw(base.float().cpu()).numpy(), y.numpy(), rtol=2e-3, atol=2e-3)
    accepted = []
    for tile in range(16):
        for var in range(8):
            cfg = (tile << 4) | var
            monkeypatch.setenv("BSY_CONV_CFG", str(cfg))
            try:
                out = O.conv2d_nhwc(xd, wp, bp, cout, k, s, True)
            except L.BsyError:
                continue
            accepted.append(cfg)
            assert torch.equal(out, base), hex(cfg)
    monkeypatch.delenv("BSY_CONV_CFG")
    torch.cuda.synchronize()
    assert len(accepted) >= (1 if ko == 0 and k == 3 else 4), [hex(c) for c in accepted]
    for cfg in accepted:
        tile, var = cfg >> 4, cfg & 7
        if tile < 8:  # implicit-GEMM ids carry the walk: chunk-major bit set iff the layer is chunk-major; no 64-deep steps on 32-channel chunks
            assert bool(var & 4) == (ko != 0) and not (ko == 1 and (var & 3) == 3), hex(cfg)
        else:
            assert not (var & 4), hex(cfg)
    if ko == 1 and s == 1:
        assert any(c >> 4 >= 10 for c in accepted) and any(c >> 4 < 8 for c in accepted)  # patch AND implicit-GEMM tiles took part
    if ko == 2:
        assert any((c & 3) == 3 for c in accepted) and any((c & 3) in (1, 2) for c in accepted)  # 64-deep and 32-deep K-steps


def test_conv_two_sources_and_upsample():
    """Virtual Concat + virtual nn.Upsample: cv1(cat(upsample(a), b)) (yolo11 head layers 11-13)."""
    g = torch.Generator().manual_seed(3)
    a = h16(torch.randn(2, 64, 5, 6, generator=g))
    b = h16(torch.randn(2, 32, 10, 12, generator=g))
    w = h16(torch.randn(48, 96, 1, 1, generator=g) * 0.15)
    bias = torch.randn(48, generator=g) * 0.1
    ref = F.silu(F.conv2d(torch.cat((F.interpolate(a, scale_factor=2.0, mode="nearest"), b), 1), w, bias))
    # through the engine's op interface: a tiny hand-built plan
    import ctypes as C
    from bs_yolo_amd import lib as L
    wp, bp = O.pack_conv_weight(w, bias, "cpu")
    blob = wp.numpy().tobytes()
    boff = (len(blob) + 255) // 256 * 256
    blob = blob + b"\0" * (boff - len(blob)) + bp.numpy().tobytes()
    eng = C.c_void_p()
    L.check(L.lib.bsy_engine_create(0, C.byref(eng)))
    L.check(L.lib.bsy_engine_load_weights(eng, (C.c_char * len(blob)).from_buffer_copy(blob), len(blob)))
    op = L.Op()
    op.kind = L.OP_CONV
    op.B, op.H, op.W, op.OH, op.OW = 2, 10, 12, 10, 12
    op.src0, op.src1 = L.View(L.BSY_EXT_BASE + 0, 64, 0, 64), L.View(L.BSY_EXT_BASE + 1, 32, 0, 32)
    op.up0, op.up1 = 1, 0
    op.dst, op.res = L.View(L.BSY_EXT_BASE + 2, 48, 0, 48), L.View(*L.NO_VIEW)
    op.ksize, op.stride, op.pad, op.act, op.out_f32, op.dst_scale = 1, 1, 0, 1, 0, 1
    op.w_off, op.b_off = 0, boff
    plan = C.c_void_p()
    L.check(L.lib.bsy_plan_create(eng, (L.Op * 1)(op), 1, None, 0, C.byref(plan)))
    ta, tb = nhwc(a).half().to(DEV), nhwc(b).half().to(DEV)
    out = torch.zeros(2, 10, 12, 48, dtype=torch.float16, device=DEV)
    ext = (C.c_void_p * 3)(ta.data_ptr(), tb.data_ptr(), out.data_ptr())
    L.check(L.lib.bsy_plan_run(plan, ext, 3, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    L.lib.bsy_plan_destroy(plan)
    L.lib.bsy_engine_destroy(eng)
    np.testing.assert_allclose(nchw(out.float().cpu()).numpy(), ref.numpy(), rtol=2e-3, atol=2e-3)


def test_conv_rejects_bad_arguments():
    x = torch.zeros(1, 4, 4, 12, dtype=torch.float16, device=DEV)  # 12 channels: not a multiple of 8
    wp, bp = O.pack_conv_weight(torch.zeros(8, 12, 1, 1), torch.zeros(8), DEV)
    with pytest.raises(L.BsyError, match="multiples of 8"):
        O.conv2d_nhwc(x, wp, bp, 8, 1)
    x = torch.zeros(1, 4, 4, 16, dtype=torch.float16, device=DEV)
    wp, bp = O.pack_conv_weight(torch.zeros(8, 16, 5, 5)[:, :, :1, :1], torch.zeros(8), DEV)
    with pytest.raises(L.BsyError, match="ksize"):
        O.conv2d_nhwc(x, wp, bp, 8, 5)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_conv_first_matches_oracle(dtype):
    g = torch.Generator().manual_seed(11)
    img = torch.rand(2, 3, 70, 200, generator=g)  # odd tile tails: OH = 35 (4-row tiles), OW = 100 (64-col tiles)
    w = h16(torch.randn(32, 3, 3, 3, generator=g) * 0.3)
    b = torch.randn(32, generator=g) * 0.2
    ref = F.silu(F.conv2d(h16(img), w, b, 2, 1))  # the kernel parks the image in LDS as fp16
    out = O.conv_first(img.to(dtype).to(DEV), w, b)
    torch.cuda.synchronize()
    np.testing.assert_allclose(nchw(out.float().cpu()).numpy(), ref.numpy(), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("shape,c0,c1", [((2, 3, 64, 64), 32, 64), ((1, 3, 96, 160), 32, 64), ((3, 3, 38, 52), 32, 64),
                                         ((2, 3, 70, 36), 16, 32), ((70, 3, 32, 32), 32, 64)])
def test_stem_fused_matches_unfused_and_oracle(dtype, shape, c0, c1):
    """Layers 0 + 1 as one launch (csrc/stem_fused.hip): bit-identical to bsy_conv_first + bsy_conv2d, and equal to
    the fp32 reference of the two Conv.forward_fuse calls (conv.py:149-151) with the layer-0 map rounded to fp16.
    Shapes cover ragged tiles (layer-1 maps 9 x 13, 17 x 9), heights that are not multiples of 4, and more tiles
    than persistent workgroups (70 images)."""
    g = torch.Generator().manual_seed(12)
    img = torch.rand(*shape, generator=g)
    w0 = h16(torch.randn(c0, 3, 3, 3, generator=g) * 0.3)
    b0 = torch.randn(c0, generator=g) * 0.2
    w1 = h16(torch.randn(c1, c0, 3, 3, generator=g) * (2.0 / (9 * c0)) ** 0.5)
    b1 = torch.randn(c1, generator=g) * 0.2
    x = img.to(dtype).to(DEV)
    fused = O.stem_fused(x, w0, b0, w1, b1)
    mid = O.conv_first(x, w0, b0)
    w1p, b1p = O.pack_conv_weight(w1, b1, DEV)
    two = O.conv2d_nhwc(mid, w1p, b1p, c1, 3, 2, True)
    torch.cuda.synchronize()
    assert fused.shape == two.shape
    assert torch.equal(fused, two)
    ref = F.silu(F.conv2d(h16(F.silu(F.conv2d(h16(img), w0, b0, 2, 1))), w1, b1, 2, 1))
    np.testing.assert_allclose(nchw(fused.float().cpu()).numpy(), ref.numpy(), rtol=3e-3, atol=3e-3)


def test_stem_fused_rejects_unsupported():
    assert L.lib.bsy_stem_fused_supported(32, 64, 640, 640) == 1
    assert L.lib.bsy_stem_fused_supported(64, 128, 640, 640) == 0  # yolo11m/l/x: layer-1 weights do not fit in registers
    assert L.lib.bsy_stem_fused_supported(32, 64, 64, 66) == 0     # W % 4
    img = torch.rand(1, 3, 64, 66, device=DEV).half()
    with pytest.raises(L.BsyError):
        O.stem_fused(img, torch.zeros(32, 3, 3, 3), torch.zeros(32), torch.zeros(64, 32, 3, 3), torch.zeros(64))


@pytest.mark.parametrize("widths", [(32, 16), (64, 32)])
@pytest.mark.parametrize("shape", [(2, 16, 32), (1, 40, 24), (3, 9, 13), (70, 8, 16), (1, 3, 5), (300, 16, 16)])
def test_bottleneck_fused_matches_unfused_and_oracle(shape, widths, monkeypatch):
    """Bottleneck (block.py:3405-3419, k = (3,3), shortcut) as one launch (csrc/bneck_fused.hip; 32/16: YOLO11s model.2,
    64/32: model.4 / model.16): bit-identical to two bsy_conv2d launches (the second with the residual) and equal to the fp32 reference with the hidden map rounded to fp16.  Input and output are
    channel slices of one wider buffer, as inside C3k2's concat buffer; shapes cover ragged tiles, maps smaller than one tile
    and more tiles than persistent workgroups."""
    B, H, W = shape
    c, ch = widths
    ld = 3 * c
    g = torch.Generator().manual_seed(21)
    buf = h16(torch.randn(B, H, W, ld, generator=g))
    w1 = h16(torch.randn(ch, c, 3, 3, generator=g) * (2.0 / (9 * c)) ** 0.5)
    b1 = torch.randn(ch, generator=g) * 0.2
    w2 = h16(torch.randn(c, ch, 3, 3, generator=g) * (2.0 / (9 * ch)) ** 0.5)
    b2 = torch.randn(c, generator=g) * 0.2
    dbuf = buf.half().to(DEV)
    x = dbuf[..., c:2 * c]
    O.bottleneck_fused(x, w1, b1, w2, b2, out=dbuf[..., 2 * c:])
    w1p, b1p = O.pack_conv_weight(w1, b1, DEV)
    w2p, b2p = O.pack_conv_weight(w2, b2, DEV)
    xc = x.contiguous()
    monkeypatch.delenv("BSY_CONV_CFG", raising=False)  # any configuration: a layer's K walk depends on its shape only (conv_korder)
    mid = O.conv2d_nhwc(xc, w1p, b1p, ch, 3, 1, True)
    two = O.conv2d_nhwc(mid, w2p, b2p, c, 3, 1, True, res=xc)
    torch.cuda.synchronize()
    assert torch.equal(dbuf[..., 2 * c:], two)
    assert torch.equal(dbuf[..., :2 * c].cpu(), buf[..., :2 * c].half())  # the other slices are untouched
    xr = nchw(buf[..., c:2 * c])
    ref = xr + F.silu(F.conv2d(h16(F.silu(F.conv2d(xr, w1, b1, 1, 1))), w2, b2, 1, 1))
    np.testing.assert_allclose(nchw(two.float().cpu()).numpy(), ref.numpy(), rtol=4e-3, atol=4e-3)


def test_bottleneck_fused_rejects_unsupported():
    assert L.lib.bsy_bottleneck_fused_supported(32, 16) == 1 and L.lib.bsy_bottleneck_fused_supported(64, 32) == 1
    assert L.lib.bsy_bottleneck_fused_supported(64, 64) == 0 and L.lib.bsy_bottleneck_fused_supported(128, 64) == 0
    x = torch.zeros(1, 8, 8, 128, dtype=torch.float16, device=DEV)
    with pytest.raises(L.BsyError):
        O.bottleneck_fused(x, torch.zeros(64, 128, 3, 3), torch.zeros(64), torch.zeros(128, 64, 3, 3), torch.zeros(128))


@pytest.mark.parametrize("shape", [(2, 16, 32), (1, 40, 24), (3, 9, 13), (300, 8, 16), (1, 3, 5), (2, 33, 47)])
def test_c3k2_fused_matches_unfused_and_oracle(shape):
    """The whole C3k2 block (c3k = False, one Bottleneck; block.py:3796-3804) as ONE launch (csrc/c3k2_fused.hip): bit-identical
    to the three-launch path it replaces (cv1 conv -> fused Bottleneck on the y1 slice of the concat buffer -> cv2 conv) and
    equal to the fp32 reference with every intermediate map rounded to fp16.  Shapes: ragged tiles, maps smaller than one
    tile, more tiles than persistent workgroups; the input is a channel slice of a wider buffer."""
    B, H, W = shape
    cin, c, c2 = 64, 32, 128
    g = torch.Generator().manual_seed(33)

    def wt(co, ci, k):
        return h16(torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5), torch.randn(co, generator=g) * 0.2
    (w1, b1), (wa, ba), (wb, bb), (w4, b4) = wt(2 * c, cin, 1), wt(c // 2, c, 3), wt(c, c // 2, 3), wt(c2, 3 * c, 1)
    buf = h16(torch.randn(B, H, W, cin + 16, generator=g))
    dbuf = buf.half().to(DEV)
    x = dbuf[..., 8:8 + cin]
    out = O.c3k2_fused(x, w1, b1, wa, ba, wb, bb, w4, b4)
    # the three-launch path through the same C ABI
    p1, pb1 = O.pack_conv_weight(w1, b1, DEV)
    p4, pb4 = O.pack_conv_weight(w4, b4, DEV)
    cat = torch.zeros(B, H, W, 3 * c, dtype=torch.float16, device=DEV)
    cat[..., :2 * c] = O.conv2d_nhwc(x.contiguous(), p1, pb1, 2 * c, 1, 1, True)
    O.bottleneck_fused(cat[..., c:2 * c], wa, ba, wb, bb, out=cat[..., 2 * c:])
    three = O.conv2d_nhwc(cat, p4, pb4, c2, 1, 1, True)
    torch.cuda.synchronize()
    assert torch.equal(out, three)
    assert torch.equal(dbuf.cpu(), buf.half())  # the input buffer is untouched
    xr = nchw(buf[..., 8:8 + cin])
    y01 = h16(F.silu(F.conv2d(xr, w1, b1)))
    y1 = y01[:, c:]
    y2 = h16(y1 + h16(F.silu(F.conv2d(h16(F.silu(F.conv2d(y1, wa, ba, 1, 1))), wb, bb, 1, 1))))
    ref = F.silu(F.conv2d(torch.cat([y01, y2], 1), w4, b4))
    np.testing.assert_allclose(nchw(out.float().cpu()).numpy(), ref.numpy(), rtol=4e-3, atol=4e-3)


def test_c3k2_fused_rejects_unsupported():
    assert L.lib.bsy_c3k2_fused_supported(64, 32, 128) == 1 and L.lib.bsy_c3k2_fused_supported(128, 64, 256) == 0
    x = torch.zeros(1, 8, 8, 128, dtype=torch.float16, device=DEV)
    with pytest.raises(L.BsyError):
        O.c3k2_fused(x, torch.zeros(128, 128, 1, 1), torch.zeros(128), torch.zeros(32, 64, 3, 3), torch.zeros(32),
                     torch.zeros(64, 32, 3, 3), torch.zeros(64), torch.zeros(256, 192, 1, 1), torch.zeros(256))


def test_engine_c3k2_fusion_is_bit_identical():
    """Engine level: YOLO11s with the C3k2 block of model.2 as one launch vs the three-launch plan (fuse_tail=False), and
    YOLO11n (model.4 has the same widths)."""
    for scale, shape in (("s", (3, 96, 160)), ("n", (2, 128, 64))):
        m = R.Model("yolo11", scale, 80, "detect")
        P = R.synth_params(m, 2)
        cfg = stock_cfg("yolo11", scale)
        fused = YoloEngine(cfg, P, autotune=False, fuse_tail=True)
        plain = YoloEngine(cfg, P, autotune=False, fuse_tail=False)
        B, H, W = shape
        pf, _ = fused.plan_for(B, H, W, torch.float16, torch.float16)
        pp, _ = plain.plan_for(B, H, W, torch.float16, torch.float16)
        assert sum(o["kind"] == L.OP_C3K2 for o in pf.ops) == 1 and len(pp.ops) - len(pf.ops) == 2
        x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(9)).half().to(DEV)
        yf, rf = fused(x)
        yp, rp = plain(x)
        assert torch.equal(yf, yp) and all(torch.equal(a, b) for a, b in zip(rf, rp))
        fused.close()
        plain.close()


@pytest.mark.parametrize("kh,kw,stride,act", [(5, 5, 1, True), (7, 7, 1, True), (3, 3, 2, False), (1, 21, 1, False),
                                              (21, 1, 1, False), (1, 5, 1, False), (11, 1, 1, False)])
def test_dwconv_generic_matches_oracle(kh, kw, stride, act):
    """Depthwise kh x kw conv (PMSFA 5x5 / 7x7, block.py:3040-3042; SCDown.cv2, block.py:4529; MSCAAttention strips,
    MSCA.py:26-39) on maps smaller than the kernel and with odd extents."""
    g = torch.Generator().manual_seed(31)
    x = h16(torch.randn(2, 24, 9, 13, generator=g))
    w = torch.randn(24, 1, kh, kw, generator=g) * (1.0 / (kh * kw)) ** 0.5
    b = torch.randn(24, generator=g) * 0.2
    ref = F.conv2d(x, w, b, stride, (kh // 2, kw // 2), 1, 24)
    if act:
        ref = F.silu(ref)
    out = O.dwconv_nhwc(nhwc(x).half().to(DEV), w, b, stride, act)
    torch.cuda.synchronize()
    np.testing.assert_allclose(nchw(out.float().cpu()).numpy(), ref.numpy(), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("k,shape,c", [(5, (2, 37, 45), 16), (7, (2, 37, 45), 16), (5, (1, 20, 20), 64), (7, (3, 8, 100), 32), (7, (1, 40, 40), 8),
                                       (5, (2, 9, 13), 24), (7, (1, 160, 160), 40)])
def test_dwconv_tiled_equals_window_kernel_and_oracle(k, shape, c, monkeypatch):
    """LDS-tiled depthwise 5 x 5 / 7 x 7 (dwconv_tile_kernel: PMSFA's convs, block.py:3035-3054) against the window kernel -- the same
    FMAs in the same order, so bit for bit -- and the fp32 reference: 1 / 2 / 4 chunks per workgroup, a narrower last channel group
    (24, 40 channels), ragged tile rows and columns, maps smaller than a tile."""
    B, H, W = shape
    g = torch.Generator().manual_seed(100 * k + c)
    x = h16(torch.randn(B, c, H, W, generator=g))
    w = torch.randn(c, 1, k, k, generator=g) * (1.0 / (k * k)) ** 0.5
    b = torch.randn(c, generator=g) * 0.2
    ref = F.silu(F.conv2d(x, w, b, 1, k // 2, 1, c))
    xd = nhwc(x).half().to(DEV)
    out = O.dwconv_nhwc(xd, w, b, 1, True)
    monkeypatch.setenv("BSY_NO_DWTILE", "1")
    base = O.dwconv_nhwc(xd, w, b, 1, True)
    torch.cuda.synchronize()
    assert torch.equal(out, base)
    np.testing.assert_allclose(nchw(out.float().cpu()).numpy(), ref.numpy(), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("tag", ["ela64", "ela256", "ela40", "ela24"])  # 40 / 24: GroupNorm groups of 20 / 24 channels
def test_ela_matches_reference_golden(tag):
    """ELA (nn/Addmodules/ELA.py:33-101) through bsy_ela against the fork's own module output (modules_bsyolo.npz)."""
    z = np.load(GOLDEN / "modules_bsyolo.npz", allow_pickle=False)
    cases = json.loads(str(z["cases"]))
    mod = R.ELA("m", cases[tag][1])
    P = {n: R.synth_param(n, s, 9) for n, s in mod.specs()}
    x = torch.from_numpy(z[tag + ".x"])
    coef = [torch.sigmoid(P[f"m.{k}"]).item() for k in ("ch_weight", "sp_weight", "res_weight")]
    out = O.ela_nhwc(nhwc(h16(x)).half().to(DEV), P["m.spatial_conv.weight"], P["m.ch_att.2.weight"], P["m.gn.weight"],
                     P["m.gn.bias"], coef)
    torch.cuda.synchronize()
    got = nchw(out.float().cpu()).numpy()
    ref = z[tag + ".y"]
    assert np.abs(got - ref).max() < 4e-3 * max(1.0, np.abs(ref).max())  # fp16 input / output rounding only


@pytest.mark.parametrize("shape,c,c2", [((2, 24, 40), 128, 128), ((1, 17, 21), 64, 80), ((3, 8, 16), 256, 64), ((2, 20, 20), 32, 128)])
def test_dwpw_fused_matches_unfused_and_oracle(shape, c, c2):
    """DWConv 3x3 + SiLU -> Conv 1x1 + SiLU (head.py:49-57) in one launch (dwpw_fused_kernel): bit-identical to
    bsy_dwconv3x3 + bsy_conv2d and equal to the fp32 reference with the depthwise map rounded to fp16."""
    B, H, W = shape
    g = torch.Generator().manual_seed(41)
    x = h16(torch.randn(B, c, H, W, generator=g))
    wd = torch.randn(c, 1, 3, 3, generator=g) * 0.4
    bd = torch.randn(c, generator=g) * 0.2
    w = h16(torch.randn(c2, c, 1, 1, generator=g) * (2.0 / c) ** 0.5)
    b = torch.randn(c2, generator=g) * 0.2
    xd = nhwc(x).half().to(DEV)
    fused = O.dwpw_fused(xd, wd, bd, w, b)
    mid = O.dwconv3x3_nhwc(xd, wd, bd, True)
    wp, bp = O.pack_conv_weight(w, b, DEV)
    two = O.conv2d_nhwc(mid, wp, bp, c2, 1, 1, True)
    torch.cuda.synchronize()
    assert torch.equal(fused, two)
    ref = F.silu(F.conv2d(h16(F.silu(F.conv2d(x, wd, bd, 1, 1, 1, c))), w, b))
    np.testing.assert_allclose(nchw(fused.float().cpu()).numpy(), ref.numpy(), rtol=3e-3, atol=3e-3)


@pytest.mark.parametrize("act,use_res", [(True, False), (False, True)])
def test_dwconv_matches_oracle(act, use_res):
    g = torch.Generator().manual_seed(5)
    x = h16(torch.randn(2, 64, 9, 7, generator=g))
    w = torch.randn(64, 1, 3, 3, generator=g) * 0.4
    b = torch.randn(64, generator=g) * 0.2
    ref = F.conv2d(x, w, b, 1, 1, 1, 64)
    if act:
        ref = F.silu(ref)
    res = h16(torch.randn(ref.shape, generator=g)) if use_res else None
    if use_res:
        ref = ref + res
    out = O.dwconv3x3_nhwc(nhwc(x).half().to(DEV), w, b, act, nhwc(res).half().to(DEV) if use_res else None)
    torch.cuda.synchronize()
    np.testing.assert_allclose(nchw(out.float().cpu()).numpy(), ref.numpy(), rtol=2e-3, atol=2e-3)


def test_sppf_pool_exact():
    g = torch.Generator().manual_seed(6)
    x1 = h16(torch.randn(2, 32, 20, 13, generator=g))
    y = [x1]
    for _ in range(3):
        y.append(F.max_pool2d(y[-1], 5, 1, 2))
    ref = torch.cat(y, 1)
    out = O.sppf_pool_nhwc(nhwc(x1).half().to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(nchw(out.float().cpu()), ref)  # max of fp16 values is exact


@pytest.mark.parametrize("B,heads,N,kd,hd", [(2, 2, 100, 32, 64), (1, 4, 400, 32, 64), (1, 2, 37, 32, 64),
                                             (2, 1, 100, 48, 96),   # one head of 48 / 96: C2PSA at a width multiple of 0.1875 (block.py:4253-4258)
                                             (1, 3, 70, 16, 32), (1, 2, 400, 64, 128), (2, 2, 33, 48, 64), (1, 1, 9, 16, 128)])
def test_attention_matches_oracle(B, heads, N, kd, hd):
    g = torch.Generator().manual_seed(7)
    q = h16(torch.randn(B, heads, kd, N, generator=g))
    k = h16(torch.randn(B, heads, kd, N, generator=g))
    v = h16(torch.randn(B, heads, hd, N, generator=g))
    scale = kd ** -0.5
    attn = ((q.transpose(-2, -1) @ k) * scale).softmax(-1)      # block.py:4284-4285
    ref = (v @ attn.transpose(-2, -1)).reshape(B, heads * hd, N)  # (B, C, N)
    qkv = torch.cat((q.reshape(B, heads * kd, N), k.reshape(B, heads * kd, N), v.reshape(B, heads * hd, N)), 1)
    out = O.attention_nhwc(qkv.transpose(1, 2).contiguous().half().to(DEV), heads, kd, hd, scale)
    torch.cuda.synchronize()
    got = out.float().cpu().transpose(1, 2)
    # P is rounded to fp16 before the PV product (like any fp16 attention): 2^-11 relative on the weights
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=3e-3, atol=3e-3)


def test_detect_decode_golden():
    z = np.load(GOLDEN / "modules.npz")
    raws = [torch.from_numpy(z[f"decode.raw{i}"]) for i in range(3)]  # (2, 69, h, w): 64 box + 5 cls
    box = [nhwc(r[:, :64]).reshape(-1, 64).contiguous().to(DEV) for r in raws]
    cls = [F.pad(nhwc(r[:, 64:]).reshape(-1, 5), (0, 3)).contiguous().to(DEV) for r in raws]
    hw = [tuple(r.shape[2:]) for r in raws]
    y = O.detect_decode(box, cls, hw, [8.0, 16.0, 32.0], 5, torch.float32)
    torch.cuda.synchronize()
    np.testing.assert_allclose(y.cpu().numpy(), z["decode.y"], rtol=1e-5, atol=2e-4)


# ------------------------------------------------------------------------------------------------------------
# whole graph through the engine
# ------------------------------------------------------------------------------------------------------------
def _engine_vs_oracle(tag, dtype):
    z = np.load(GOLDEN / f"graph_{tag}.npz")
    meta = json.loads(str(z["meta"]))
    m = R.Model(meta["family"], meta["scale"], meta["nc"], meta["task"])
    P = R.synth_params(m, meta["seed"])
    eng = YoloEngine(stock_cfg(meta["family"], meta["scale"], meta["nc"], meta["task"]), P)
    stats = []
    si = 0
    while f"x{si}" in z:
        x = torch.from_numpy(z[f"x{si}"])
        y, aux = eng(x.to(dtype).to(DEV))
        torch.cuda.synchronize()
        R.FP16_EMULATION = True  # oracle with the engine's storage precision (fp16 weights/activations, fp32 accumulate)
        try:
            with torch.inference_mode():
                yq, _ = m.forward(P, x)
        finally:
            R.FP16_EMULATION = False
        if meta["task"] == "segment":  # Segment.forward: (cat(y, mc), (raw list, mc, protos)) (head.py:197)
            raws, mc, proto = aux
            assert torch.equal(mc, y[:, 4 + meta["nc"]:])
            pe = np.abs(proto.float().cpu().numpy() - z[f"proto{si}"])
            assert pe.max() < 2e-2 * np.abs(z[f"proto{si}"]).max(), pe.max()
        else:
            raws = aux
        stats.append((y.float().cpu().numpy(), z[f"y{si}"], yq.numpy(), [r.float().cpu().numpy() for r in raws],
                      [z[f"raw{si}_{l}"] for l in range(3)], meta["nc"]))
        si += 1
    eng.close()
    return stats


@pytest.mark.parametrize("tag", ["yolo11n_detect", "yolo11s_detect", "yolo11m_detect", "yolo11n_segment",
                                 "yolov8n_segment", "bsyolo11n_detect", "bsyolo11s_detect", "yolov5n_detect", "yolov5s_detect"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_engine_matches_reference_golden(tag, dtype):
    """Whole graph through the engine vs (a) the REFERENCE's own fp32 CPU outputs (golden) and (b) the oracle run
    with the engine's storage precision.

    The engine stores activations in fp16 (fp32 accumulate, fp32 head outputs + decoder).  (b) isolates kernel errors:
    what is left is summation order plus rare 1-ulp fp16 flips that the ~25-layer random-weight network amplifies in a
    few anchors (measured: max 3.8e-3, mean <= 1.9e-5).  (a) adds the fp16 storage error itself, which a pure-CPU fp16
    emulation shows too (4.1e-3 on the 96x160 case): measured max 5.8e-3 / mean 2.7e-5 on scores, 0.49 px on boxes.
    Box error scales with stride x DFL-bin error (not with image size): 0.5 px = 7.8e-4 of a 640 image."""
    # The BS-YOLO graph (MSCA / ELA gates, nc = 12, these random weights) is more sensitive to fp16 storage: the pure-CPU
    # fp16-storage emulation of the oracle already differs from the fp32 reference by 1.31 px max / 0.074 px mean on boxes
    # and 1.5e-4 mean on scores (stock graphs: 0.41 px / 0.02 px / 2.5e-5), so its bounds are 3x wider.  Per-layer outputs
    # agree with the reference to ~1e-3 of each layer's range in both families (tools/gpu_explore.py layers).
    # Bounds = stated statistics (max, 99.9th percentile, mean), set ~2x above the measurements of tools/parity_stats.py
    # (r02: stock graphs score max 4.6e-3 / p99.9 2.6e-3 / mean 2.4e-5, box max 0.72 px / mean 0.03 px; BS-YOLO score max
    # 5.5e-3 / mean 1.6e-4, box max 2.0 px / mean 0.07 px) -- NOT the north-star's 1e-3: fp16 storage cannot meet that on
    # these random weights (the CPU emulation of fp16 storage shows the same gap), the fp32 mode does and is tested for it
    # (test_engine_fp32_mode_meets_the_north_star_tolerance).
    k = 3.0 if tag.startswith("bsyolo") else 1.0

    def q(e):
        return float(np.quantile(e, 0.999))
    for y, yref, yq, raws, rawref, nc in _engine_vs_oracle(tag, dtype):
        es, eb = np.abs(y[:, 4:4 + nc] - yref[:, 4:4 + nc]), np.abs(y[:, :4] - yref[:, :4])
        assert es.max() < k * 1e-2 and q(es) < k * 5e-3 and es.mean() < k * 1e-4, (es.max(), q(es), es.mean())
        assert eb.max() < k * 1.0 and q(eb) < k * 0.8 and eb.mean() < k * 0.05, (eb.max(), q(eb), eb.mean())
        qs, qb = np.abs(y[:, 4:4 + nc] - yq[:, 4:4 + nc]), np.abs(y[:, :4] - yq[:, :4])
        # mean vs the fp16-emulating oracle: 1e-4 like the bound above -- on the 126-anchor BS-YOLO11s case the statistic moved from
        # 1.4e-4 to 1.7e-4 when the stride-2 convs changed their fp32 summation ORDER (chunk-major K walk): it measures how these
        # random weights amplify 1-ulp flips, not a kernel's accuracy (the per-layer tests do that)
        assert qs.max() < k * 1e-2 and q(qs) < k * 5e-3 and qs.mean() < k * 1e-4, (qs.max(), q(qs), qs.mean())
        assert qb.max() < k * 1.0 and q(qb) < k * 0.8 and qb.mean() < k * 0.03, (qb.max(), q(qb), qb.mean())
        if y.shape[1] > 4 + nc:  # mask coefficients (raw conv outputs, O(1..10) magnitude)
            em = np.abs(y[:, 4 + nc:] - yref[:, 4 + nc:])
            assert em.max() < 2e-2 * np.abs(yref[:, 4 + nc:]).max(), em.max()
        for r, rr in zip(raws, rawref):
            assert np.abs(r - rr).max() / np.abs(rr).max() < 2e-2


def test_engine_yolo11m_1280_config3_shape():
    """BASELINE config 3 geometry (YOLO11m, 1280x1280, A = 33600) on a 2-image shard: engine vs the oracle."""
    m = R.Model("yolo11", "m", 80, "detect")
    P = R.synth_params(m, 3)
    x = torch.rand(2, 3, 1280, 1280, generator=torch.Generator().manual_seed(2))
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.inference_mode():
        yref, _ = m.forward(P, x)
    eng = YoloEngine(stock_cfg("yolo11", "m"), P)
    y, raws = eng(x.to(DEV))  # fp32 in -> fp32 y (an fp16 y would round 1280-px box coordinates to 1-px steps)
    torch.cuda.synchronize()
    assert tuple(y.shape) == (2, 84, 33600) and [tuple(r.shape[2:]) for r in raws] == [(160, 160), (80, 80), (40, 40)]
    es = (y[:, 4:].float().cpu() - yref[:, 4:]).abs()
    eb = (y[:, :4].float().cpu() - yref[:, :4]).abs()
    assert es.max() < 2e-2 and es.mean() < 1e-4, (es.max(), es.mean())
    assert eb.max() < 2.0 and eb.mean() < 0.05, (eb.max(), eb.mean())
    eng.close()


def _damped(P, k=0.8):
    """Deep / wide graphs (v8l, 11x) with the un-damped synthetic BatchNorm gains blow up numerically: every anchor scores
    1.0 and the CPU fp16-storage emulation itself is 0.8 off the fp32 oracle.  BN gains x 0.8 keep activations O(1), so
    the comparison measures the kernels rather than chaos."""
    return {n: (v * k if n.endswith("bn.weight") else v) for n, v in P.items()}


def test_engine_yolov8l_seg_config4_shard():
    """BASELINE config 4 (YOLOv8l-seg, 640x640) on a 2-image shard: boxes, raw logit maps, mask coefficients and prototype
    masks against the oracle (the l scale is not among the golden graphs: 2 x 220 GFLOP on the CPU is the affordable size)."""
    m = R.Model("yolov8", "l", 80, "segment")
    P = _damped(R.synth_params(m, 4))
    x = torch.rand(2, 3, 640, 640, generator=torch.Generator().manual_seed(4))
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.inference_mode():
        yref, (rawref, mcref, pref) = m.forward(P, x)
    eng = YoloEngine(stock_cfg("yolov8", "l", 80, "segment"), P)
    y, (raws, mc, proto) = eng(x.half().to(DEV))
    torch.cuda.synchronize()
    assert tuple(y.shape) == (2, 4 + 80 + 32, 8400) and tuple(proto.shape) == (2, 32, 160, 160)
    y = y.float().cpu()
    es, eb = (y[:, 4:84] - yref[:, 4:84]).abs(), (y[:, :4] - yref[:, :4]).abs()
    assert es.max() < 1e-2 and eb.max() < 1.0 and eb.mean() < 0.05, (es.max(), eb.max(), eb.mean())
    for r, rr in zip(raws, rawref):
        assert (r.float().cpu() - rr).abs().max() / rr.abs().max() < 2e-2
    em = (y[:, 84:] - yref[:, 84:]).abs().max() / yref[:, 84:].abs().max()
    ep = (proto.float().cpu() - pref).abs().max() / pref.abs().max()
    assert em < 2e-2 and ep < 2e-2, (em, ep)
    eng.close()


def test_engine_yolo11x_config5_crops():
    """BASELINE config 5 runs YOLO11x on 640-pixel crops: the x scale (C3k blocks everywhere, 1.5x width -- no fused stem /
    bottleneck, 384 / 768-channel convs) on two crops against the oracle."""
    m = R.Model("yolo11", "x", 80, "detect")
    P = _damped(R.synth_params(m, 5))
    x = torch.rand(2, 3, 640, 640, generator=torch.Generator().manual_seed(5))
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.inference_mode():
        yref, rawref = m.forward(P, x)
    eng = YoloEngine(stock_cfg("yolo11", "x"), P)
    y, raws = eng(x.half().to(DEV))
    torch.cuda.synchronize()
    y = y.float().cpu()
    es, eb = (y[:, 4:] - yref[:, 4:]).abs(), (y[:, :4] - yref[:, :4]).abs()
    assert es.max() < 1e-2 and eb.max() < 1.0 and eb.mean() < 0.05, (es.max(), eb.max(), eb.mean())
    for r, rr in zip(raws, rawref):
        assert (r.float().cpu() - rr).abs().max() / rr.abs().max() < 2e-2
    eng.close()


def test_predict_pipeline_config1_end_to_end():
    """BASELINE config 1 shape of work (YOLO11n, one 1080 x 810 BGR image -> letterbox -> forward -> NMS -> scale_boxes), every
    stage on the device, against the same pipeline through the oracle: same number of detections, every detection matched by
    one of the same class with IoU > 0.95 and |dscore| < 1e-2 for >= 90 % of them (NMS decisions are discrete: fp16 storage
    moves scores by up to ~5e-3, enough to swap a pair of near-tied overlapping boxes)."""
    m = R.Model("yolo11", "n", 80, "detect")
    P = R.synth_params(m, 0)
    rng = np.random.default_rng(8)
    im = rng.integers(0, 256, (1080, 810, 3), dtype=np.uint8)
    xr = LB.preprocess([im], (640, 640), half=False, pt=True, stride=32)
    with torch.inference_mode():
        _, raw0 = m.forward(P, xr)
        # shift the class head so that 2 % of the anchors (~170) pass conf 0.25 on this input
        top = torch.cat([r[:, 64:].flatten(2) for r in raw0], 2).amax(1).flatten()
        shift = float(np.log(0.25 / 0.75) - torch.quantile(top, 0.98))
        for k in P:
            if ".cv3." in k and k.endswith(".2.bias"):
                P[k] = P[k] + shift
        yr, _ = m.forward(P, xr)
    ref = PP.non_max_suppression(yr.clone(), 0.25, 0.7)[0]
    ref[:, :4] = PP.scale_boxes(xr.shape[2:], ref[:, :4], im.shape)
    xd = HLB.preprocess([im], (640, 640), half=True, pt=True, stride=32, device=DEV)
    assert tuple(xd.shape) == tuple(xr.shape)
    eng = YoloEngine(stock_cfg("yolo11", "n"), P)
    y, _ = eng(xd)
    det, counts = HN.nms_batched(y, 0.25, 0.7, max_det=300)
    HN.scale_boxes_batched(det, counts, xd.shape[2:], [im.shape])
    torch.cuda.synchronize()
    n = int(counts[0])
    got = det[0, :n].cpu()
    # detections whose score sits within the fp16-storage error (~5e-3) of conf 0.25 may fall on either side: a tenth of the count
    assert 5 <= ref.shape[0] <= 300 and abs(n - ref.shape[0]) <= max(3, ref.shape[0] // 10), (n, ref.shape[0])
    from oracle import val_ref as V
    iou = torch.from_numpy(V.box_iou(ref[:, :4].numpy(), got[:, :4].numpy()))
    iou = iou * (ref[:, 5:6] == got[:, 5][None]).float()
    best, j = iou.max(1)
    ok = (best > 0.95) & ((ref[:, 4] - got[j, 4]).abs() < 1e-2)
    assert ok.float().mean() >= 0.9, (ok.float().mean(), n, ref.shape[0], best, (ref[:, 4] - got[j, 4]).abs())
    eng.close()


@pytest.mark.parametrize("scale,shape", [("n", (1, 32, 32)), ("s", (3, 96, 224)), ("s", (2, 160, 32)), ("n", (5, 64, 416)), ("s", (16, 320, 320))])
def test_engine_tuned_equals_untuned_bit_for_bit(scale, shape):
    """Results do not depend on what the autotuner timed (round-2 VERDICT 3): every kernel configuration that is valid for a layer
    sums in that layer's K walk (conv_mfma.hip conv_korder), so a tuned engine -- patch / persistent / big-tile kernels wherever
    they won on THIS box -- returns bit for bit what the heuristic plan (autotune off) returns, and two separately tuned engines
    agree with each other, through both tuners."""
    m = R.Model("yolo11", scale, 80, "detect")
    P = R.synth_params(m, 2)
    cfg = stock_cfg("yolo11", scale)
    B, H, W = shape
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(H * W)).half().to(DEV)
    plain = YoloEngine(cfg, P, autotune=False)
    yp, rp = plain(x)
    tuned_a = YoloEngine(cfg, P, autotune=True)
    tuned_b = YoloEngine(cfg, P, autotune=True)
    tuned_b.tune_mode = "0"  # the one-layer-at-a-time tuner (bsy_plan_autotune)
    ya, ra = tuned_a(x)
    yb, rb = tuned_b(x)
    torch.cuda.synchronize()
    assert tuned_a.tune_stats["timed_ops"] > 0 and tuned_b.tune_stats["timed_ops"] > 0
    assert torch.equal(ya, yp) and torch.equal(yb, yp)
    for a, b, c in zip(ra, rb, rp):
        assert torch.equal(a, c) and torch.equal(b, c)
    for e in (plain, tuned_a, tuned_b):
        e.close()


@pytest.mark.parametrize("fam,scale,nc,task,shape", [("yolo11", "s", 80, "detect", (8, 640, 640)), ("yolo11", "n", 80, "segment", (2, 96, 160)),
                                                     ("bsyolo11", "n", 12, "detect", (3, 128, 96))])
@pytest.mark.parametrize("own_stream", [False, True])
def test_engine_graph_mode_equals_eager(fam, scale, nc, task, shape, own_stream):
    """Graph mode (bsy_plan_graph_launch: the forward captured once per set of buffer addresses, replayed as one hipGraph launch,
    head lanes as graph edges) returns the eager engine's bits; a graph is captured once per ring slot and then replayed; a
    returned tensor is overwritten graph_ring forwards later and not before; on the legacy default stream the engine captures
    on a stream of its own."""
    m = R.Model("yolo11" if fam == "yolo11" else fam, scale, nc, task)
    P = R.synth_params(m, 3)
    cfg = stock_cfg(fam, scale, nc, task)
    B, H, W = shape
    xs = [torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(10 + i)).half().to(DEV) for i in range(2)]
    eager = YoloEngine(cfg, P, autotune=False)
    graph = YoloEngine(cfg, P, autotune=False, graph=True, graph_ring=2)
    pg, _ = graph.plan_for(B, H, W, torch.float16, torch.float16)
    assert any(o.get("lane", 0) > 0 for o in pg.ops)  # lanes at every size in graph mode
    stream = torch.cuda.Stream(device=DEV) if own_stream else torch.cuda.current_stream(DEV)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        ref = [eager(x) for x in xs]
        outs = []
        for rep in range(3):
            for x in xs:
                y, extra = graph(x)
                outs.append((y, extra, y.clone()))
    torch.cuda.synchronize()
    # 2 inputs x 2 ring slots: the pairing (input, slot) repeats with period 2 -> two captures, the rest replays
    assert graph.graph_stats == {"captures": 2, "replays": 4, "eager": 0}, graph.graph_stats
    for i, (y, extra, snap) in enumerate(outs):
        y_ref, e_ref = ref[i % 2]
        assert torch.equal(snap, y_ref)                   # what the call returned, when it returned
        assert y.data_ptr() == outs[i % 2][0].data_ptr()  # ring of two: the same two buffers over and over
    flat = lambda e: [t for t in (e if isinstance(e, (list, tuple)) else [e]) for t in (t if isinstance(t, (list, tuple)) else [t]) if t is not None]  # noqa: E731
    for a, b in zip(flat(outs[-1][1]), flat(ref[1][1])):
        assert torch.equal(a, b)
    eager.close()
    graph.close()


@pytest.mark.parametrize("scale,shape", [("s", (8, 640, 640)), ("n", (6, 320, 224))])
def test_engine_latency_mode_split_k(scale, shape):
    """Latency mode (round 4): long-K conv layers with few tiles per image run split-K -- the slices of a layer's K
    walk are computed by separate workgroups of one launch and added in slice order.  The split factors are a function of the layer
    shape only (Plan.split_factors), so (a) a batch cut into shards returns the bits of the whole batch, (b) a tuned engine returns
    the bits of the untuned one (every tile configuration walks the slices alike), (c) the result stays within f32 summation-order
    distance of the default mode and inside the fp16 path's bounds against the oracle.  (Which layers are split: Plan.split_factors --
    only long, thin ones; splitting everything that has few tiles was measured SLOWER.)"""
    m = R.Model("yolo11", scale, 80, "detect")
    P = R.synth_params(m, 9)
    cfg = stock_cfg("yolo11", scale)
    B, H, W = shape
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(B * H)).half().to(DEV)
    lat = YoloEngine(cfg, P, autotune=False, latency=True)
    plan, _ = lat.plan_for(B, H, W, torch.float16, torch.float16)
    nsplit = sum(1 for o in plan.ops if o.get("ksplit"))
    assert nsplit >= 1, nsplit
    y, raws = lat(x)
    ys = [lat(x[i:i + B // 2]) for i in (0, B // 2)]                # two shards: other plans of the same engine
    tuned = YoloEngine(cfg, P, autotune=True, latency=True)
    yt, rt = tuned(x)
    base = YoloEngine(cfg, P, autotune=False)
    yb, rb = base(x)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([ys[0][0], ys[1][0]]), y)
    for l in range(3):
        assert torch.equal(torch.cat([ys[0][1][l], ys[1][1][l]]), raws[l]) and torch.equal(rt[l], raws[l])
    assert torch.equal(yt, y) and tuned.tune_stats["timed_ops"] > 0
    d = (y.float() - yb.float()).abs()  # another f32 summation order + fp16 storage downstream: the fp16 path's own class of distance
    assert float(d[:, 4:].max()) < 2e-2 and float(d[:, :4].max()) < 4.0 and float(d[:, 4:].mean()) < 1e-4, (float(d[:, 4:].max()), float(d[:, :4].max()))
    with torch.inference_mode():
        yref, _ = m.forward(P, x.float().cpu())
    # against the oracle the split plan is as good as the default one (these un-calibrated seed-9 weights put the fp16 path itself at
    # ~1e-2 / 2.5 px on the s scale: the bound is the default engine's own distance, not an absolute)
    e, eb = (y.float().cpu() - yref).abs(), (yb.float().cpu() - yref).abs()
    assert float(e[:, 4:].max()) <= 1.5 * float(eb[:, 4:].max()) + 1e-3 and float(e[:, :4].max()) <= 1.5 * float(eb[:, :4].max()) + 0.1, \
        (float(e[:, 4:].max()), float(eb[:, 4:].max()), float(e[:, :4].max()), float(eb[:, :4].max()))
    assert float(e[:, 4:].mean()) <= 1.2 * float(eb[:, 4:].mean()) + 1e-6
    for eng in (lat, tuned, base):
        eng.close()


def test_engine_graph_mode_follows_a_weight_reload():
    """captured graphs hold addresses inside the weight blob.  Reloading weights on an engine that has captured graphs
    (bsy_engine_load_weights frees and reallocates the blob) must drop them: the next graph launch re-captures and returns the bits
    of an eager engine built from the new weights -- never a replay that reads the freed blob."""
    m = R.Model("yolo11", "n", 80, "detect")
    P0, P1 = R.synth_params(m, 3), R.synth_params(m, 4)
    cfg = stock_cfg("yolo11", "n")
    x = torch.rand(2, 3, 96, 160, generator=torch.Generator().manual_seed(5)).half().to(DEV)
    stream = torch.cuda.Stream(device=DEV)
    g = YoloEngine(cfg, P0, autotune=False, graph=True, graph_ring=1)
    e0, e1 = YoloEngine(cfg, P0, autotune=False), YoloEngine(cfg, P1, autotune=False)
    with torch.cuda.stream(stream):
        a0 = g(x)[0].clone()
        a0b = g(x)[0].clone()
        assert g.graph_stats == {"captures": 1, "replays": 1, "eager": 0}, g.graph_stats
        g.load_weights(P1)
        a1 = g(x)[0].clone()
        a1b = g(x)[0].clone()
        r0, r1 = e0(x)[0], e1(x)[0]
    torch.cuda.synchronize()
    assert g.graph_stats == {"captures": 2, "replays": 2, "eager": 0}, g.graph_stats  # the old graph was not replayed
    assert torch.equal(a0, r0) and torch.equal(a0b, r0)
    assert torch.equal(a1, r1) and torch.equal(a1b, r1) and not torch.equal(r0, r1)
    for e in (g, e0, e1):
        e.close()


def test_engine_serialises_forwards_across_streams_and_threads():
    """All plans of an engine share one liveness-packed arena (round 2), so two forwards must never overlap: forwards
    enqueued on different torch streams, of different shapes (different plans aliasing the same memory), and from two host threads
    return what the same calls return one after the other."""
    import threading
    m = R.Model("yolo11", "s", 80, "detect")
    P = R.synth_params(m, 4)
    eng = YoloEngine(stock_cfg("yolo11", "s"), P, autotune=False)
    shapes = [(6, 320, 320), (2, 640, 480), (9, 256, 256)]
    xs = [torch.rand(*((b, 3, h, w)), generator=torch.Generator().manual_seed(i)).half().to(DEV) for i, (b, h, w) in enumerate(shapes)]
    ref = [eng(x)[0].clone() for x in xs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=DEV) for _ in range(3)]
    outs = []
    for rep in range(4):  # back to back on three streams, no host synchronisation in between
        for i, x in enumerate(xs):
            with torch.cuda.stream(streams[(i + rep) % 3]):
                outs.append((i, eng(x)[0]))
    torch.cuda.synchronize()
    assert all(torch.equal(y, ref[i]) for i, y in outs)
    got = {}

    def worker(i):
        with torch.cuda.stream(streams[i]):
            for _ in range(5):
                got[i] = eng(xs[i])[0]
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    torch.cuda.synchronize()
    assert all(torch.equal(got[i], ref[i]) for i in range(3))
    eng.close()


@pytest.mark.parametrize("scale,shape", [("n", (1, 32, 32)), ("s", (3, 96, 224)), ("s", (2, 160, 32)), ("n", (5, 64, 416))])
def test_engine_fused_equals_plain_on_odd_shapes(scale, shape):
    """Every fusion against the plain plan (one launch per layer) on small, narrow and wide inputs, both with the heuristic
    configurations: the conv kernels and their fused forms agree bit for bit; what differs is the fused Detect decoder (DFL
    expectation summed across a lane pair, head.py:141-146 in the conv epilogue) against the stand-alone decode kernel -- a few f32
    ulps of a distance, i.e. at most one f16 ulp of an output."""
    m = R.Model("yolo11", scale, 80, "detect")
    P = R.synth_params(m, 2)
    cfg = stock_cfg("yolo11", scale)
    full = YoloEngine(cfg, P, autotune=False)
    plain = YoloEngine(cfg, P, fuse_stem=False, fuse_bneck=False, fuse_head=False, fuse_dwpw=False, merge_c3k=False, fuse_tail=False, autotune=False)
    B, H, W = shape
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(H * W)).half().to(DEV)
    yf, rf = full(x)
    yp, rp = plain(x)
    torch.cuda.synchronize()
    d = (yf.float() - yp.float()).abs()
    # one ulp of a score is <= 9.8e-4, of a coordinate in [256, 512) 0.25 px (the widest input here is 416 pixels)
    assert d[:, 4:].max() < 2e-3 and d[:, :4].max() <= 0.25, (d[:, 4:].max(), d[:, :4].max())
    assert d[:, 4:].mean() < 5e-5 and d[:, :4].mean() < 0.02, (d[:, 4:].mean(), d[:, :4].mean())
    for a, b in zip(rf, rp):
        assert torch.equal(a, b)  # the raw maps are the conv outputs themselves: identical
    full.close()
    plain.close()


@pytest.mark.parametrize("fam,scale,nc,task,shapes", [
    ("yolo11", "s", 80, "detect", [(2, 640, 640), (1, 1280, 1280), (3, 64, 96), (1, 32, 32)]),
    ("yolo11", "n", 80, "segment", [(2, 320, 320), (1, 160, 96)]),
    ("yolov8", "s", 80, "detect", [(2, 320, 640)]),
    ("bsyolo11", "n", 12, "detect", [(1, 1024, 1024), (2, 96, 160)]),
])
def test_engine_no_out_of_bounds_stores(fam, scale, nc, task, shapes, monkeypatch):
    """Every workspace buffer gets a 4 KiB guard band (BSY_PLAN_GUARD, a test aid of bsy_plan_create): after tuned and
    un-tuned forwards on whole, ragged and tiny maps no kernel of the plan has stored outside its destination."""
    from bs_yolo_amd.plan import Plan
    from bs_yolo_amd.weights import synth_state_dict
    monkeypatch.setenv("BSY_PLAN_GUARD", "4096")
    cfg = stock_cfg(fam, scale, nc, task)
    sd = synth_state_dict(Plan(cfg, 1, 64, 64), seed=0)
    for tune in (False, True):
        eng = YoloEngine(cfg, sd, autotune=tune)
        for B, H, W in shapes:
            eng(torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(H + W)).half().to(DEV))
        assert eng.check_guards() == []
        eng.close()


@pytest.mark.parametrize("precision", ["fp32x", "fp32"])
def test_fp32_storage_modes_store_inside_their_buffers_and_read_inside_the_weights(precision, monkeypatch):
    """The two tests above for the fp32-storage modes (round 4: the fp32x kernels -- implicit GEMM, 256-pixel tiles, patch kernel
    with 8 x 16 and 6 x 20 tiles, image conv, attention -- and the LDS pool / windowed depthwise kernels both modes share): guard bands
    behind every workspace buffer stay intact on whole, ragged and tiny maps, latency-mode split-K slabs included for the fp16 path,
    and a weight blob followed by poison returns the plain engine's bits."""
    cfg = stock_cfg("yolo11", "s")
    m = R.Model("yolo11", "s", 80, "detect")
    P = R.synth_params(m, 2)
    shapes = [(2, 160, 32), (3, 96, 224), (1, 640, 416), (5, 64, 64)]
    monkeypatch.setenv("BSY_PLAN_GUARD", "4096")
    eng = YoloEngine(cfg, P, precision=precision)
    for B, H, W in shapes:
        eng(torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(H + W)).to(DEV))
    assert eng.check_guards() == []
    eng.close()
    lat = YoloEngine(cfg, P, autotune=False, latency=True)
    for B, H, W in shapes:
        lat(torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(H + W)).half().to(DEV))
    assert lat.check_guards() == []
    lat.close()
    monkeypatch.delenv("BSY_PLAN_GUARD")
    plain = YoloEngine(cfg, P, precision=precision)
    monkeypatch.setenv("BSY_WEIGHT_GUARD", str(1 << 20))
    poisoned = YoloEngine(cfg, P, precision=precision)
    for B, H, W in shapes:
        x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(H * W)).to(DEV)
        y0, a0 = plain(x)
        y1, a1 = poisoned(x)
        assert torch.equal(y0, y1) and all(torch.equal(p, q) for p, q in zip(a0, a1))
    plain.close()
    poisoned.close()


@pytest.mark.parametrize("fam,scale,nc,task,shapes", [
    ("yolo11", "n", 80, "detect", [(1, 32, 32), (3, 96, 224)]),
    ("yolo11", "s", 80, "detect", [(2, 160, 32), (8, 320, 320)]),
    ("yolo11", "m", 80, "segment", [(1, 64, 96)]),
    ("yolov8", "s", 80, "detect", [(2, 64, 64)]),
    ("bsyolo11", "s", 12, "detect", [(1, 96, 160)]),
])
def test_engine_no_harmful_reads_past_the_weight_blob(fam, scale, nc, task, shapes, monkeypatch):
    """The weight blob followed by 1 MiB of poison (BSY_WEIGHT_GUARD: f16 NaNs) returns bit for bit what the plain engine
    returns, tuned (every kernel configuration the autotuner tries runs on the poisoned blob) and un-tuned: no kernel uses
    bytes it read past the packed weights (the guard bands only caught out-of-bounds STORES)."""
    from bs_yolo_amd.plan import Plan
    from bs_yolo_amd.weights import synth_state_dict
    cfg = stock_cfg(fam, scale, nc, task)
    sd = synth_state_dict(Plan(cfg, 1, 64, 64), seed=3)
    monkeypatch.delenv("BSY_WEIGHT_GUARD", raising=False)
    plain = YoloEngine(cfg, sd, autotune=False)
    monkeypatch.setenv("BSY_WEIGHT_GUARD", str(1 << 20))
    poisoned = YoloEngine(cfg, sd, autotune=False)
    tuned = YoloEngine(cfg, sd, autotune=True)
    for (B, H, W) in shapes:
        x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(H + W)).half().to(DEV)
        y0, a0 = plain(x)
        y1, a1 = poisoned(x)
        y2, _ = tuned(x)
        assert torch.equal(y0, y1) and bool(torch.isfinite(y2.float()).all())
        r0, r1 = (a0[0], a1[0]) if task == "segment" else (a0, a1)
        assert all(torch.equal(p, q) for p, q in zip(r0, r1))
        # the tuned plan may pick kernels with another (equally valid) summation order: close, and never NaN
        assert float((y2.float() - y0.float()).abs().max()) < 0.05 * max(1.0, float(y0.float().abs().max()))
    for e in (plain, poisoned, tuned):
        e.close()


@pytest.mark.parametrize("fam,scale,nc,shape", [("yolo11", "s", 80, (2, 640, 640)), ("yolo11", "n", 80, (3, 512, 544)), ("yolov8", "n", 80, (1, 512, 512)),
                                                ("bsyolo11", "n", 12, (1, 640, 512))])
def test_engine_box_branch_tail_is_bit_identical(fam, scale, nc, shape, monkeypatch):
    """cv2.i.1 + cv2.i.2 + DFL in one launch (conv3x3_patch_kernel TAIL) against the two launches: y and the raw maps bit for bit, f16 and
    f32 outputs, 8x16 and 6x20 tiles, ragged tile rows / columns.  Every level is at least 16 pixels high and wide, so the un-tuned plan
    runs the separate 3x3 conv on the patch kernel too (the heuristic's condition): same K walk on both sides; smaller maps are
    covered with tolerances by test_engine_tuned_fused_equals_plain_on_odd_shapes and the golden-graph tests."""
    from bs_yolo_amd.plan import Plan
    from bs_yolo_amd.weights import synth_state_dict
    cfg = stock_cfg(fam, scale, nc)
    sd = synth_state_dict(Plan(cfg, 1, 64, 64), seed=2)
    B, H, W = shape
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(8))
    outs = []
    for tail in ("1", "0"):
        monkeypatch.setenv("BSY_FUSE_BOXTAIL", tail)
        eng = YoloEngine(cfg, sd, autotune=False)
        plan = eng.plan_for(B, H, W, torch.float16, torch.float16)[0]
        assert sum(1 for o in plan.ops if o["kind"] == L.OP_CONV and o.get("mid_c")) == (3 if tail == "1" else 0)
        y16, r16 = eng(x.half().to(DEV))
        y32, r32 = eng(x.to(DEV))
        torch.cuda.synchronize()
        outs.append([y16.clone(), y32.clone()] + [r.clone() for r in r16] + [r.clone() for r in r32])
        eng.close()
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("fam,scale,nc,shape", [("yolo11", "s", 80, (8, 640, 640)), ("bsyolo11", "n", 12, (2, 1024, 1024))])
def test_engine_lanes_equal_serial_schedule(fam, scale, nc, shape, monkeypatch):
    """The product schedule (Detect branches on six side streams, running beside each other) returns bit for bit what the
    same plan returns with every op on one stream (BSY_LANES=0) -- 20 forwards.  (a depthwise kernel
    whose f16 converts fed packed f32 FMAs failed exactly this kind of comparison.)"""
    from bs_yolo_amd.plan import Plan
    from bs_yolo_amd.weights import synth_state_dict
    cfg = stock_cfg(fam, scale, nc)
    sd = synth_state_dict(Plan(cfg, 1, 64, 64), seed=0)
    B, H, W = shape
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(5)).half().to(DEV)
    monkeypatch.setenv("BSY_LANES", "0")
    ser = YoloEngine(cfg, sd, autotune=False)
    y0, r0 = ser(x)
    assert max(o.get("lane", 0) for o in ser.plan_for(B, H, W, torch.float16, torch.float16)[0].ops) == 0
    monkeypatch.setenv("BSY_LANES", "1")
    eng = YoloEngine(cfg, sd, autotune=False)
    assert max(o.get("lane", 0) for o in eng.plan_for(B, H, W, torch.float16, torch.float16)[0].ops) >= 5
    for _ in range(20):
        y, r = eng(x)
        assert torch.equal(y, y0) and all(torch.equal(a, b) for a, b in zip(r, r0))
    eng.close()
    ser.close()


def test_engine_batch_independence_and_determinism():
    """Images are independent units: a batch equals its images run one by one; reruns are bit-identical."""
    m = R.Model("yolo11", "n", 80, "detect")
    P = R.synth_params(m, 0)
    eng = YoloEngine(stock_cfg("yolo11", "n"), P, autotune=False)  # same kernel configuration for every batch size
    x = torch.rand(3, 3, 96, 64, generator=torch.Generator().manual_seed(1)).half().to(DEV)
    y, _ = eng(x)
    y2, _ = eng(x)
    assert torch.equal(y, y2)
    for i in range(3):
        yi, _ = eng(x[i:i + 1])
        assert torch.equal(yi[0], y[i])
    eng.close()


@pytest.mark.parametrize("scale", ["n", "s"])
def test_engine_stem_fusion_is_bit_identical(scale, monkeypatch):
    """The OP_STEM / OP_BNECK / OP_C3K2 / OP_DWPW plan (fused launches, merged C3k branch convs) returns exactly what the plan of
    plain convs returns."""
    m = R.Model("yolo11", scale, 80, "detect")
    P = R.synth_params(m, 0)
    cfg = stock_cfg("yolo11", scale)
    # autotune off: the tuner may pick kernels with different (equally valid) summation orders per plan; no patch kernel
    # (chunk-major K order): the fused Bottleneck / C3k2 kernels sum tap-major like the implicit-GEMM kernel
    monkeypatch.setenv("BSY_NO_PATCH", "1")
    fused = YoloEngine(cfg, P, fuse_stem=True, fuse_bneck=True, fuse_dwpw=True, autotune=False)
    plain = YoloEngine(cfg, P, fuse_stem=False, fuse_bneck=False, fuse_dwpw=False, merge_c3k=False, fuse_tail=False, autotune=False)
    x = torch.rand(2, 3, 160, 96, generator=torch.Generator().manual_seed(3)).half().to(DEV)
    pf, _ = fused.plan_for(2, 160, 96, torch.float16, torch.float16)
    pp, _ = plain.plan_for(2, 160, 96, torch.float16, torch.float16)
    assert pf.ops[0]["kind"] == L.OP_STEM and pp.ops[0]["kind"] == L.OP_CONV_FIRST
    assert any(o["kind"] == L.OP_C3K2 for o in pf.ops) and ((scale != "n") or any(o["kind"] == L.OP_BNECK for o in pf.ops))
    assert not any(o["kind"] in (L.OP_BNECK, L.OP_C3K2) for o in pp.ops)
    yf, rf = fused(x)
    yp, rp = plain(x)
    assert torch.equal(yf, yp)
    for a, b in zip(rf, rp):
        assert torch.equal(a, b)
    fused.close()
    plain.close()


@pytest.mark.parametrize("scale,shape", [("s", (3, 160, 224)), ("n", (2, 96, 160)), ("m", (1, 256, 192)), ("l", (1, 160, 160)), ("s", (5, 320, 320))])
def test_engine_chain_fusion_is_bit_identical(scale, shape, monkeypatch):
    """Two 1x1 convs chained per pixel as one launch (csrc/chain1x1.hip; C3k2.cv1 -> C3k.cv1|cv2, C2PSA.cv1 -> qkv, C3k.cv3 -> C3k2.cv2,
    ffn[1] -> C2PSA.cv2, SPPF.cv2 -> C2PSA.cv1 and C3k -> C3k at the scales whose widths fit; block.py:3796-3815, :4429-4468) return
    exactly what the two launches return -- the prediction, the raw maps and EVERY top-level layer output: 128- and 256-cout passes,
    one- and two-operand first convs, resident tiles of 128 / 256 channels, the shortcut in the first epilogue (ffn[1]) and in the
    second (ffn.0 -> ffn.1 at scale n), pixel counts that are not multiples of the 128-pixel tile (105 ... 1600 pixels)."""
    monkeypatch.setenv("BSY_CHAIN_MIN_PIXELS", "1")
    monkeypatch.setenv("BSY_ARENA_REUSE", "0")
    m = R.Model("yolo11", scale, 80, "detect")
    P = R.synth_params(m, 4)
    cfg = stock_cfg("yolo11", scale)
    fused = YoloEngine(cfg, P, fuse_chain=True, autotune=False)
    plain = YoloEngine(cfg, P, fuse_chain=False, autotune=False)
    B, H, W = shape
    pf, hf = fused.plan_for(B, H, W, torch.float16, torch.float16)
    pp, hp = plain.plan_for(B, H, W, torch.float16, torch.float16)
    nchain = sum(o["kind"] == L.OP_CHAIN for o in pf.ops)
    assert nchain >= {"n": 6, "s": 8, "m": 14, "l": 21}[scale] and len(pp.ops) - len(pf.ops) == nchain and not any(o["kind"] == L.OP_CHAIN for o in pp.ops)
    assert sum(o.get("mfma_flops", 0) for o in pf.ops) == sum(o.get("mfma_flops", 0) for o in pp.ops)
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(H + W)).half().to(DEV)
    yf, rf = fused(x)
    yp, rp = plain(x)
    torch.cuda.synchronize()

    def same(a, b, what):  # equal element by element; the deep scales overflow f16 with these synthetic weights: a NaN equals a NaN here
        bad = (a != b) & ~(a.isnan() & b.isnan())
        assert not bad.any(), f"{what}: {int(bad.sum())} of {a.numel()} differ (NaN: {int(a.isnan().sum())} / {int(b.isnan().sum())}), first at {bad.nonzero()[:3].tolist()}"

    for i, (tf, tp) in enumerate(zip(pf.layer_out, pp.layer_out)):
        if tf is None or isinstance(tf, list):
            continue
        same(fused.read_view(pf, hf, tf), plain.read_view(pp, hp, tp), f"layer {i}")
    same(yf.float(), yp.float(), "y")
    for j, (a, b) in enumerate(zip(rf, rp)):
        same(a.float(), b.float(), f"raw map {j}")
    fused.close()
    plain.close()


def test_engine_chain_fusion_threshold_and_batch_independence():
    """The chain launch (opt-in: fuse_chain=True / BSY_FUSE_CHAIN=1) is taken from plan.CHAIN_MIN_PIXELS pixels per layer on (one
    128-pixel tile per workgroup); below it the plan keeps the two launches.  Both return the same bits, so a batch and its shards
    agree whichever side of the threshold each falls on."""
    from bs_yolo_amd.plan import CHAIN_MIN_PIXELS, Plan
    cfg = stock_cfg("yolo11", "s")
    big, small = Plan(cfg, 64, 640, 640, fuse_chain=True), Plan(cfg, 8, 640, 640, fuse_chain=True)
    assert sum(o["kind"] == L.OP_CHAIN for o in big.ops) == 8
    assert [o["name"] for o in small.ops if o["kind"] == L.OP_CHAIN] == []  # 8 x 40 x 40 = 12 800 pixels < CHAIN_MIN_PIXELS
    assert CHAIN_MIN_PIXELS == 24576
    assert not any(o["kind"] == L.OP_CHAIN for o in Plan(cfg, 64, 640, 640).ops)  # opt-in: measured slower than the pairs at this size
    m = R.Model("yolo11", "s", 80, "detect")
    P = R.synth_params(m, 9)
    eng = YoloEngine(cfg, P, autotune=False, fuse_chain=True)
    x = torch.rand(64, 3, 320, 320, generator=torch.Generator().manual_seed(5)).half().to(DEV)
    y, _ = eng(x)          # 64 x 20 x 20 = 25 600 pixels at stride 16: chains there, none at stride 32
    y8, _ = eng(x[8:16])   # no chain anywhere
    p64, _ = eng.plan_for(64, 320, 320, torch.float16, torch.float16)
    p8, _ = eng.plan_for(8, 320, 320, torch.float16, torch.float16)
    assert sum(o["kind"] == L.OP_CHAIN for o in p64.ops) == 2 and not any(o["kind"] == L.OP_CHAIN for o in p8.ops)
    assert torch.equal(y[8:16], y8)
    eng.close()


def test_engine_bsyolo_large_input_matches_oracle():
    """BS-YOLO11n on a 1024 x 1280 input (the golden graphs stop at 96 x 160): 128 x 160 ELA rows / columns, a 32 x 40 MSCA
    map through the one-launch spatial kernel, depthwise windows over 256 x 320 maps -- against the oracle with the engine's
    storage precision: raw maps within 2 % of their range, decoded outputs by mean error (see below), reruns bit-identical."""
    m = R.Model("bsyolo11", "n", 12, "detect")
    P = R.synth_params(m, 1)
    x = torch.rand(1, 3, 1024, 1280, generator=torch.Generator().manual_seed(77))
    eng = YoloEngine(stock_cfg("bsyolo11", "n", 12), P, autotune=False)
    y, raws = eng(x.half().to(DEV))
    y2, _ = eng(x.half().to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    R.FP16_EMULATION = True
    try:
        with torch.inference_mode():
            yq, rq = m.forward(P, x.half().float())
    finally:
        R.FP16_EMULATION = False
    for a, b in zip(raws, rq):
        assert (a.float().cpu() - b).abs().max() < 2e-2 * b.abs().max(), (a.float().cpu() - b).abs().max() / b.abs().max()
    # Decoded outputs: with these synthetic weights the graph amplifies fp16 storage rounding at this size -- the CPU
    # fp16-emulating oracle itself is 0.21 (scores) / 37 px (boxes) off the fp32 oracle on 1.8 % of the 26 880 anchors, mean
    # 0.07 px -- so the bar is statistical: mean errors and the share of anchors that moved.
    d = (y.float().cpu() - yq).abs()
    moved = (d[:, :4].amax(1) > 3.0).float().mean()
    assert d[:, 4:].mean() < 1e-3 and d[:, :4].mean() < 0.2 and moved < 0.05, (d[:, 4:].mean(), d[:, :4].mean(), moved)
    eng.close()


@pytest.mark.parametrize("scale,shape", [("n", (2, 96, 160)), ("s", (1, 160, 128)), ("n", (1, 640, 640))])
def test_engine_pmsfa_pass_through_half_is_bit_identical(scale, shape, monkeypatch):
    """PMSFA.conv3 (block.py:3042) runs as a depthwise 7 x 7 over [q1 | q2] whose q2 half carries an identity kernel; the tiled
    depthwise kernel writes x + 0 for that half instead of walking 49 taps (csrc/bsyolo_ops.hip dwconv_tile_kernel, ident_c0).  The
    window kernel (BSY_NO_DWTILE=1) walks them all: the two forwards must agree bit for bit."""
    m = R.Model("bsyolo11", scale, 12, "detect")
    P = R.synth_params(m, 5)
    cfg = stock_cfg("bsyolo11", scale, 12)
    B, H, W = shape
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(H + W)).half().to(DEV)
    eng = YoloEngine(cfg, P, autotune=False, fuse_pmsfa=False)
    plan, _ = eng.plan_for(B, H, W, torch.float16, torch.float16)
    assert sum(1 for o in plan.ops if o["kind"] == L.OP_DWCONV_G and o.get("mid_c", 0) > 0) >= 6
    y1, r1 = eng(x)
    torch.cuda.synchronize()
    monkeypatch.setenv("BSY_NO_DWTILE", "1")
    y2, r2 = eng(x)
    torch.cuda.synchronize()
    monkeypatch.delenv("BSY_NO_DWTILE")
    assert torch.equal(y1, y2)
    for a, b in zip(r1, r2):
        assert torch.equal(a, b)
    eng.close()


@pytest.mark.parametrize("scale,shape", [("s", (2, 96, 160)), ("s", (1, 640, 640)), ("n", (3, 64, 64)), ("s", (2, 416, 288)), ("n", (1, 1280, 1280)),
                                         ("m", (1, 160, 96))])
def test_engine_pmsfa_tail_fusion_is_bit_identical(scale, shape):
    """BS-YOLO: PMSFA's depthwise 5 x 5 -> depthwise 7 x 7 -> concat -> 1 x 1 conv + shortcut as one launch (block.py:3046-3054,
    csrc/pmsfa_fused.hip) returns exactly what the three launches it replaces return: widths 32 / 64 (scale s: model.2 / 4 / 6; n: model.4 / 6 / 8;
    m: model.2 -- 16- and 128-wide modules stay unfused), maps that are not multiples of the 16 x 16 / 8 x 16 tiles (104 x 72 ...
    13 x 9: partial tiles, halos that leave the image on every side), 2 x 2 maps (the whole halo is padding) and 320 x 320 maps."""
    m = R.Model("bsyolo11", scale, 12, "detect")
    P = R.synth_params(m, 11)
    cfg = stock_cfg("bsyolo11", scale, 12)
    fused = YoloEngine(cfg, P, fuse_pmsfa=True, autotune=False)
    plain = YoloEngine(cfg, P, fuse_pmsfa=False, autotune=False)
    B, H, W = shape
    pf, _ = fused.plan_for(B, H, W, torch.float16, torch.float16)
    pp, _ = plain.plan_for(B, H, W, torch.float16, torch.float16)
    nt = sum(o["kind"] == L.OP_PMSFA_TAIL for o in pf.ops)
    assert nt >= (1 if scale == "m" else 4) and len(pp.ops) - len(pf.ops) == 2 * nt and not any(o["kind"] == L.OP_PMSFA_TAIL for o in pp.ops)
    assert {o["dst"].C for o in pf.ops if o["kind"] == L.OP_PMSFA_TAIL} <= {32, 64}
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(H + W)).half().to(DEV)
    yf, rf = fused(x)
    yp, rp = plain(x)
    assert torch.equal(yf, yp)
    for a, b in zip(rf, rp):
        assert torch.equal(a, b)
    fused.close()
    plain.close()


@pytest.mark.parametrize("shape", [(2, 96, 160), (1, 640, 640), (3, 64, 64), (1, 1280, 1280)])
def test_engine_msca_spatial_fusion_is_bit_identical(shape):
    """BS-YOLO: the one-launch MSCAAttention spatial part (nine depthwise convs + four global means out of LDS) returns
    exactly what the thirteen separate launches return -- on 3 x 5, 20 x 20, 2 x 2 and 40 x 40 attention maps."""
    m = R.Model("bsyolo11", "n", 12, "detect")
    P = R.synth_params(m, 3)
    cfg = stock_cfg("bsyolo11", "n", 12)
    fused = YoloEngine(cfg, P, fuse_msca=True, autotune=False)
    plain = YoloEngine(cfg, P, fuse_msca=False, autotune=False)
    B, H, W = shape
    pf, _ = fused.plan_for(B, H, W, torch.float16, torch.float16)
    pp, _ = plain.plan_for(B, H, W, torch.float16, torch.float16)
    assert sum(o["kind"] == L.OP_MSCA_SPATIAL for o in pf.ops) == 1 and len(pp.ops) - len(pf.ops) == 12
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(H + W)).half().to(DEV)
    yf, rf = fused(x)
    yp, rp = plain(x)
    assert torch.equal(yf, yp)
    for a, b in zip(rf, rp):
        assert torch.equal(a, b)
    fused.close()
    plain.close()


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_engine_fused_decoder_matches_decode_kernel(dtype):
    """Detect._inference (head.py:113-148) fused into the last conv of every head branch (conv_epilogue_head) against the
    separate decode kernel on f32 logit maps: same y and same raw maps, with and without the raw maps requested."""
    m = R.Model("yolo11", "s", 80, "detect")
    P = R.synth_params(m, 0)
    cfg = stock_cfg("yolo11", "s")
    fused = YoloEngine(cfg, P, fuse_head=True, autotune=False)
    plain = YoloEngine(cfg, P, fuse_head=False, autotune=False)
    x = torch.rand(2, 3, 96, 160, generator=torch.Generator().manual_seed(5)).to(dtype).to(DEV)
    pf, _ = fused.plan_for(2, 96, 160, dtype, dtype)
    pp, _ = plain.plan_for(2, 96, 160, dtype, dtype)
    assert not any(o["kind"] in (L.OP_DECODE, L.OP_RAW_NCHW) for o in pf.ops)
    assert sum(1 for o in pf.ops if o["kind"] == L.OP_CONV and o.get("out_f32", 0) >= 2) == 6
    assert any(o["kind"] == L.OP_DECODE for o in pp.ops)
    yf, rf = fused(x)
    yp, rp = plain(x)
    yn, rn = fused(x, want_raw=False)
    torch.cuda.synchronize()
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=1e-3, atol=1e-3)
    # scores: identical formula on identical logits.  Boxes: fp32 DFL with a different summation order -- and, since the box branch's
    # last 3x3 conv runs inside the tail launch (patch kernel, chunk-major K walk) but as an implicit-GEMM launch in the plain plan
    # (maps under 16 pixels here), a few 1-ulp flips of its fp16 map: 15 of 52 920 coordinates moved by up to 0.015 px
    np.testing.assert_allclose(yf.float().cpu().numpy()[:, 4:], yp.float().cpu().numpy()[:, 4:], **tol)
    np.testing.assert_allclose(yf.float().cpu().numpy()[:, :4], yp.float().cpu().numpy()[:, :4], rtol=1e-3, atol=5e-2)
    assert torch.equal(yn, yf) and all(r is None for r in rn)
    for a, b in zip(rf, rp):
        assert a.shape == b.shape
        np.testing.assert_allclose(a.float().cpu().numpy(), b.float().cpu().numpy(), rtol=1e-3, atol=1e-3)
    fused.close()
    plain.close()


def test_val_match_golden_and_random():
    """Validator matching on the device (csrc/val_match.hip) == the reference's own match_predictions on its own box_iou
    (val_match.npz), case by case and as one padded batch; then random batches against the oracle restatement."""
    from bs_yolo_amd import val as HV
    from oracle import val_ref as V
    z = np.load(GOLDEN / "val_match.npz", allow_pickle=False)
    iouv = torch.linspace(0.5, 0.95, 10)
    cases = json.loads(str(z["cases"]))
    for ci in cases:
        det, lab, lcls = (torch.from_numpy(z[f"c{ci}.{k}"]) for k in ("det", "lab", "lcls"))
        got = HV.process_batch(det.to(DEV), lab.to(DEV), lcls.to(DEV), iouv)
        assert got.shape == (det.shape[0], 10) and (got.cpu().numpy() == z[f"c{ci}.correct"]).all(), ci
    # all cases as ONE batch (max_det 300, ragged counts)
    B, md = len(cases), 300
    Lmax = max(z[f"c{ci}.lab"].shape[0] for ci in cases)
    detb, cnt = torch.zeros(B, md, 6), torch.zeros(B, dtype=torch.int32)
    gb, gc, gn = torch.zeros(B, Lmax, 4), torch.zeros(B, Lmax), torch.zeros(B, dtype=torch.int32)
    for i, ci in enumerate(cases):
        d, l = z[f"c{ci}.det"], z[f"c{ci}.lab"]
        detb[i, :len(d)] = torch.from_numpy(d); cnt[i] = len(d)
        gb[i, :len(l)] = torch.from_numpy(l); gc[i, :len(l)] = torch.from_numpy(z[f"c{ci}.lcls"]); gn[i] = len(l)
    tp = HV.match_batched(detb.to(DEV), cnt.to(DEV), gb.to(DEV), gc.to(DEV), gn.to(DEV), iouv).cpu().numpy()
    for i, ci in enumerate(cases):
        n = int(cnt[i])
        assert (tp[i, :n] == z[f"c{ci}.correct"]).all() and not tp[i, n:].any(), ci
    # random batches vs the oracle (continuous boxes: no exact IoU ties)
    rng = np.random.default_rng(5)
    for _ in range(3):
        B, md, Lmax = 6, 512, 40
        detb = torch.zeros(B, md, 6); cnt = torch.from_numpy(rng.integers(0, md + 1, B).astype(np.int32))
        gn = torch.from_numpy(rng.integers(0, Lmax + 1, B).astype(np.int32))
        xy = rng.uniform(0, 500, (B, Lmax, 2)); wh = rng.uniform(10, 100, (B, Lmax, 2))
        gb = torch.from_numpy(np.concatenate([xy, xy + wh], 2).astype(np.float32))
        gc = torch.from_numpy(rng.integers(0, 4, (B, Lmax)).astype(np.float32))
        src = rng.integers(0, Lmax, (B, md))
        boxes = np.take_along_axis(gb.numpy(), src[..., None], 1) + rng.normal(0, 5.0, (B, md, 4)).astype(np.float32)
        detb[..., :4] = torch.from_numpy(boxes.astype(np.float32))
        detb[..., 4] = torch.from_numpy(rng.uniform(0.25, 1, (B, md)).astype(np.float32))
        detb[..., 5] = torch.from_numpy(rng.integers(0, 4, (B, md)).astype(np.float32))
        tp = HV.match_batched(detb.to(DEV), cnt.to(DEV), gb.to(DEV), gc.to(DEV), gn.to(DEV), iouv).cpu().numpy()
        for b in range(B):
            n, m = int(cnt[b]), int(gn[b])
            ref = V.process_batch(detb[b, :n].numpy(), gb[b, :m].numpy(), gc[b, :m].numpy(), iouv.numpy()) if n and m else \
                np.zeros((n, 10), bool)
            assert (tp[b, :n] == ref).all() and not tp[b, n:].any()


# ------------------------------------------------------------------------------------------------------------
# NMS: bit-exact against the oracle (and the reference's golden outputs)
# ------------------------------------------------------------------------------------------------------------
def test_ap_per_class_golden_and_large():
    """bs_yolo_amd.val.ap_per_class (csrc/val_ap.hip, float64) against the REFERENCE's own ap_per_class outputs
    (tests/golden/ap_per_class.npz) -- every one of the 12 returned arrays within 1e-12 -- and against the oracle on
    200 000 detections of 80 classes (class segments longer than any workgroup pass, confidence ties included)."""
    from bs_yolo_amd import val as HV
    from oracle import val_ref as V
    z = np.load(GOLDEN / "ap_per_class.npz")
    names = ["tp", "fp", "p", "r", "f1", "ap", "unique_classes", "p_curve", "r_curve", "f1_curve", "x", "prec_values"]
    for ci in json.loads(str(z["cases"])):
        got = HV.ap_per_class(z[f"c{ci}.tp"], z[f"c{ci}.conf"], z[f"c{ci}.pred_cls"], z[f"c{ci}.target_cls"], device=DEV)
        for k, g in zip(names, got):
            want = z[f"c{ci}.out.{k}"]
            assert np.asarray(g).shape == want.shape, (ci, k, np.asarray(g).shape, want.shape)
            np.testing.assert_allclose(np.asarray(g, dtype=np.float64), want.astype(np.float64), rtol=0, atol=1e-12, err_msg=f"case {ci} {k}")
    rng = np.random.default_rng(6)
    n, m, ncls = 200_000, 30_000, 80
    conf = np.round(rng.uniform(0.001, 1.0, n), 4).astype(np.float32)      # ~20 detections share every confidence value
    pred_cls = rng.integers(0, ncls, n).astype(np.float32)
    target_cls = rng.integers(0, ncls, m).astype(np.float32)
    tp = np.logical_and.accumulate(np.stack([rng.random(n) < 0.1 * conf * (1.0 - 0.07 * j) for j in range(10)], 1), 1)
    want = V.ap_per_class(tp, conf, pred_cls, target_cls)
    got = HV.ap_per_class(torch.from_numpy(tp).to(DEV), torch.from_numpy(conf).to(DEV), torch.from_numpy(pred_cls).to(DEV), target_cls, device=DEV)
    for k, g, w in zip(names, got, want):
        np.testing.assert_allclose(np.asarray(g, dtype=np.float64), np.asarray(w, dtype=np.float64), rtol=0, atol=1e-12, err_msg=k)
    # no detections / no labels
    e = HV.ap_per_class(np.zeros((0, 10), bool), np.zeros(0, np.float32), np.zeros(0, np.float32), np.array([1.0, 1.0, 3.0]), device=DEV)
    assert e[5].shape == (2, 10) and not e[5].any() and e[11].shape == (0, 1000) and list(e[6]) == [1, 3]


def _nms_compare(pred, kw):
    ref_in = pred.clone()
    ref = PP.non_max_suppression(ref_in, **kw)
    dpred = pred.clone().to(DEV)
    got = HN.non_max_suppression(dpred, **kw)
    torch.cuda.synchronize()
    assert len(got) == len(ref)
    if kw.get("in_place", True):
        assert torch.equal(dpred[:, :4].cpu(), ref_in[:, :4])  # same in-place xywh->xyxy mutation
    for b, (g_, r_) in enumerate(zip(got, ref)):
        assert tuple(g_.shape) == tuple(r_.shape), (b, g_.shape, r_.shape)
        assert torch.equal(g_.cpu(), r_), f"image {b}"


def test_nms_golden_cases():
    z = np.load(GOLDEN / "nms.npz")
    for tag, kw, n_out in json.loads(str(z["cases"])):
        pred = torch.from_numpy(z[tag + ".pred"].copy())
        dpred = pred.clone().to(DEV)
        got = HN.non_max_suppression(dpred, **kw)
        torch.cuda.synchronize()
        assert len(got) == n_out
        assert np.array_equal(dpred[:, :4].cpu().numpy(), z[tag + ".pred_after"]), tag
        for i, r in enumerate(got):
            exp = z[f"{tag}.out{i}"]
            assert tuple(r.shape) == exp.shape, (tag, i, r.shape, exp.shape)
            assert np.array_equal(r.cpu().numpy(), exp), f"{tag}[{i}]"


def _rand_pred(B, nc, A, nm, seed, dup=True):
    g = torch.Generator().manual_seed(seed)
    base = max(A // 6, 1)
    c = torch.rand(B, 2, base, generator=g) * 600 + 20
    wh = torch.rand(B, 2, base, generator=g) * 120 + 8
    rep = (A + base - 1) // base
    box = torch.cat((c, wh), 1).repeat(1, 1, rep)[:, :, :A] + torch.randn(B, 4, A, generator=g) * (3.0 if dup else 50.0)
    box[:, 2:] = box[:, 2:].abs() + 1
    cls = torch.rand(B, nc, A, generator=g) ** 4
    parts = [box, cls] + ([torch.randn(B, nm, A, generator=g)] if nm else [])
    return torch.cat(parts, 1)


@pytest.mark.parametrize("kw", [
    dict(conf_thres=0.25, iou_thres=0.7),
    dict(conf_thres=0.001, iou_thres=0.7, multi_label=True),
    dict(conf_thres=0.3, iou_thres=0.45, agnostic=True, max_det=17),
    dict(conf_thres=0.2, iou_thres=0.5, classes=[0, 2], in_place=False),
])
def test_nms_random_bit_exact(kw):
    _nms_compare(_rand_pred(3, 12, 3000, 0, 21), kw)


def test_nms_edge_cases():
    _nms_compare(_rand_pred(2, 6, 1, 0, 1), dict(conf_thres=0.0, iou_thres=0.5))        # a single anchor
    _nms_compare(_rand_pred(2, 6, 257, 0, 2), dict(conf_thres=0.99999, iou_thres=0.5))  # nothing passes
    p = _rand_pred(1, 3, 600, 0, 3)
    p[:, 4:] = 0.5                                                                      # every score ties
    _nms_compare(p, dict(conf_thres=0.25, iou_thres=0.6))
    p = _rand_pred(1, 2, 900, 0, 4)
    p[:, :4] = torch.tensor([100.0, 100.0, 50.0, 50.0]).view(1, 4, 1)                   # all boxes identical
    _nms_compare(p, dict(conf_thres=0.01, iou_thres=0.5, multi_label=True))
    _nms_compare(_rand_pred(2, 4, 700, 32, 5), dict(conf_thres=0.3, iou_thres=0.7, nc=4))  # mask coefficients ride along
    with pytest.raises(AssertionError, match="Invalid Confidence"):
        HN.non_max_suppression(torch.zeros(1, 6, 8, device=DEV), conf_thres=1.5)


def test_nms_large_candidate_set_and_cap():
    """> 8192 candidates (global-memory sort path) and the max_nms cap (ops.py:285-286)."""
    pred = _rand_pred(2, 10, 8400, 0, 9, dup=False)
    _nms_compare(pred, dict(conf_thres=0.001, iou_thres=0.7, multi_label=True, max_nms=30000))
    _nms_compare(pred, dict(conf_thres=0.001, iou_thres=0.7, multi_label=True, max_nms=5000))


def test_nms_fp16_prediction():
    """fp16 predictions: boxes are converted in fp16 (as the reference's in-place op does), decisions in fp32."""
    pred = _rand_pred(2, 12, 2000, 0, 13).half()
    kw = dict(conf_thres=0.25, iou_thres=0.7)
    dpred = pred.clone().to(DEV)
    got = HN.non_max_suppression(dpred, **kw)
    torch.cuda.synchronize()
    ref_in = pred.clone()
    tp = ref_in.transpose(-1, -2)
    tp[..., :4] = PP.xywh2xyxy(tp[..., :4])  # fp16 arithmetic on the CPU
    assert torch.equal(dpred[:, :4].cpu(), ref_in[:, :4])
    ref = PP.non_max_suppression(ref_in.float(), boxes_xyxy=True, **kw)
    for g_, r_ in zip(got, ref):
        assert g_.dtype == torch.float16
        assert torch.equal(g_.float().cpu(), r_.half().float())


def test_nms_properties_at_full_size():
    """BASELINE config sizes (B=64, nc=80, A=8400): size-independent properties instead of an O(n^2) CPU run."""
    pred = _rand_pred(64, 80, 8400, 0, 17)
    pred[:, 4:] = pred[:, 4:] ** 3  # ~1-2 % of anchors above 0.25
    det, counts = HN.nms_batched(pred.to(DEV), 0.25, 0.7)
    det2, counts2 = HN.nms_batched(pred.to(DEV), 0.25, 0.7)
    torch.cuda.synchronize()
    assert torch.equal(det, det2) and torch.equal(counts, counts2)  # deterministic
    det, counts = det.cpu(), counts.cpu()
    assert int(counts.max()) <= 300 and int(counts.min()) >= 0
    for b in range(0, 64, 7):
        n = int(counts[b])
        d = det[b, :n]
        assert torch.all(d[:-1, 4] >= d[1:, 4])            # sorted by confidence
        assert torch.all(d[:, 4] > 0.25)
        assert torch.all(det[b, n:] == 0)                   # padding rows are zero
        off = d[:, :4] + d[:, 5:6] * 7680
        lt = torch.max(off[:, None, :2], off[None, :, :2])
        rb = torch.min(off[:, None, 2:], off[None, :, 2:])
        inter = (rb - lt).clamp(min=0).prod(-1)
        area = (off[:, 2] - off[:, 0]) * (off[:, 3] - off[:, 1])
        iou = inter / (area[:, None] + area[None] - inter)
        iou.fill_diagonal_(0)
        assert float(iou.max()) <= 0.7                      # survivors of one class do not overlap above the threshold
        # idempotence: NMS of the survivors keeps all of them
        again = PP.greedy_nms(off, d[:, 4], 0.7)
        assert again.tolist() == list(range(n))


@pytest.mark.parametrize("upsample", [False, True])
@pytest.mark.parametrize("pdtype", [torch.float32, torch.float16])
def test_process_mask_matches_oracle(upsample, pdtype):
    """process_mask / crop_mask (ops.py:663-694): binary masks equal the oracle's except where the pre-threshold
    value is within fp32 summation noise of zero (different dot-product order than torch.matmul)."""
    from bs_yolo_amd import masks as HM
    g = torch.Generator().manual_seed(8)
    nm, mh, mw, ih, iw, n = 32, 40, 48, 160, 192, 13
    protos = torch.randn(nm, mh, mw, generator=g).to(pdtype)
    coef = torch.randn(n, nm, generator=g)
    xy = torch.rand(n, 2, generator=g) * torch.tensor([iw * 0.6, ih * 0.6])
    wh = torch.rand(n, 2, generator=g) * torch.tensor([iw * 0.4, ih * 0.4]) + 4
    boxes = torch.cat((xy, xy + wh), 1)
    ref = PP.process_mask(protos.float(), coef, boxes, (ih, iw), upsample)
    got = HM.process_mask(protos.to(DEV), coef.to(DEV), boxes.to(DEV), (ih, iw), upsample)
    torch.cuda.synchronize()
    assert got.shape == ref.shape and got.dtype == torch.float32
    mism = (got.cpu() != ref).float().mean().item()
    assert mism < 2e-4, mism
    # empty input: (0, h, w)
    e = HM.process_mask(protos.to(DEV), coef[:0].to(DEV), boxes[:0].to(DEV), (ih, iw), upsample)
    assert e.shape[0] == 0


def test_scale_boxes_matches_oracle():
    g = torch.Generator().manual_seed(2)
    B, max_det = 3, 20
    det = torch.rand(B, max_det, 6, generator=g) * 700 - 30
    counts = torch.tensor([20, 7, 0], dtype=torch.int32)
    shapes = [(1080, 810), (720, 1280), (333, 500)]
    ref = det.clone()
    for b in range(B):
        n = int(counts[b])
        ref[b, :n, :4] = PP.scale_boxes((640, 640), ref[b, :n, :4].clone(), shapes[b])
    d = det.clone().to(DEV)
    HN.scale_boxes_batched(d, counts.to(DEV), (640, 640), shapes)
    torch.cuda.synchronize()
    np.testing.assert_allclose(d.cpu().numpy(), ref.numpy(), rtol=1e-6, atol=1e-4)


# ------------------------------------------------------------------------------------------------------------
# letterbox: bit-exact pixels against the oracle's OpenCV restatement
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shapes,imgsz", [
    ([(100, 37), (100, 37)], (160, 160)),          # equal shapes -> auto (minimal rectangle)
    ([(75, 120), (64, 64), (256, 192)], (128, 128)),  # ragged -> pad to imgsz; 64->128 upscale; exact-2x downscale
    ([(333, 500)], (160, 224)),
    ([(1080, 810)], (640, 640)),                   # bus.jpg geometry (BASELINE config 1)
    ([(720, 1280), (720, 1280)], (640, 640)),
])
@pytest.mark.parametrize("half", [False, True])
def test_letterbox_bit_exact(shapes, imgsz, half):
    rng = np.random.default_rng(4)
    ims = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    ref = LB.preprocess(ims, imgsz, half=half)
    got = HLB.preprocess(ims, imgsz, half=half, device=DEV)
    torch.cuda.synchronize()
    assert tuple(got.shape) == tuple(ref.shape)
    assert torch.equal(got.cpu(), ref)


def test_letterbox_golden_pixels():
    z = np.load(GOLDEN / "letterbox.npz")
    for k, c in enumerate(json.loads(str(z["pixel_cases"]))):
        kw = dict(c["kw"])
        new_shape = tuple(kw.pop("new_shape"))
        img = z[f"img{k}"]
        lb = HLB.LetterBox(new_shape, **kw)
        H2, W2, nw, nh, left, top, _ = lb.geometry(img.shape[:2])
        assert [H2, W2, 3] == list(z[f"lb{k}"].shape)
        import ctypes as C
        d = torch.from_numpy(img).to(DEV)
        ptrs = torch.tensor([d.data_ptr()], dtype=torch.int64, device=DEV)
        hw = torch.tensor([[img.shape[0], img.shape[1]]], dtype=torch.int32, device=DEV)
        geom = torch.tensor([[nw, nh, left, top]], dtype=torch.int32, device=DEV)
        out = torch.empty((1, 3, H2, W2), dtype=torch.float32, device=DEV)
        L.check(L.lib.bsy_letterbox(C.c_void_p(ptrs.data_ptr()), C.c_void_p(hw.data_ptr()), C.c_void_p(geom.data_ptr()),
                                    1, H2, W2, C.c_void_p(out.data_ptr()), L.BSY_F32,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        exp = torch.from_numpy(np.ascontiguousarray(z[f"lb{k}"][..., ::-1].transpose(2, 0, 1))).float() / 255
        assert torch.equal(out[0].cpu(), exp)


# ------------------------------------------------------------------------------------------------------------
# per-layer parity: every top-level layer output of the engine against the REFERENCE's own layer
# outputs (fixtures layer0_i, generated by tests/golden/make_fixtures.py from the imported reference) and against the
# oracle run with the engine's storage precision.  An end-to-end bound cannot localise a bad kernel; this one does.
# ------------------------------------------------------------------------------------------------------------
# Measured (r02, two boxes, tuned and heuristic kernel configurations): per layer, as shares of the layer's range, max error
# 0.4-2.9e-3 (stock) / 0.4-2.8e-3 (BS-YOLO), 99.9th percentile <= 1.8e-3, mean <= 1.8e-4 -- against BOTH references: one f16
# ulp of a value near the top of the range is 1e-3 of it, and the emulation does not share the kernels' summation order, so
# "vs emulation" is tighter than "vs fp32 reference" only in the early layers and in the mean.  A kernel bug shows as a layer
# whose error jumps by an order of magnitude against its predecessor; bounds (max, p99.9, mean) sit ~1.5x above the measurements.
LAYER_TOL = {"yolo11n_detect": (4e-3, 2.5e-3, 2.5e-4), "bsyolo11n_detect": (5e-3, 3e-3, 3e-4), "yolov5n_detect": (4e-3, 2.5e-3, 2.5e-4)}


@pytest.mark.parametrize("tag", ["yolo11n_detect", "bsyolo11n_detect", "yolov5n_detect"])
@pytest.mark.parametrize("fused", [True, False])
def test_engine_every_layer_matches_reference(tag, fused, monkeypatch):
    monkeypatch.setenv("BSY_ARENA_REUSE", "0")  # keep every layer's buffer alive until the end of the forward
    z = np.load(GOLDEN / f"graph_{tag}.npz")
    meta = json.loads(str(z["meta"]))
    m = R.Model(meta["family"], meta["scale"], meta["nc"], meta["task"])
    P = R.synth_params(m, meta["seed"])
    kw = {} if fused else dict(fuse_stem=False, fuse_bneck=False, fuse_head=False, fuse_dwpw=False, merge_c3k=False, fuse_msca=False,
                               fuse_tail=False, fuse_pmsfa=False)
    eng = YoloEngine(stock_cfg(meta["family"], meta["scale"], meta["nc"], meta["task"]), P, **kw)
    x = torch.from_numpy(z["x0"])
    eng(x.half().to(DEV))
    torch.cuda.synchronize()
    plan, h = eng.plan_for(x.shape[0], x.shape[2], x.shape[3], torch.float16, torch.float16)
    R.FP16_EMULATION = True
    try:
        with torch.inference_mode():
            _, emu = m.forward(P, x.half().float(), return_layers=True)
    finally:
        R.FP16_EMULATION = False
    rows, checked = [], 0
    for i, t in enumerate(plan.layer_out):
        if t is None:  # fused away (layer 0 inside the stem launch) or the head
            continue
        got = (torch.cat([eng.read_view(plan, h, v) for v in t], 1) if isinstance(t, list) else eng.read_view(plan, h, t)).numpy()
        ref, q = z[f"layer0_{i}"], emu[i].numpy()
        assert got.shape == ref.shape == q.shape, (i, got.shape, ref.shape)
        scale = np.abs(ref).max()
        eq, er = np.abs(got - q).ravel() / scale, np.abs(got - ref).ravel() / scale
        rows.append((i, eq.max(), np.quantile(eq, 0.999), eq.mean(), er.max(), np.quantile(er, 0.999), er.mean()))
        checked += 1
    eng.close()
    assert checked >= len(plan.layer_out) - 3
    tmax, tq, tmean = LAYER_TOL[tag]
    bad = [r for r in rows if r[1] > tmax or r[4] > tmax or r[2] > tq or r[5] > tq or r[3] > tmean or r[6] > tmean]
    assert not bad, "layer: max / p99.9 / mean error vs fp16-emulating oracle | vs reference (shares of the layer's range): " + \
        "; ".join(f"{r[0]}: {r[1]:.1e} {r[2]:.1e} {r[3]:.1e} | {r[4]:.1e} {r[5]:.1e} {r[6]:.1e}" for r in rows)
    print("per-layer worst: max %.2e p99.9 %.2e mean %.2e (vs emulation), max %.2e p99.9 %.2e mean %.2e (vs reference)" %
          tuple(max(r[k] for r in rows) for k in range(1, 7)))


def test_engine_memory_and_tuning_stay_bounded_over_rect_shapes():
    """val runs rect=True by default (engine/model.py:635: per-batch shapes, data/base.py:261-284) and predict's `auto`
    letterbox varies (H, W) too.  40 distinct shapes through ONE engine: the activation arena stays at the size of the
    largest plan (round 1 kept a private 3.8 GB workspace per shape), old plans are evicted, no conv shape is timed twice,
    and results do not depend on which plans came before."""
    cfg = stock_cfg("yolo11", "n")
    m = R.Model("yolo11", "n", 80, "detect")
    P = R.synth_params(m, 0)
    eng = YoloEngine(cfg, P, max_plans=4)
    shapes = [(4, 32 * h, 32 * w) for h in range(6, 14) for w in range(8, 13)]  # 192..416 x 256..384
    assert len(set(shapes)) == 40
    free0, _ = torch.cuda.mem_get_info()
    g = torch.Generator().manual_seed(0)
    first = {}
    largest = 0
    for (B, H, W) in shapes:
        x = torch.rand(B, 3, H, W, generator=g).half().to(DEV)
        y, _ = eng(x)
        first[(B, H, W)] = (x, y.clone())
        plan, _ = eng.plan_for(B, H, W, torch.float16, torch.float16)
        largest = max(largest, plan.arena_bytes)
    torch.cuda.synchronize()
    assert len(eng._plans) == 4
    assert eng.arena_bytes <= largest + 4096, (eng.arena_bytes, largest)
    # device memory actually taken by the library (hipMalloc, not torch's allocator): arena + weights + slack
    torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info()
    held = sum(x.numel() * 2 + y.numel() * 2 for x, y in first.values())
    assert free0 - free1 < 1.5 * largest + held + (64 << 20), (free0 - free1, largest, held)
    st = dict(eng.tune_stats)
    assert st["autotune_calls"] == 40
    # revisit every shape (36 of them were evicted): nothing is timed again, outputs are bit-identical to the first visit
    for (B, H, W), (x, y0) in first.items():
        y, _ = eng(x)
        assert torch.equal(y, y0), (B, H, W)
    assert eng.tune_stats["timed_ops"] == st["timed_ops"]
    # the family cache (same layer, pixel count within 2x) answers most rect shapes without timing
    assert st["family_ops"] > 10 * st["timed_ops"] / 40, st
    eng.close()


def test_engine_drops_presets_that_do_not_fit_the_shape(tmp_path, monkeypatch):
    """configuration ids that reach a plan from the tune file (or the similar-size family cache) and cannot run the
    op's shape are removed from the caches and timed afresh -- they must not be persisted.  A patch-kernel id (3x3 only) is written
    for every conv shape of a YOLO11n forward; the 1x1 / stride-2 layers cannot take it."""
    import json
    cfg = stock_cfg("yolo11", "n", 80, "detect")
    P = R.synth_params(R.Model("yolo11", "n", 80, "detect"), 0)
    x = torch.rand(2, 3, 96, 128, generator=torch.Generator().manual_seed(3)).half().to(DEV)
    ref = YoloEngine(cfg, P, autotune=False)
    y_ref, _ = ref(x)
    plan, _ = ref.plan_for(2, 96, 128, torch.float16, torch.float16)
    sigs = sorted({sg for sg in plan.conv_sigs if sg is not None})
    ref.close()
    f = tmp_path / "tune.json"
    json.dump([[list(sg), 0xb2] for sg in sigs], open(f, "w"))
    monkeypatch.setenv("BSY_TUNE_CACHE", str(f))
    eng = YoloEngine(cfg, P)
    y, _ = eng(x)
    assert torch.equal(y, y_ref)
    bad = eng.tune_stats.get("invalid_presets", 0)
    assert 0 < bad, eng.tune_stats
    assert eng.tune_stats["timed_ops"] >= 1
    plan, h = eng.plan_for(2, 96, 128, torch.float16, torch.float16)
    import ctypes as C
    from bs_yolo_amd import lib as L
    ext, n = eng._ext(x, y, [None, None, None], None)
    valid = (C.c_int32 * len(plan.ops))()
    L.check(L.lib.bsy_plan_check_tuning(h, ext, n, valid, len(plan.ops)))
    assert 0 not in list(valid) and 1 in list(valid)
    stored = {tuple(k): v for k, v in json.load(open(f))}
    kept = [sg for sg in sigs if stored.get(sg) == 0xb2]
    assert len(kept) < len(sigs)  # the ids that did not fit were replaced by timed winners in the file too
    eng.close()


def test_engine_splits_batches_by_the_largest_view():
    """the automatic batch split must size by the largest activation view (YOLOv8 C2f concat buffers), not by the
    first conv's output."""
    cfg = stock_cfg("yolov8", "n", 80, "detect")
    m = R.Model("yolov8", "n", 80, "detect")
    eng = YoloEngine(cfg, R.synth_params(m, 0), autotune=False)
    per_img = eng._max_view_elems(640, 640)
    p1 = eng.plan_for(1, 640, 640, torch.float16, torch.float16)[0]
    widest = max((t.H * t.W * t.ld) for o in p1.ops for t in [o.get("dst")] if t is not None and t.buf < 0x100000)
    assert per_img >= widest
    eng.close()


# ------------------------------------------------------------------------------------------------------------
# fp32 correctness mode (csrc/ref32.hip): the north-star's tolerance, asserted as stated -- |dscore| <= 1e-3 and
# |dbox| <= 1e-3 * imgsz against the REFERENCE's own fp32 outputs, on all seven golden graphs and every golden input.
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["fp32", "fp32x"])
@pytest.mark.parametrize("tag", ["yolo11n_detect", "yolo11s_detect", "yolo11m_detect", "yolo11n_segment",
                                 "yolov8n_segment", "bsyolo11n_detect", "bsyolo11s_detect", "yolov5n_detect", "yolov5s_detect"])
def test_engine_fp32_mode_meets_the_north_star_tolerance(tag, precision):
    """Both fp32-storage modes -- "fp32" (exact fp32 arithmetic) and "fp32x" (round 4: dense convs on the fp16 matrix pipe with
    split-f16 operands) -- against the REFERENCE's own fp32 outputs at the north-star's tolerance, as stated."""
    z = np.load(GOLDEN / f"graph_{tag}.npz")
    meta = json.loads(str(z["meta"]))
    m = R.Model(meta["family"], meta["scale"], meta["nc"], meta["task"])
    P = R.synth_params(m, meta["seed"])
    eng = YoloEngine(stock_cfg(meta["family"], meta["scale"], meta["nc"], meta["task"]), P, precision=precision)
    nc, si = meta["nc"], 0
    while f"x{si}" in z:
        x = torch.from_numpy(z[f"x{si}"])
        y, aux = eng(x.to(DEV))
        torch.cuda.synchronize()
        y, yref = y.cpu().numpy(), z[f"y{si}"]
        imgsz = max(x.shape[2], x.shape[3])
        es, eb = np.abs(y[:, 4:4 + nc] - yref[:, 4:4 + nc]).max(), np.abs(y[:, :4] - yref[:, :4]).max()
        assert es <= 1e-3, (tag, si, "score", es)
        assert eb <= 1e-3 * imgsz, (tag, si, "box", eb, imgsz)
        raws = aux[0] if meta["task"] == "segment" else aux
        for l, r in enumerate(raws):
            rr = z[f"raw{si}_{l}"]
            assert np.abs(r.cpu().numpy() - rr).max() <= 1e-3 * max(1.0, np.abs(rr).max()), (tag, si, "raw", l)
        if meta["task"] == "segment":
            mc, proto = aux[1], aux[2]
            em = np.abs(y[:, 4 + nc:] - yref[:, 4 + nc:]).max()
            assert em <= 1e-3 * max(1.0, np.abs(yref[:, 4 + nc:]).max()), (tag, si, "mask coefficients", em)
            pr = z[f"proto{si}"]
            assert np.abs(proto.cpu().numpy() - pr).max() <= 1e-3 * max(1.0, np.abs(pr).max()), (tag, si, "proto")
        si += 1
    assert si >= 1
    # per-layer, where the fixtures hold the reference's layer outputs: fp32 storage tracks them to 1e-4 of the range
    if "layer0_0" in z:
        x = torch.from_numpy(z["x0"])
        plan, h = eng.plan_for(x.shape[0], x.shape[2], x.shape[3], torch.float32, torch.float32)
        if not eng.reuse:
            for i, t in enumerate(plan.layer_out):
                if t is None:
                    continue
                got = (torch.cat([eng.read_view(plan, h, v) for v in t], 1) if isinstance(t, list) else eng.read_view(plan, h, t)).numpy()
                ref = z[f"layer0_{i}"]
                assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), (tag, "layer", i)
    eng.close()


@pytest.mark.parametrize("precision", ["fp32", "fp32x"])
def test_engine_fp32_mode_layers_match_reference(monkeypatch, precision):
    """The same per layer (buffers kept alive: BSY_ARENA_REUSE=0): every top-level layer of the fp32-storage modes against the
    reference's own layer outputs, 1e-4 of the layer's range (fp32x: the image conv, the patch / implicit-GEMM split-f16 kernels and
    every non-conv kernel in place)."""
    monkeypatch.setenv("BSY_ARENA_REUSE", "0")
    for tag in ("yolo11n_detect", "bsyolo11n_detect", "yolov5n_detect"):
        z = np.load(GOLDEN / f"graph_{tag}.npz")
        meta = json.loads(str(z["meta"]))
        m = R.Model(meta["family"], meta["scale"], meta["nc"], meta["task"])
        eng = YoloEngine(stock_cfg(meta["family"], meta["scale"], meta["nc"], meta["task"]), R.synth_params(m, meta["seed"]), precision=precision)
        x = torch.from_numpy(z["x0"])
        eng(x.to(DEV))
        torch.cuda.synchronize()
        plan, h = eng.plan_for(x.shape[0], x.shape[2], x.shape[3], torch.float32, torch.float32)
        n = 0
        for i, t in enumerate(plan.layer_out):
            if t is None:
                continue
            got = (torch.cat([eng.read_view(plan, h, v) for v in t], 1) if isinstance(t, list) else eng.read_view(plan, h, t)).numpy()
            ref = z[f"layer0_{i}"]
            assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), (tag, i, np.abs(got - ref).max(), np.abs(ref).max())
            n += 1
        assert n >= 22
        eng.close()


@pytest.mark.parametrize("precision", ["fp32", "fp32x"])
@pytest.mark.parametrize("scale", ["s", "m"])
def test_engine_fp32_modes_match_the_oracle_at_640(scale, precision):
    """the fp32 modes were pinned to the reference only on <= 96 x 160 inputs, while the tiles, occupancy and partial
    rounds of the real workload occur at 640 x 640.  Here: YOLO11s / YOLO11m, two 640 x 640 images, both fp32-storage modes against
    the CPU oracle (itself pinned by the reference's golden vectors, tests/test_oracle_golden.py) at the north-star's tolerance:
    |dscore| <= 1e-3, |dbox| <= 1e-3 * imgsz, raw maps <= 1e-3 of their range (measured: ~1e-5)."""
    m = R.Model("yolo11", scale, 80, "detect")
    P = R.synth_params(m, 5)
    x = torch.rand(2, 3, 640, 640, generator=torch.Generator().manual_seed(640))
    with torch.inference_mode():
        yref, rawref = m.forward(P, x)
    eng = YoloEngine(stock_cfg("yolo11", scale), P, precision=precision)
    y, raws = eng(x.to(DEV))
    torch.cuda.synchronize()
    y = y.cpu()
    es, eb = float((y[:, 4:] - yref[:, 4:]).abs().max()), float((y[:, :4] - yref[:, :4]).abs().max())
    assert tuple(y.shape) == (2, 84, 8400) and es <= 1e-3 and eb <= 1e-3 * 640, (scale, precision, es, eb)
    assert es <= 2e-4 and eb <= 2e-4 * 640, (scale, precision, es, eb)  # what the modes actually hold, with margin
    for r, rr in zip(raws, rawref):
        assert float((r.cpu() - rr).abs().max()) <= 1e-3 * max(1.0, float(rr.abs().max()))
    eng.close()


@pytest.mark.parametrize("precision", ["fp32", "fp32x", "fp16"])
def test_config1_on_the_reference_image(precision):
    """BASELINE config 1 on the reference's own test image (ultralytics/assets/bus.jpg; pixels decoded once in the build container,
    tests/golden/config1_bus.npz, every expected value produced by RUNNING THE REFERENCE: LetterBox -> YOLO11n -> non_max_suppression
    -> scale_boxes, make_fixtures.py config1_fixture).  HIP letterbox: the reference's letterboxed pixels bit for bit (crc32);
    forward: y at the reference's 1000 highest-scoring anchors; HIP NMS + scale_boxes: the reference's detections one for one in the
    fp32 modes (same count, same classes, boxes to 1e-3 * imgsz, scores to 1e-3); the fp16 path: its documented statistics."""
    import zlib
    z = np.load(GOLDEN / "config1_bus.npz")
    meta = json.loads(str(z["meta"]))
    bgr = z["bgr"]
    m = R.Model("yolo11", "n", 80, "detect")
    P = R.synth_params(m, meta["seed"])
    for k in P:
        if ".cv3." in k and k.endswith(".2.bias"):
            P[k] = P[k] + meta["cls_shift"]
    half = precision == "fp16"
    xd = HLB.preprocess([bgr], (640, 640), half=half, pt=True, stride=32, device=DEV)
    assert tuple(xd.shape) == (1, 3, 640, 480)
    lb = (xd[0].float() * 255.0).round().to(torch.uint8).permute(1, 2, 0).flip(-1).contiguous().cpu().numpy()  # back to HWC BGR u8
    assert list(lb.shape) == meta["lb_shape"] and zlib.crc32(lb.tobytes()) == meta["lb_crc32"]
    eng = YoloEngine(stock_cfg("yolo11", "n"), P, **({} if half else {"precision": precision}))
    y, _ = eng(xd)
    assert list(y.shape) == meta["y_shape"]
    ytop = y[0].float().cpu()[:, torch.from_numpy(z["y_idx"])].numpy()  # before NMS: it converts the box rows to xyxy in place (ops.py:243-244)
    det, counts = HN.nms_batched(y, meta["conf"], meta["iou"], max_det=300)
    HN.scale_boxes_batched(det, counts, xd.shape[2:], [bgr.shape])
    torch.cuda.synchronize()
    es, eb = np.abs(ytop[4:] - z["y_top"][4:]).max(), np.abs(ytop[:4] - z["y_top"][:4]).max()
    n = int(counts[0])
    got, ref_pred, ref_boxes = det[0, :n].cpu().numpy(), z["pred"], z["boxes"]
    if half:
        # fp16 storage on a real photograph (wider activation range than the seeded-noise inputs of the golden graphs): measured score
        # max 8.1e-3, box max 1.8 px over the 1000 anchors -- the fp16 path's stated class of error, not 1e-3
        assert es < 2e-2 and eb < 4.0, (es, eb)
        assert abs(n - len(ref_pred)) <= max(3, len(ref_pred) // 10), (n, len(ref_pred))
        return eng.close()
    assert es <= 1e-3 and eb <= 1e-3 * 640, (precision, es, eb)
    assert n == meta["n_det"] == len(ref_pred), (n, meta["n_det"])
    assert np.array_equal(got[:, 5], ref_pred[:, 5])                       # same classes in the same (score) order
    assert np.abs(got[:, 4] - ref_pred[:, 4]).max() <= 1e-3
    assert np.abs(got[:, :4] - ref_boxes).max() <= 1e-3 * 1080             # boxes in the ORIGINAL image's pixels (1080 x 810)
    eng.close()


def test_fp32_mode_fast_pool_and_depthwise_kernels_return_the_same_bits(monkeypatch):
    """Round 4 gave the fp32-storage modes an LDS form of SPPF's pools and a register-window form of the 3 x 3 depthwise conv
    (csrc/ref32.hip).  Maxima are exact and every depthwise output still sums its taps in (kh, kw) order from the bias, so the exact
    fp32 mode returns the bits of the round-3 kernels (BSY_REF32_SLOW=1) -- on a graph with SPPF, Attention.pe and Detect's DWConv
    units, at a size with ragged window blocks (W = 72: 18 blocks of four; 9 x 7 maps at stride 32)."""
    m = R.Model("yolo11", "s", 80, "detect")
    P = R.synth_params(m, 6)
    x = torch.rand(2, 3, 288, 224, generator=torch.Generator().manual_seed(6)).to(DEV)
    outs = []
    for slow in ("1", "0"):
        monkeypatch.setenv("BSY_REF32_SLOW", slow)
        eng = YoloEngine(stock_cfg("yolo11", "s"), P, precision="fp32")
        y, raws = eng(x)
        torch.cuda.synchronize()
        outs.append((y.clone(), [r.clone() for r in raws]))
        eng.close()
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)


def test_engine_fp16_path_vs_fp32_mode_at_the_benchmark_size():
    """The benchmark configuration itself (YOLO11s, 64 x 640 x 640, bench weights, fp16 in): the fp16-storage product path
    against the engine's fp32 correctness mode on the same device -- the full-size parity statement.  Measured (r02): score
    max 2.7e-3 / p99.9 4.1e-4 / mean 2.2e-5; box max 7.3 px / p99.9 2.0 px / mean 0.11 px (a DFL expectation over a flat
    16-bin distribution at stride 32 moves 6 px for a 0.2-bin shift: random weights); 0.5 % of the anchors change side of
    conf 0.25.  Bounds ~2x above."""
    from bs_yolo_amd.plan import Plan
    from bs_yolo_amd.weights import synth_state_dict
    cfg = stock_cfg("yolo11", "s")
    sd = synth_state_dict(Plan(cfg, 1, 64, 64), seed=0)
    x = torch.rand(64, 3, 640, 640, generator=torch.Generator().manual_seed(1234)).half().to(DEV)
    e16, e32 = YoloEngine(cfg, sd), YoloEngine(cfg, sd, precision="fp32")
    y16 = e16(x, want_raw=False)[0].float()
    y32 = e32(x.float(), want_raw=False)[0]
    torch.cuda.synchronize()
    assert tuple(y16.shape) == tuple(y32.shape) == (64, 84, 8400)
    ds, db = (y16[:, 4:] - y32[:, 4:]).abs(), (y16[:, :4] - y32[:, :4]).abs()
    qs = float(torch.quantile(ds.flatten()[::7].float(), 0.999))  # every 7th element: torch.quantile caps the input size
    qb = float(torch.quantile(db.flatten().float(), 0.999))
    assert float(ds.max()) < 6e-3 and qs < 1e-3 and float(ds.mean()) < 5e-5, (float(ds.max()), qs, float(ds.mean()))
    assert float(db.max()) < 16.0 and qb < 4.0 and float(db.mean()) < 0.25, (float(db.max()), qb, float(db.mean()))
    n16, n32 = int((y16[:, 4:].amax(1) > 0.25).sum()), int((y32[:, 4:].amax(1) > 0.25).sum())
    assert abs(n16 - n32) <= 0.02 * n32 and n32 > 5000, (n16, n32)
    e16.close()
    e32.close()


# ------------------------------------------------------------------------------------------------------------
# retina_masks path: process_mask_native + scale_masks (utils/ops.py:696-737) on the device
# ------------------------------------------------------------------------------------------------------------
def test_process_mask_native_and_scale_masks_match_reference_golden():
    """bs_yolo_amd.masks.process_mask_native / scale_masks (csrc/masks.hip) against the REFERENCE's own outputs
    (tests/golden/masks_native.npz): the 0/1 masks bit-exact up to pixels whose pre-threshold value is within float rounding
    of zero (none in the fixtures), the resized maps within 1e-5; fp16 prototypes against the oracle on the same values."""
    from bs_yolo_amd import masks as HM
    z = np.load(GOLDEN / "masks_native.npz")
    for case in json.loads(str(z["cases"])):
        ci, shape = case["ci"], tuple(case["shape"])
        protos, coef, boxes = (torch.from_numpy(z[f"c{ci}.{k}"]) for k in ("protos", "coef", "boxes"))
        got = HM.process_mask_native(protos.to(DEV), coef.to(DEV), boxes.to(DEV), shape)
        assert tuple(got.shape) == (len(coef),) + shape and got.dtype == torch.float32
        want = z[f"c{ci}.native"]
        diff = got.cpu().numpy().astype(np.uint8) != want
        if diff.any():  # only where the reference's own pre-threshold value is ~0 (fp32 summation order of the nm-term dot)
            pre = PP.scale_masks((coef @ protos.view(protos.shape[0], -1)).view(-1, *protos.shape[1:])[None], shape)[0].numpy()
            assert np.abs(pre[diff]).max() < 1e-5, (ci, int(diff.sum()), np.abs(pre[diff]).max())
        for pad, key in ((True, "scaled"), (False, "scaled_nopad")):
            s = HM.scale_masks(protos[None, :2].to(DEV), shape, padding=pad)
            np.testing.assert_allclose(s.cpu().numpy(), z[f"c{ci}.{key}"], rtol=0, atol=1e-5, err_msg=f"case {ci} {key}")
        if len(coef):  # fp16 prototypes (what a half=True Segment head returns)
            ph = protos.half()
            g16 = HM.process_mask_native(ph.to(DEV), coef.to(DEV), boxes.to(DEV), shape, out_dtype=torch.uint8)
            w16 = PP.process_mask_native(ph.float(), coef, boxes, shape).numpy().astype(np.uint8)
            d16 = g16.cpu().numpy() != w16
            assert d16.mean() < 1e-4, (ci, d16.mean())
            s16 = HM.scale_masks(ph[None, :2].to(DEV), shape)
            assert s16.dtype == torch.float16
            np.testing.assert_allclose(s16.float().cpu().numpy(), PP.scale_masks(ph[None, :2].float(), shape).numpy(), rtol=0, atol=2e-3)


@pytest.mark.parametrize("fam,nc,task", [("yolo11", 1, "detect"), ("yolo11", 3, "detect"), ("yolo11", 20, "detect"), ("yolov8", 5, "detect"),
                                         ("yolo11", 2, "segment"), ("yolo11", 101, "detect"), ("yolov8", 70, "detect"), ("yolo11", 91, "segment")])
def test_engine_class_counts_of_custom_datasets(fam, nc, task):
    """nc = 1 .. 101 (custom datasets: the class convs have 1, 2, 3, 5, 20, 101 output channels -- not multiples of 8 -- and the class
    BRANCH is c3 = max(ch0, min(nc, 100)) wide (head.py:39): 70, 91 or 100 channels on the n scale, which the engine pads to a multiple
    of 8 with zero weights): both precisions against the oracle, fused decoder and Segment's decode op, legacy and depthwise heads."""
    m = R.Model(fam, "n", nc, task)
    P = R.synth_params(m, 6)
    for k in P:
        if ".cv3." in k and k.endswith(".2.bias"):
            P[k] = P[k] + 4.0  # some anchors above conf 0.25, whatever nc is
    cfg = stock_cfg(fam, "n", nc, task)
    x = torch.rand(2, 3, 96, 160, generator=torch.Generator().manual_seed(nc))
    with torch.inference_mode():
        out = m.forward(P, x)
    yref = out[0]
    e32 = YoloEngine(cfg, P, precision="fp32")
    y32 = e32(x.to(DEV))[0].cpu()
    assert y32.shape == yref.shape == (2, 4 + nc + (32 if task == "segment" else 0), 12 * 20 + 6 * 10 + 3 * 5)
    assert float((y32[:, 4:4 + nc] - yref[:, 4:4 + nc]).abs().max()) <= 1e-3 and float((y32[:, :4] - yref[:, :4]).abs().max()) <= 1e-3 * 160
    e16 = YoloEngine(cfg, P)
    for xin in (x.half(), x):   # f16 and f32 output tensors of the fp16 engine
        y16 = e16(xin.to(DEV))[0].float().cpu()
        d = (y16 - yref).abs()
        assert float(d[:, 4:4 + nc].max()) < 1e-2 and float(d[:, :4].max()) < 1.0, (float(d[:, 4:4 + nc].max()), float(d[:, :4].max()))
    from bs_yolo_amd import nms as HN
    det, cnt = HN.nms_batched(e16(x.half().to(DEV))[0], 0.25, 0.7, max_det=300, nc=nc)
    n0 = int(cnt[0])
    assert int(cnt.sum()) > 0 and (n0 == 0 or float(det[0, :n0, 5].max()) <= nc - 1)
    e16.close()
    e32.close()


@pytest.mark.parametrize("fam,scale", [("yolo11", "l"), ("yolo11", "x"), ("yolov8", "m"), ("yolov8", "x"), ("yolov5", "m"), ("yolov5", "l"), ("bsyolo11", "m")])
def test_engine_large_scales_match_oracle(fam, scale):
    """The l / x / m scales the golden fixtures do not hold (depth multiples 1.0 - 1.33, widths to 1.5: repeated C3k / C2f / C3 blocks,
    96-wide box branch of YOLO11x, 640-channel maps of YOLOv8x): fp32 mode at the north-star tolerance, fp16 engine at its stated bounds."""
    nc = 12 if fam == "bsyolo11" else 80
    m = R.Model(fam, scale, nc, "detect")
    P = _damped(R.synth_params(m, 7), 0.7)  # random weights this deep saturate the class scores otherwise (see _damped)
    cfg = stock_cfg(fam, scale, nc)
    x = torch.rand(1, 3, 64, 96, generator=torch.Generator().manual_seed(7))
    with torch.inference_mode():
        yref, _ = m.forward(P, x)
    e32 = YoloEngine(cfg, P, precision="fp32")
    y32 = e32(x.to(DEV))[0].cpu()
    e32.close()
    assert float((y32[:, 4:] - yref[:, 4:]).abs().max()) <= 1e-3 and float((y32[:, :4] - yref[:, :4]).abs().max()) <= 1e-3 * 96
    e16 = YoloEngine(cfg, P)
    d = (e16(x.half().to(DEV))[0].float().cpu() - yref).abs()
    e16.close()
    k = 3.0 if fam == "bsyolo11" else 1.0
    assert float(d[:, 4:].max()) < k * 1e-2 and float(d[:, :4].max()) < k * 1.0, (float(d[:, 4:].max()), float(d[:, :4].max()))


@pytest.mark.parametrize("shape", [(1, 32, 1280), (2, 1280, 32), (7, 96, 32), (1, 1600, 1600), (130, 64, 64)])
def test_engine_extreme_shapes_match_fp32_mode(shape):
    """Strips, a 1600 x 1600 image and 130 images of 64 x 64: the fp16 engine against the fp32 correctness mode (itself pinned to the
    reference at 1e-3) -- maps one pixel high at stride 32, 40 000-pixel levels, more images than any tile count assumption."""
    cfg = stock_cfg("yolo11", "n")
    m = R.Model("yolo11", "n", 80, "detect")
    P = R.synth_params(m, 1)
    B, H, W = shape
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(H + W))
    e32 = YoloEngine(cfg, P, precision="fp32")
    y32 = e32(x.to(DEV))[0].cpu()
    e32.close()
    ex = YoloEngine(cfg, P, precision="fp32x")   # round 4: the split-f16 mode on the same shapes -- its patch / big-tile / 6 x 20-tile kernels
    yx = ex(x.to(DEV))[0].cpu()                  # on one-pixel-high maps, 200 x 200 levels and 130 images -- against the exact mode
    ex.close()
    assert float((yx[:, 4:] - y32[:, 4:]).abs().max()) <= 2e-4 and float((yx[:, :4] - y32[:, :4]).abs().max()) <= 2e-4 * max(H, W)
    e16 = YoloEngine(cfg, P)
    y16 = e16(x.half().to(DEV))[0].float().cpu()
    e16.close()
    assert y16.shape == y32.shape == (B, 84, (H // 8) * (W // 8) + (H // 16) * (W // 16) + (H // 32) * (W // 32))
    d = (y16 - y32).abs()
    # max over up to 650 000 anchors of the fp16-storage error (stock-graph bounds of the golden tests x 2; mean below 1e-4 / 0.05 px)
    # boxes leave the fp16 engine as f16 for f16 inputs (as the reference's half model does): one ulp of a coordinate in [1024, 2048)
    # is a whole pixel, so the box bounds follow the ulp of the image's larger side
    ulp = 2.0 ** (math.floor(math.log2(max(H, W))) - 10)
    assert float(d[:, 4:].max()) < 2e-2 and float(d[:, :4].max()) < max(3.0, 3 * ulp), (float(d[:, 4:].max()), float(d[:, :4].max()))
    assert float(d[:, 4:].mean()) < 1e-4 and float(d[:, :4].mean()) < max(0.05, 0.25 * ulp), (float(d[:, 4:].mean()), float(d[:, :4].mean()))
    if B * H * W <= 64 * 64 * 8:
        with torch.inference_mode():
            yref, _ = m.forward(P, x)
        assert float((y32[:, 4:] - yref[:, 4:]).abs().max()) <= 1e-3


def test_engine_top_level_dwconv_layers_match_oracle():
    """A graph with `DWConv` layers of its own (conv.py:224-229; 3x3 s1 and 5x5 s2) through the engine, both precisions,
    against the oracle."""
    fam = "t_dw"
    R.GRAPHS[fam] = [(-1, 1, "Conv", (16, 3, 2)), (-1, 1, "DWConv", (16, 3, 1)), (-1, 1, "Conv", (32, 3, 2)), (-1, 1, "DWConv", (32, 5, 2)),
                     (-1, 1, "Conv", (64, 3, 2)), (-1, 1, "Conv", (64, 3, 2))]
    R.SCALES[fam] = {"n": (1.0, 1.0, 1024)}
    R.HEAD_FROM[fam] = (3, 4, 5)
    cfg = {"nc": 80, "scale": "n", "scales": {"n": [1.0, 1.0, 1024]},
           "backbone": [[-1, 1, "Conv", [16, 3, 2]], [-1, 1, "DWConv", [16, 3, 1]], [-1, 1, "Conv", [32, 3, 2]], [-1, 1, "DWConv", [32, 5, 2]],
                        [-1, 1, "Conv", [64, 3, 2]], [-1, 1, "Conv", [64, 3, 2]]],
           "head": [[[3, 4, 5], 1, "Detect", ["nc"]]]}
    try:
        m = R.Model(fam, "n", 80, "detect")
        P = R.synth_params(m, 4)
        x = torch.rand(2, 3, 96, 128, generator=torch.Generator().manual_seed(4))
        with torch.inference_mode():
            yref, rref = m.forward(P, x)
        e32 = YoloEngine(cfg, P, precision="fp32")
        y32, r32 = e32(x.to(DEV))
        assert float((y32.cpu()[:, 4:] - yref[:, 4:]).abs().max()) <= 1e-3 and float((y32.cpu()[:, :4] - yref[:, :4]).abs().max()) <= 1e-3 * 128
        e16 = YoloEngine(cfg, P)
        y16, _ = e16(x.half().to(DEV))
        d = (y16.float().cpu() - yref).abs()
        assert float(d[:, 4:].max()) < 1e-2 and float(d[:, :4].max()) < 1.0, (float(d[:, 4:].max()), float(d[:, :4].max()))
        assert sum(o["kind"] == L.OP_DWCONV_G for o in e16.plan_for(2, 96, 128, torch.float16, torch.float16)[0].ops) == 2
        e16.close()
        e32.close()
    finally:
        for d_ in (R.GRAPHS, R.SCALES, R.HEAD_FROM):
            d_.pop(fam, None)


@pytest.mark.parametrize("fam,width", [("yolo11", 0.375), ("yolo11", 0.125), ("yolov8", 0.375), ("bsyolo11", 0.375), ("yolo11", 0.1875), ("yolov8", 0.1875),
                                       ("yolov8", 0.3125), ("yolov5", 0.1875), ("yolov5", 0.3125)])
def test_engine_on_custom_width_multiples(fam, width):
    """Width multiples other than the stock scales' (a `scales:` entry of the user's yaml, tasks.py:937-941): at 0.375 / 0.125 the
    Bottlenecks inside YOLO11's C3k2 blocks have 12 / 4 hidden channels (carried on 16 / 8 with zero weights in the padding,
    plan.py bottleneck()), BS-YOLO's PMSFA runs on 24 channels (_pmsfa_padded).  At 0.1875 / 0.3125 / 0.15625 the CHUNKS of C2f / C3k2 / C3 /
    C3k blocks are 12 / 20 channels wide (block.py:3295-3317, 3320-3334; tasks.py:1016): round 4 carries them as zero-padded 16- / 24-channel
    pieces of the concat buffers on the fp16 path (plan.py T.cmap; before, only the fp32-storage modes ran such graphs).
    YOLO11's C2PSA then has one attention head of 48 / 96 channels (csrc/attention.hip takes key_dim 16 .. 64, head_dim 32 .. 128; 0.3125 would
    give 40 / 80, which only the fp32-storage modes run: test_plan_says_which_precisions_run_a_width below).  Every precision against the oracle."""
    R.SCALES[fam] = dict(R.SCALES[fam], t=(0.5, width, 1024))
    try:
        nc = 12 if fam == "bsyolo11" else 80
        m = R.Model(fam, "t", nc, "detect")
        P = R.synth_params(m, 11)
        cfg = stock_cfg(fam, "n", nc)
        cfg["scale"], cfg["scales"] = "t", {"t": [0.5, width, 1024]}
        x = torch.rand(2, 3, 96, 128, generator=torch.Generator().manual_seed(11))
        with torch.inference_mode():
            yref, rref = m.forward(P, x)
        for prec in ("fp32", "fp32x"):  # fp32x: channel counts its kernels do not take (Cin % 8, Cout % 4) run on the exact ones
            e32 = YoloEngine(cfg, P, precision=prec)
            y32, _ = e32(x.to(DEV))
            assert float((y32.cpu()[:, 4:] - yref[:, 4:]).abs().max()) <= 1e-3 and float((y32.cpu()[:, :4] - yref[:, :4]).abs().max()) <= 1e-3 * 128, prec
            e32.close()
        e16 = YoloEngine(cfg, P)
        y16, r16 = e16(x.half().to(DEV))
        d = (y16.float().cpu() - yref).abs()
        if width in (0.1875, 0.3125):
            # these seeded random models put logits of +-500 .. 800 on the head's raw maps (narrow stems, no trained normalisation), so the
            # DECODED scores of fp16 storage flip wherever a logit sits near 0 (yolov5 0.1875: score max 0.25 at a mean of 7e-4) -- the bound
            # is on the raw maps, relative to their range: measured max 2.3e-3 .. 3.0e-3, mean 3e-4 .. 4e-4 (the backbone layers of the same
            # run: 2e-3 .. 3e-3 max).  A padding channel read as data would be O(1) of the range.
            for a_, b_ in zip(r16, rref):
                e = (a_.float().cpu() - b_).abs()
                rng = float(b_.abs().max())
                assert float(e.max()) <= 8e-3 * rng and float(e.mean()) <= 1.5e-3 * rng, (float(e.max()) / rng, float(e.mean()) / rng)
            assert float(d[:, 4:].mean()) < 2e-3 and float(d[:, :4].mean()) < 0.5
            e16.close()
            return
        # fp16 storage on seeded random weights (test_engine_matches_reference_golden explains the statistics).  Measured (max / mean of
        # scores, max / mean px of boxes): yolo11 0.375 1.7e-2 / 3.8e-4 / 2.5 / 0.11, yolo11 0.125 6.9e-3 / 3.1e-5 / 0.66 / 0.04, yolov8 0.375
        # -- which has NO padded block -- 1.4e-2 / 3.7e-4 / 3.3 / 0.09, bsyolo11 0.375 6.5e-3 / 8.7e-5 / 1.1 / 0.08: the noise of these
        # weights at this depth, not the padding; the fp32 mode above holds 1e-3.  Padding read as data would be O(1) wrong.
        stats = (float(d[:, 4:].max()), float(d[:, 4:].mean()), float(d[:, :4].max()), float(d[:, :4].mean()))
        assert stats[0] < 5e-2 and stats[1] < 1e-3 and stats[2] < 8.0 and stats[3] < 0.3, stats
        e16.close()
    finally:
        R.SCALES[fam].pop("t", None)


def test_engine_pmsfa_on_widths_that_are_not_multiples_of_16():
    """`PMSFA` (block.py:3035-3054) on 24 and 8 channels -- halves of 12 / 4, quarters of 6 / 2 -- inside C3k2_gai blocks (c3k False:
    PMSFA(c); c3k True: C3k_gai with two PMSFA(c / 2), block.py:3079-3095) in a graph of its own through the engine, both precisions,
    against the oracle (pinned for these widths by modules_bsyolo.npz: pmsfa24 / pmsfa12 / pmsfa8 / c3k2_gai_f24 / _t24).  The plan lays
    conv1's output out in 8-channel-aligned zero-padded pieces (plan.py _pmsfa_padded).  Width 12 (a C3k_gai of 24 channels) runs in
    the fp32 mode only: the fp16 conv kernels read 8-channel pieces, so the blocks around such a PMSFA are not on the fp16 path either."""
    fam = "t_pmsfa"

    def graphs(c3k_width):
        rows = [(-1, 1, "Conv", (24, 3, 2)), (-1, 1, "C3k2_gai", (48, False, 0.5)), (-1, 1, "Conv", (48, 3, 2)),
                (-1, 1, "C3k2_gai", (c3k_width, True)), (-1, 1, "Conv", (64, 3, 2)), (-1, 1, "C3k2_gai", (32, False, 0.25))]
        cfg = {"nc": 12, "scale": "n", "scales": {"n": [1.0, 1.0, 1024]}, "backbone": [[f, n, t, list(a)] for f, n, t, a in rows],
               "head": [[[1, 3, 5], 1, "Detect", ["nc"]]]}
        return rows, cfg

    R.SCALES[fam] = {"n": (1.0, 1.0, 1024)}
    R.HEAD_FROM[fam] = (1, 3, 5)
    x = torch.rand(2, 3, 96, 128, generator=torch.Generator().manual_seed(7))
    try:
        for c3k_width, fp16 in ((48, False), (96, True)):  # C3k_gai(24) -> PMSFA(12): fp32 mode only; C3k_gai(48) -> PMSFA(24)
            R.GRAPHS[fam], cfg = graphs(c3k_width)
            m = R.Model(fam, "n", 12, "detect")
            P = R.synth_params(m, 7)
            with torch.inference_mode():
                yref, rref = m.forward(P, x)
            e32 = YoloEngine(cfg, P, precision="fp32")
            y32, r32 = e32(x.to(DEV))
            assert float((y32.cpu()[:, 4:] - yref[:, 4:]).abs().max()) <= 1e-3 and float((y32.cpu()[:, :4] - yref[:, :4]).abs().max()) <= 1e-3 * 128
            e32.close()
            if not fp16:
                with pytest.raises(NotImplementedError):
                    YoloEngine(cfg, P)(x.half().to(DEV))
                continue
            e16 = YoloEngine(cfg, P)
            y16, _ = e16(x.half().to(DEV))
            d = (y16.float().cpu() - yref).abs()
            assert float(d[:, 4:].max()) < 1e-2 and float(d[:, :4].max()) < 1.0, (float(d[:, 4:].max()), float(d[:, :4].max()))
            plan = e16.plan_for(2, 96, 128, torch.float16, torch.float16)[0]
            assert sum(1 for r in plan.wrecs.values() if r.rows is not None) == 12 and sum(1 for r in plan.wrecs.values() if r.cols is not None) == 4
            e16.close()
    finally:
        for d_ in (R.GRAPHS, R.SCALES, R.HEAD_FROM):
            d_.pop(fam, None)


def test_engine_ela_on_widths_that_are_not_multiples_of_16():
    """`ELA` (nn/Addmodules/ELA.py:33-101) on 24 and 40 channels -- GroupNorm(max(1, c // 16), c) = one group of 24, two of 20 -- in a
    graph of its own through the engine, both precisions, against the oracle (pinned for these widths by modules_bsyolo.npz)."""
    fam = "t_ela"
    R.GRAPHS[fam] = [(-1, 1, "Conv", (24, 3, 2)), (-1, 1, "ELA", (24,)), (-1, 1, "Conv", (40, 3, 2)), (-1, 1, "ELA", (40,)),
                     (-1, 1, "Conv", (64, 3, 2)), (-1, 1, "Conv", (64, 3, 2))]
    R.SCALES[fam] = {"n": (1.0, 1.0, 1024)}
    R.HEAD_FROM[fam] = (3, 4, 5)
    cfg = {"nc": 12, "scale": "n", "scales": {"n": [1.0, 1.0, 1024]},
           "backbone": [[-1, 1, "Conv", [24, 3, 2]], [-1, 1, "ELA", [24]], [-1, 1, "Conv", [40, 3, 2]], [-1, 1, "ELA", [40]],
                        [-1, 1, "Conv", [64, 3, 2]], [-1, 1, "Conv", [64, 3, 2]]],
           "head": [[[3, 4, 5], 1, "Detect", ["nc"]]]}
    try:
        m = R.Model(fam, "n", 12, "detect")
        P = R.synth_params(m, 5)
        x = torch.rand(2, 3, 96, 128, generator=torch.Generator().manual_seed(5))
        with torch.inference_mode():
            yref, rref = m.forward(P, x)
        e32 = YoloEngine(cfg, P, precision="fp32")
        y32, r32 = e32(x.to(DEV))
        assert float((y32.cpu()[:, 4:] - yref[:, 4:]).abs().max()) <= 1e-3 and float((y32.cpu()[:, :4] - yref[:, :4]).abs().max()) <= 1e-3 * 128
        e16 = YoloEngine(cfg, P)
        y16, _ = e16(x.half().to(DEV))
        d = (y16.float().cpu() - yref).abs()
        assert float(d[:, 4:].max()) < 1e-2 and float(d[:, :4].max()) < 1.0, (float(d[:, 4:].max()), float(d[:, :4].max()))
        assert sum(o["kind"] == L.OP_ELA for o in e16.plan_for(2, 96, 128, torch.float16, torch.float16)[0].ops) == 2
        e16.close()
        e32.close()
    finally:
        for d_ in (R.GRAPHS, R.SCALES, R.HEAD_FROM):
            d_.pop(fam, None)
