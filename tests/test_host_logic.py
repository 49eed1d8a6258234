"""CPU tests of the host side: C-ABI surface, graph flattening, weight packing, letterbox geometry, sharding."""
import json
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, ROOT

from bs_yolo_amd import lib as L
from bs_yolo_amd.graphs import stock_cfg
from bs_yolo_amd.letterbox import LetterBox
from bs_yolo_amd.parallel import shard_bounds
from bs_yolo_amd.plan import Plan
from bs_yolo_amd.weights import fold_conv_bn, pack_plan_weights, pack_record, synth_state_dict
from oracle import yolo_ref as R


def test_library_exports_every_declared_symbol():
    hdr = (ROOT / "include" / "bsyolo.h").read_text()
    declared = set(re.findall(r"\b(bsy_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    for s in declared:
        assert hasattr(L.lib, s), s
    assert L.lib.bsy_version() >= 1


def test_c_abi_rejects_bad_arguments_without_a_gpu():
    import ctypes as C
    cp, kp = C.c_int(), C.c_int()
    assert L.lib.bsy_conv_packed_dims(80, 64, 3, C.byref(cp), C.byref(kp)) == 0
    assert (cp.value, kp.value) == (128, 576)
    assert L.lib.bsy_conv_packed_dims(16, 8, 3, C.byref(cp), C.byref(kp)) == 0
    assert (cp.value, kp.value) == (128, 96)
    assert L.lib.bsy_conv_packed_dims(16, 8, 5, C.byref(cp), C.byref(kp)) != 0
    assert b"bad shape" in L.lib.bsy_last_error()
    with pytest.raises(L.BsyError):
        L.check(L.lib.bsy_nms(None, 0, 1, 1, 0, 1, 0.25, 0.45, None, 0, 0, 0, 300, 30000, 7680.0, 1, None, None, None, 0, None))
    assert L.lib.bsy_nms_workspace_bytes(64, 8400, 80, 0, 30000) >= 64 * 16384 * 8
    assert L.lib.bsy_nms_workspace_bytes(2, 8400, 80, 1, 30000) >= 2 * (1 << 20) * 8


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/bsyolo.h compiles as strict C99 (no C++ in the signatures), and a C translation unit that
    names every declared entry point links against the shared library (no GPU call is made)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    hdr = ROOT / "include" / "bsyolo.h"
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", str(hdr)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    names = sorted(set(re.findall(r"\b(bsy_[a-z0-9_]+)\s*\(", hdr.read_text())))
    src = tmp_path / "abi.c"
    src.write_text('#include "bsyolo.h"\n#include <stdio.h>\nint main(void) {\n  const void* f[] = {' + ", ".join(f"(const void*)(size_t){n}" for n in names) +
                   '};\n  printf("%d %d %d\\n", (int)(sizeof(f) / sizeof(f[0])), bsy_version(), bsy_sizeof_op());\n  return 0;\n}\n')
    exe = tmp_path / "abi"
    r = subprocess.run([gcc, "-std=c99", "-I", str(hdr.parent), str(src), "-o", str(exe), str(L.LIB_PATH), f"-Wl,-rpath,{L.LIB_PATH.parent}",
                        "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-Wl,--allow-shlib-undefined"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    n, ver, sz = map(int, r.stdout.split())
    import ctypes as C
    assert n == len(names) == len(L.SYMBOLS) and ver >= 1 and sz == C.sizeof(L.Op)


def test_chain_fusion_peephole():
    """plan.py _fuse_chains (opt-in, csrc/chain1x1.hip): two consecutive 1x1 Conv launches of which the second reads (part of) the first's
    output at the same pixel become ONE OP_CHAIN op: C3k2.cv1 -> C3k.cv1|cv2 and C2PSA.cv1 -> qkv (HEAD form: a channel slice of the first
    output; it is still written), C3k.cv3 -> C3k2.cv2 and ffn[1] -> C2PSA.cv2 (TAIL form: the first output is the LAST channels of the
    second's source and is no longer written).  Same weights records, same work, fewer ops; nothing below plan.CHAIN_MIN_PIXELS; off by default."""
    from bs_yolo_amd.plan import CHAIN_MIN_PIXELS, chain_supported
    cfg = stock_cfg("yolo11", "s")
    a, b = Plan(cfg, 64, 640, 640, fuse_chain=False), Plan(cfg, 64, 640, 640, fuse_chain=True)
    assert not any(o["kind"] == L.OP_CHAIN for o in Plan(cfg, 64, 640, 640).ops)
    ch = [o for o in b.ops if o["kind"] == L.OP_CHAIN]
    assert [o["name"] for o in ch] == ["model.6.cv1->m.0.cv1", "model.6.m.0.cv3->cv2", "model.8.cv1->m.0.cv1", "model.8.m.0.cv3->cv2",
                                       "model.10.cv1->m.0.attn.qkv", "model.10.m.0.ffn.1->cv2", "model.22.cv1->m.0.cv1", "model.22.m.0.cv3->cv2"]
    assert len(a.ops) - len(b.ops) == 8 and list(a.wrecs) == list(b.wrecs)
    assert sum(o.get("mfma_flops", 0) for o in a.ops) == sum(o.get("mfma_flops", 0) for o in b.ops) and a.flops == b.flops
    for o in ch:
        d1, h2, r1 = o["box"]
        n1, keep0, lc = o["heads"], o["key_dim"], o["mid_c"]
        ca1 = o["src1"].C if o["src1"] is not None else 0
        assert chain_supported(o["src0"].C, ca1, n1, keep0, lc, h2.C if h2 is not None else 0, o["dst"].C)
        if h2 is None:   # HEAD: the first conv's output is written, the second reads its channels [keep0, keep0 + lc)
            assert d1 is not None and d1.C == n1 and keep0 + lc <= n1 and r1 is None
        else:            # TAIL: [h2 | kept] is the second conv's source view; the first output stays on chip
            assert d1 is None and keep0 == 0 and lc == n1 and h2.C % 64 == 0
        assert set(b.op_buffers(o)) >= {t.buf for t in (o["src0"], o["dst"]) if t.buf < L.BSY_EXT_BASE}
    assert ch[5]["box"][2] is not None and ch[5]["nl"] == 0 and ch[5]["act"] == 1    # ffn[1]: no activation, shortcut operand in the first epilogue
    assert ch[4]["act"] == 0 and ch[4]["nl"] == 1                                    # qkv: a bare conv behind cv1's SiLU
    offs, top = b.assign_offsets()
    assert top > 0 and len(offs) == len(b.buf_bytes)
    small = Plan(cfg, 8, 640, 640, fuse_chain=True)     # 8 x 40 x 40 = 12 800 pixels < CHAIN_MIN_PIXELS: the two launches stay
    assert CHAIN_MIN_PIXELS == 24576 and not any(o["kind"] == L.OP_CHAIN for o in small.ops)
    assert not any(o["kind"] == L.OP_CHAIN for o in Plan(cfg, 64, 640, 640, fuse_chain=True, precision="fp32x").ops)
    for sc, n in (("n", 6), ("m", 14), ("l", 21), ("x", 0)):  # x: 384-channel chunks exceed the 256-channel resident tile
        assert sum(o["kind"] == L.OP_CHAIN for o in Plan(stock_cfg("yolo11", sc), 64, 640, 640, fuse_chain=True).ops) == n
    assert L.OP_CHAIN == 21


def test_op_struct_layout():
    import ctypes as C
    # mirrors include/bsyolo.h: the two int64 offsets sit on an 8-byte boundary after 32 int32 fields
    assert L.Op.w_off.offset == 128 and L.Op.b_off.offset == 136
    assert C.sizeof(L.View) == 16
    assert C.sizeof(L.Op) % 8 == 0


@pytest.mark.parametrize("scale,gflops,nconv", [("n", 6.48, 80), ("s", 21.47, 80), ("m", 67.98, 105)])
def test_plan_work_matches_reference_counts(scale, gflops, nconv, monkeypatch):
    """Dense-conv FLOPs per 640x640 image equal the reference graph (BASELINE.md section 2 / SURVEY 8d); the
    reference counts 81 dense convs for n/s because DFL is a 1x1 conv there (block.py:58-77), here it lives in decode."""
    monkeypatch.setenv("BSY_FUSE_BOXTAIL", "0")  # one op per conv module
    p = Plan(stock_cfg("yolo11", scale), 1, 640, 640, fuse_stem=False, fuse_bneck=False, fuse_dwpw=False, merge_c3k=False, fuse_tail=False)
    dense = 0
    n = 0
    for o in p.ops:
        if o["kind"] in (L.OP_CONV, L.OP_CONV_FIRST):
            cin = 3 if o["kind"] == L.OP_CONV_FIRST else o["src0"].C + (o["src1"].C if o.get("src1") else 0)
            dense += 2 * o["OH"] * o["OW"] * o.get("cout", o["dst"].C) * cin * o["ksize"] ** 2
            n += 1
    assert n == nconv
    assert abs(dense / 1e9 - gflops) < 0.01 * gflops
    assert p.meta["A"] == 8400 and p.meta["strides"] == [8.0, 16.0, 32.0]


def test_box_branch_tail_is_one_op(monkeypatch):
    """Detect's cv2.i.1 + cv2.i.2 (+ DFL) become ONE conv op (plan.py _box_tail) where the branch is 64 wide; the weight records --
    and with them the blob layout and synth_state_dict's draws -- are those of the unfused plan."""
    cfg = stock_cfg("yolo11", "s")
    fused = Plan(cfg, 2, 640, 640)
    monkeypatch.setenv("BSY_FUSE_BOXTAIL", "0")
    plain = Plan(cfg, 2, 640, 640)
    assert list(fused.wrecs) == list(plain.wrecs)
    assert len(plain.ops) - len(fused.ops) == 3 and fused.flops == plain.flops
    tails = [o for o in fused.ops if o["kind"] == L.OP_CONV and o.get("mid_c")]
    assert [o["name"] for o in tails] == [f"model.23.cv2.{i}.1+2" for i in range(3)]
    for o in tails:
        assert (o["ksize"], o["out_f32"], o["act"], o["cout"], o["wkey2"]) == (3, 3, 1, 64, o["wkey"][:-1] + "2")
    monkeypatch.delenv("BSY_FUSE_BOXTAIL")
    # YOLO11x: the box branch is 96 wide (max(16, 384 // 4, 64)) -> no tail
    assert not [o for o in Plan(stock_cfg("yolo11", "x"), 1, 64, 64).ops if o["kind"] == L.OP_CONV and o.get("mid_c")]
    # Segment heads keep the separate decode op (mask coefficients): no tail either
    assert not [o for o in Plan(stock_cfg("yolo11", "n", 80, "segment"), 1, 64, 64).ops if o["kind"] == L.OP_CONV and o.get("mid_c")]


def test_c3k_branch_merge():
    """C3k's cv1 and cv2 (same input) become one conv of twice the width writing the concat buffer; parameters, their
    random draws and the packed rows are those of the two separate convs."""
    from bs_yolo_amd.weights import pack_plan_weights, synth_state_dict
    cfg = stock_cfg("yolo11", "s")
    a, b = Plan(cfg, 1, 64, 64, merge_c3k=False), Plan(cfg, 1, 64, 64, merge_c3k=True)
    assert len(a.ops) - len(b.ops) == 3 and a.flops == b.flops
    merged = [o for o in b.ops if "+" in str(o.get("wkey", ""))]
    assert [o["name"] for o in merged] == ["model.6.m.0.cv1", "model.8.m.0.cv1", "model.22.m.0.cv1"]
    sa, sb = synth_state_dict(a, 0), synth_state_dict(b, 0)
    assert sa.keys() == sb.keys() and all(torch.equal(sa[k], sb[k]) for k in sa)
    ba, bb = pack_plan_weights(a, sa), pack_plan_weights(b, sb)
    ra1, ra2, rb = a.wrecs["model.6.m.0.cv1"], a.wrecs["model.6.m.0.cv2"], b.wrecs["model.6.m.0.cv1+model.6.m.0.cv2"]
    c_, kp = ra1.cout, 128  # 1x1 over 128 input channels: rows of 128 halves
    rows = lambda blob, off, n: blob[off: off + n * kp * 2]
    assert rb.cout == 2 * c_ and rows(bb, rb.w_off, c_) == rows(ba, ra1.w_off, c_)
    assert rows(bb, rb.w_off + c_ * kp * 2, c_) == rows(ba, ra2.w_off, c_)
    assert bb[rb.b_off: rb.b_off + 4 * c_] == ba[ra1.b_off: ra1.b_off + 4 * c_]
    assert bb[rb.b_off + 4 * c_: rb.b_off + 8 * c_] == ba[ra2.b_off: ra2.b_off + 4 * c_]


def test_stem_fusion_peephole():
    """Layers 0 + 1 become one OP_STEM launch where csrc/stem_fused.hip supports the widths (n: 16/32, s: 32/64);
    m/l/x keep two launches.  Work accounting (plan.flops) and the parameter records do not change."""
    for scale, fused in (("n", True), ("s", True), ("m", False)):
        a = Plan(stock_cfg("yolo11", scale), 2, 640, 640, fuse_stem=False, fuse_bneck=False, fuse_tail=False)
        b = Plan(stock_cfg("yolo11", scale), 2, 640, 640, fuse_stem=True, fuse_bneck=False, fuse_tail=False)
        assert a.flops == b.flops and list(a.wrecs) == list(b.wrecs)
        if fused:
            assert len(b.ops) == len(a.ops) - 1 and b.ops[0]["kind"] == L.OP_STEM
            assert b.ops[0]["dst"] is not None and b.ops[0]["mid_c"] == a.ops[0]["dst"].C
            assert b.ops[0]["wkey"] == "model.0" and b.ops[0]["wkey2"] == "model.1"
            assert b.layer_out[0] is None and sum(b.buf_bytes) < sum(a.buf_bytes)
        else:
            assert len(b.ops) == len(a.ops) and b.ops[0]["kind"] == L.OP_CONV_FIRST
    odd = Plan(stock_cfg("yolo11", "s"), 1, 64, 64, fuse_stem=True)
    assert odd.ops[0]["kind"] == L.OP_STEM
    assert L.OP_STEM == 8 and L.Op.mid_c.offset == 380 and L.Op.w2_off.offset == 384 and L.Op.b2_off.offset == 392


def test_bottleneck_fusion_peephole():
    """C3k2's thin Bottlenecks with shortcut (32 -> 16 -> 32: model.2.m.0 of YOLO11s; 64 -> 32 -> 64: model.4.m.0 / model.16.m.0)
    become one OP_BNECK launch each; other widths keep two convs.  FLOP accounting and parameter records are unchanged."""
    a = Plan(stock_cfg("yolo11", "s"), 2, 640, 640, fuse_stem=False, fuse_bneck=False, fuse_tail=False)
    b = Plan(stock_cfg("yolo11", "s"), 2, 640, 640, fuse_stem=False, fuse_bneck=True, fuse_tail=False)
    assert a.flops == b.flops and list(a.wrecs) == list(b.wrecs)
    fused = [o for o in b.ops if o["kind"] == L.OP_BNECK]
    assert [o["name"] for o in fused] == ["model.2.m.0", "model.4.m.0", "model.16.m.0"] and len(b.ops) == len(a.ops) - 3
    o = fused[0]
    assert (o["src0"].C, o["mid_c"], o["dst"].C) == (32, 16, 32) and o["src0"].buf == o["dst"].buf  # slices of the concat buffer
    assert o["wkey"] == "model.2.m.0.cv1" and o["wkey2"] == "model.2.m.0.cv2" and L.OP_BNECK == 9
    assert [(o["src0"].C, o["mid_c"]) for o in fused[1:]] == [(64, 32), (64, 32)]
    for r in ("cv1", "cv2"):
        ra, rb = a.wrecs[f"model.2.m.0.{r}"], b.wrecs[f"model.2.m.0.{r}"]
        assert (ra.kind, ra.cout, ra.cin, ra.k) == (rb.kind, rb.cout, rb.cin, rb.k)
    n = Plan(stock_cfg("yolo11", "n"), 1, 64, 64, fuse_bneck=True, fuse_tail=False)  # same widths one level down
    assert [o["name"] for o in n.ops if o["kind"] == L.OP_BNECK] == ["model.4.m.0", "model.13.m.0", "model.16.m.0", "model.19.m.0"]
    m = Plan(stock_cfg("yolo11", "m"), 1, 64, 64, fuse_bneck=True)  # C3k blocks, e = 1.0: not fused
    assert not any(o["kind"] == L.OP_BNECK for o in m.ops)


def test_yolov5u_plan():
    """YOLOv5u (cfg/models/v5/yolov5.yaml): the 6x6 stride-2 pad-2 stem becomes space-to-depth + a 3x3 conv over 16 channels
    whose packed weights are the module's 6x6 weights re-laid-out (exactly: every 6x6 tap appears once, the padding channels
    are zero); C3 blocks expand to cv1 / cv2 / n x (1x1, 3x3 + shortcut) / cv3; dense FLOPs match the documented model size."""
    from bs_yolo_amd.weights import pack_record
    p = Plan(stock_cfg("yolov5", "s"), 1, 640, 640, merge_c3k=False)
    assert p.ops[0]["kind"] == L.OP_S2D and p.ops[1]["kind"] == L.OP_CONV and (p.ops[1]["ksize"], p.ops[1]["stride"], p.ops[1]["src0"].C) == (3, 1, 16)
    r = p.wrecs["model.0"]
    assert (r.kind, r.cout, r.cin, r.k) == ("first_s2d", 32, 3, 6)
    g = torch.Generator().manual_seed(0)
    sd = {"model.0.conv.weight": torch.randn(32, 3, 6, 6, generator=g), "model.0.conv.bias": torch.randn(32, generator=g)}
    wb, _ = pack_record(sd, r)
    wp = torch.frombuffer(bytearray(wb), dtype=torch.float16).view(128, -1)[:32, :144].float().view(32, 3, 3, 16)  # [co][a][b][(dy,dx,c)]
    x = torch.randn(1, 3, 12, 16, generator=g).half().float()
    s2d = torch.zeros(1, 16, 6, 8)
    for dy in (0, 1):
        for dx in (0, 1):
            s2d[:, (dy * 2 + dx) * 3:(dy * 2 + dx) * 3 + 3] = x[:, :, dy::2, dx::2]
    got = F.conv2d(s2d, wp.permute(0, 3, 1, 2), None, 1, 1)
    want = F.conv2d(x, sd["model.0.conv.weight"].half().float(), None, 2, 2)
    assert torch.allclose(got, want, atol=1e-4)
    assert abs(p.flops / 1e9 - 24.0) < 0.5  # docs/en/models/yolov5.md: YOLOv5su 24.0 GFLOPs
    names = [o["name"] for o in p.ops if o["name"].startswith("model.2.")]
    assert names == ["model.2.cv1", "model.2.cv2", "model.2.m.0.cv1", "model.2.m.0.cv2", "model.2.cv3"]
    m0 = [o for o in p.ops if o["name"].startswith("model.2.m.0.")]
    assert (m0[0]["ksize"], m0[1]["ksize"]) == (1, 3) and m0[1]["res"] is not None  # Bottleneck k = ((1,1),(3,3)) + shortcut
    head_c3 = [o for o in p.ops if o["name"] == "model.13.m.0.cv2"][0]
    assert head_c3["res"] is None  # C3(..., False) in the neck
    assert [o["name"] for o in Plan(stock_cfg("yolov5", "m"), 1, 64, 64).ops if "+" in str(o.get("wkey", ""))][:1] == ["model.2.cv1"]  # merged cv1 + cv2


def test_top_level_dwconv_layer():
    """A `DWConv` layer of a yaml graph (conv.py:224-229) runs on the generic depthwise kernel; c1 != c2 (a grouped conv) is
    refused so the caller keeps the reference forward."""
    cfg = {"nc": 80, "scale": "n", "scales": {"n": [1.0, 1.0, 1024]},
           "backbone": [[-1, 1, "Conv", [16, 3, 2]], [-1, 1, "DWConv", [16, 3, 1]], [-1, 1, "Conv", [32, 3, 2]], [-1, 1, "DWConv", [32, 5, 2]],
                        [-1, 1, "Conv", [64, 3, 2]], [-1, 1, "Conv", [64, 3, 2]]],
           "head": [[[3, 4, 5], 1, "Detect", ["nc"]]]}
    p = Plan(cfg, 1, 64, 64)
    dw = [o for o in p.ops if o["kind"] == L.OP_DWCONV_G]
    assert [(o["ksize"], o["stride"], o["src0"].C) for o in dw] == [(3, 1, 16), (5, 2, 32)]
    assert p.wrecs["model.1"].kind == "dwg" and p.wrecs["model.3"].k == 5
    bad = dict(cfg, backbone=[[-1, 1, "Conv", [16, 3, 2]], [-1, 1, "DWConv", [32, 3, 1]]] + cfg["backbone"][2:])
    with pytest.raises(NotImplementedError):
        Plan(bad, 1, 64, 64)


def test_c3k2_fusion_peephole():
    """A whole C3k2 block with one thin Bottleneck (Cin 64 -> c 32 -> 128: model.2 of YOLO11s, model.4 of YOLO11n) becomes ONE
    OP_C3K2 launch (csrc/c3k2_fused.hip); blocks fed by a Concat (model.16 of YOLO11n) and C3k blocks keep their launches.
    FLOP accounting, parameter records and their order (synth_state_dict draws follow it) are unchanged."""
    for scale, want in (("s", ["model.2"]), ("n", ["model.4"]), ("m", [])):
        a = Plan(stock_cfg("yolo11", scale), 2, 640, 640, fuse_tail=False)
        b = Plan(stock_cfg("yolo11", scale), 2, 640, 640, fuse_tail=True)
        assert a.flops == b.flops and list(a.wrecs) == list(b.wrecs)
        fused = [o for o in b.ops if o["kind"] == L.OP_C3K2]
        assert [o["name"] for o in fused] == want
        assert sum(o.get("mfma_flops", 0) for o in a.ops) == sum(o.get("mfma_flops", 0) for o in b.ops)
        for o in fused:
            assert (o["src0"].C, o["mid_c"], o["dst"].C) == (64, 32, 128) and len(o["wkeys"]) == 4
            assert o["wkeys"] == [o["name"] + sfx for sfx in (".cv1", ".m.0.cv1", ".m.0.cv2", ".cv2")]
            assert len(a.ops) - len(b.ops) == 2  # cv1 conv + fused Bottleneck + cv2 conv -> one op
            assert sum(b.buf_bytes) < sum(a.buf_bytes)  # no concat buffer
    assert L.OP_C3K2 == 18 and L.Op.aux_off.offset % 8 == 0


def test_dwpw_fusion_peephole():
    """YOLO11's class branch units nn.Sequential(DWConv, Conv 1x1) (head.py:49-57) become one OP_DWPW launch where the
    depthwise width is <= 256 (levels 0 and 1 of YOLO11s; the 512-wide level keeps two launches)."""
    a = Plan(stock_cfg("yolo11", "s"), 2, 640, 640, fuse_dwpw=False)
    b = Plan(stock_cfg("yolo11", "s"), 2, 640, 640, fuse_dwpw=True)
    assert a.flops == b.flops and list(a.wrecs) == list(b.wrecs)
    fused = [o["name"] for o in b.ops if o["kind"] == L.OP_DWPW]
    assert fused == ["model.23.cv3.0.0", "model.23.cv3.0.1", "model.23.cv3.1.0", "model.23.cv3.1.1", "model.23.cv3.2.1"]
    assert len(b.ops) == len(a.ops) - len(fused) and L.OP_DWPW == 16
    assert any(o["kind"] == L.OP_DWCONV and o["name"] == "model.23.cv3.2.0.0" for o in b.ops)


def test_plan_consumes_exactly_the_reference_parameters():
    for fam, scale in (("yolo11", "n"), ("yolo11", "x"), ("yolov8", "s"), ("bsyolo11", "n"), ("bsyolo11", "m"), ("yolov5", "n"),
                       ("yolov5", "l")):
        p = Plan(stock_cfg(fam, scale), 2, 96, 64)
        m = R.Model(fam, scale, 80, "detect")
        want = {n for n, _ in m.param_specs() if not n.endswith("dfl.conv.weight")}
        used = set()
        for r in p.wrecs.values():
            if r.kind in ("plain", "dwg_plain"):
                used |= {r.name + ".weight", r.name + ".bias"}
                if r.post:
                    used |= {r.post + ".weight", r.post + ".bias"}
            elif r.kind == "ela":
                used |= {f"{r.name}.{s}" for s in ("ch_weight", "sp_weight", "res_weight", "ch_att.2.weight",
                                                   "spatial_conv.weight", "gn.weight", "gn.bias")}
            else:
                for nm in ((r.name, r.post) if r.kind == "conv2" else (r.name,)):
                    used |= {nm + ".conv.weight"} | {f"{nm}.bn.{s}" for s in ("weight", "bias", "running_mean", "running_var")}
        assert used == want
        # every view stays inside its buffer and respects the kernels' alignment rules
        for o in p.ops:
            for key in ("src0", "src1", "dst", "res"):
                t = o.get(key)
                if t is None or t.buf >= L.BSY_EXT_BASE:
                    continue
                assert t.coff + t.C <= t.ld
                assert t.coff % (4 if t.f32 else 8) == 0
                hs, ws = (t.H // 2, t.W // 2) if t.up else (t.H, t.W)
                if o["kind"] == L.OP_ELA and key == "res":
                    continue  # ELA's f32 scratch: sized by the op, not by a map shape
                assert p.buf_bytes[t.buf] == p.B * hs * ws * t.ld * (4 if t.f32 else 2)


@pytest.mark.parametrize("fam,width", [("yolo11", 0.1875), ("yolov8", 0.1875), ("yolov8", 0.3125), ("yolov5", 0.3125), ("yolov5", 0.1875)])
def test_c2f_chunks_that_are_not_multiples_of_8_are_zero_padded_pieces(fam, width):
    """A width multiple like 0.1875 gives C2f / C3k2 chunk widths of 12 channels (tasks.py:1016, block.py:3295-3317), 0.3125 gives 20;
    YOLOv5u's C3 then has halves of 12 / 20 channels (the C3k path).  The fp16 kernels read 8-channel pieces, so the plan carries every chunk on the next multiple of 8 as a
    zero-padded piece (plan.py T.cmap): the conv that writes a piece has zero weight rows and bias at the padding, the conv that reads it zero
    weight columns.  Checked on the packed matrices, against the module's own folded weights."""
    R.SCALES[fam] = dict(R.SCALES[fam], t=(0.5, width, 1024))
    try:
        cfg = dict(stock_cfg(fam, "n"), scale="t", scales={"t": [0.5, width, 1024]})
        p = Plan(cfg, 2, 96, 64)
        m = R.Model(fam, "t", 80, "detect")
        want = {n for n, _ in m.param_specs() if not n.endswith("dfl.conv.weight")}
        used = set()
        for r in p.wrecs.values():
            if r.kind == "plain":
                used |= {r.name + ".weight", r.name + ".bias"}
            else:
                for nm in ((r.name, r.post) if r.kind == "conv2" else (r.name,)):
                    used |= {nm + ".conv.weight"} | {f"{nm}.bn.{s}" for s in ("weight", "bias", "running_mean", "running_var")}
        assert used == want
        sd = synth_state_dict(p, 3)
        for n_, shape in m.param_specs():  # the synthetic parameters have the MODULE's shapes, not the padded ops'
            if n_ in sd:
                assert tuple(sd[n_].shape) == tuple(shape), (n_, tuple(sd[n_].shape), tuple(shape))
        blob = pack_plan_weights(p, sd)
        padded = 0
        for o in p.ops:
            if o["kind"] != L.OP_CONV:
                continue
            srcs = [t for t in (o["src0"], o.get("src1")) if t is not None]
            dst = o["dst"]
            if dst.cmap is None and all(t.cmap is None for t in srcs):
                continue
            padded += 1
            r = p.wrecs[o["wkey"]]
            k, cin, cout = o["ksize"], sum(t.C for t in srcs), dst.C
            cp, kp = (cout + 127) // 128 * 128, (k * k * cin + 31) // 32 * 32
            wp = torch.frombuffer(bytearray(blob[r.w_off:r.w_off + cp * kp * 2]), dtype=torch.float16).view(cp, kp).float()
            bp = torch.frombuffer(bytearray(blob[r.b_off:r.b_off + cp * 4]), dtype=torch.float32)
            if r.kind == "conv2":
                (w0, b0), (w1, b1) = fold_conv_bn(sd, r.name), fold_conv_bn(sd, r.post)
                wm, bm = torch.cat([w0, w1]), torch.cat([b0, b1])
            else:
                wm, bm = fold_conv_bn(sd, r.name)
            rows = list(dst.cmap) if dst.cmap is not None else list(range(cout))
            cols, base = [], 0
            for t in srcs:
                cols += [base + v if v >= 0 else -1 for v in (t.cmap if t.cmap is not None else range(t.C))]
                base += t.real
            assert wm.shape[0] == sum(v >= 0 for v in rows) and wm.shape[1] == base, (o["name"], tuple(wm.shape), base)
            wk = wp[:cout, :k * k * cin].view(cout, k * k, cin)
            for i, ri in enumerate(rows):
                if ri < 0:
                    assert torch.all(wk[i] == 0) and bp[i] == 0, o["name"]
                    continue
                assert abs(float(bp[i] - bm[ri])) < 1e-6
                for j, cj in enumerate(cols):
                    got = wk[i, :, j]
                    if cj < 0:
                        assert torch.all(got == 0), (o["name"], i, j)
                    else:
                        assert torch.equal(got, wm[ri, cj].reshape(-1).half().float()), (o["name"], i, j)
        assert padded >= 3
        for o in p.ops:  # pieces stay aligned
            for key in ("src0", "src1", "dst", "res"):
                t = o.get(key)
                if t is not None and t.buf < L.BSY_EXT_BASE and not t.f32:
                    assert t.coff % 8 == 0 and t.C % 8 == 0 or o["kind"] not in (L.OP_CONV, L.OP_BNECK), (o["name"], key, t)
    finally:
        R.SCALES[fam].pop("t", None)


def test_plan_says_which_precisions_run_a_width():
    """YOLO11 at a width multiple of 0.3125: C2PSA's attention heads are 40 / 80 channels wide (block.py:4253-4258), which the fp16 attention
    kernel does not take -- the plan says so when it is built (plugin.graph_support relies on that), the fp32-storage modes build."""
    cfg = dict(stock_cfg("yolo11", "n"), scale="t", scales={"t": [0.5, 0.3125, 1024]})
    with pytest.raises(NotImplementedError, match="attention heads"):
        Plan(cfg, 1, 64, 64)
    Plan(cfg, 1, 64, 64, precision="fp32")
    Plan(cfg, 1, 64, 64, precision="fp32x")
    Plan(dict(cfg, scales={"t": [0.5, 0.1875, 1024]}), 1, 64, 64)  # 48 / 96: on the fp16 path


def test_plan_rejects_unsupported_graphs():
    cfg = stock_cfg("yolo11", "n")
    cfg["backbone"] = list(cfg["backbone"])
    cfg["backbone"][2] = [-1, 2, "C3k2_LRSA", [256, False, 0.25]]  # one of the fork's dead experiments: not accelerated
    with pytest.raises(NotImplementedError):
        Plan(cfg, 1, 64, 64)
    cfg = stock_cfg("yolo11", "n")
    cfg["head"] = list(cfg["head"])
    cfg["head"][-1] = [[16, 19, 22], 1, "Pose", ["nc", [17, 3]]]  # other task heads stay on the reference
    with pytest.raises(NotImplementedError):
        Plan(cfg, 1, 64, 64)


def test_segment_plan_matches_reference_work():
    """YOLOv8l-seg @640 (BASELINE config 4): 210.4 GFLOP/image, protos at (160, 160)."""
    p = Plan(stock_cfg("yolov8", "l", 80, "segment"), 1, 640, 640)
    assert abs(p.flops / 1e9 - 210.4) < 0.5
    assert p.meta["proto_hw"] == (160, 160) and p.meta["nm"] == 32 and p.meta["A"] == 8400
    m = R.Model("yolov8", "l", 80, "segment")
    want = {n for n, _ in m.param_specs() if not n.endswith("dfl.conv.weight")}
    used = set()
    for r in p.wrecs.values():
        if r.kind in ("plain", "deconv"):
            used |= {r.name + ".weight", r.name + ".bias"}
        else:
            used |= {r.name + ".conv.weight"} | {f"{r.name}.bn.{s}" for s in ("weight", "bias", "running_mean", "running_var")}
    assert used == want


def test_fold_matches_reference_known_answer():
    z = np.load(GOLDEN / "modules.npz")
    c = R.Conv("m", 8, 12, 3, 1)
    sd = {n: R.synth_param(n, s, 11) for n, s in c.specs()}
    w, b = fold_conv_bn(sd, "m")
    np.testing.assert_allclose(w.numpy(), z["fuse.w"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(b.numpy(), z["fuse.b"], rtol=1e-6, atol=1e-7)
    # an already-fused state_dict (what the model holds after BaseModel.fuse) gives the same
    fused = {"m.conv.weight": torch.from_numpy(z["fuse.w"]), "m.conv.bias": torch.from_numpy(z["fuse.b"])}
    w2, b2 = fold_conv_bn(fused, "m")
    assert torch.equal(w2, fused["m.conv.weight"]) and torch.equal(b2, fused["m.conv.bias"])


def test_packed_weight_layout_is_the_gemm_the_kernel_runs():
    """[CoutPad][Kpad] with K = (kh, kw, cin): im2col rows in the kernel's K order times the packed matrix == conv2d."""
    from bs_yolo_amd.plan import WRec
    g = torch.Generator().manual_seed(0)
    cin, cout, k = 16, 24, 3
    sd = {"c.conv.weight": torch.randn(cout, cin, k, k, generator=g), "c.conv.bias": torch.randn(cout, generator=g)}
    wb, bb = pack_record(sd, WRec("c", "conv", cout, cin, k))
    wp = torch.frombuffer(bytearray(wb), dtype=torch.float16).view(128, 160).float()
    bp = torch.frombuffer(bytearray(bb), dtype=torch.float32)
    assert torch.all(wp[cout:] == 0) and torch.all(wp[:, 144:] == 0) and torch.all(bp[cout:] == 0)
    x = torch.randn(1, cin, 6, 5, generator=g)
    xp = F.pad(x, (1, 1, 1, 1)).permute(0, 2, 3, 1)  # NHWC
    rows = []
    for oh in range(6):
        for ow in range(5):
            rows.append(torch.cat([xp[0, oh + kh, ow + kw] for kh in range(3) for kw in range(3)]))
    got = torch.stack(rows) @ wp[:cout, :144].t() + bp[:cout]
    ref = F.conv2d(x, sd["c.conv.weight"].half().float(), sd["c.conv.bias"], 1, 1)[0].permute(1, 2, 0).reshape(30, cout)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)


def test_qkv_permutation_groups_heads():
    p = Plan(stock_cfg("yolo11", "s"), 1, 64, 64)
    r = p.wrecs["model.10.m.0.attn.qkv"]
    nh, kd, hd = 4, 32, 64
    assert sorted(r.perm) == list(range(nh * (2 * kd + hd)))
    assert r.perm[:kd] == list(range(0, kd))                       # head 0 q
    assert r.perm[kd:2 * kd] == list(range(128, 128 + kd))         # head 1 q (reference row 1*128 + 0..31)
    assert r.perm[nh * kd] == kd                                   # first k row of head 0
    assert r.perm[2 * nh * kd] == 2 * kd                           # first v row of head 0
    assert r.perm[2 * nh * kd + hd] == 128 + 2 * kd                # first v row of head 1


def test_pack_blob_offsets_are_aligned_and_disjoint():
    p = Plan(stock_cfg("yolo11", "n"), 1, 64, 64)
    blob = pack_plan_weights(p, synth_state_dict(p, 0))
    spans = []
    for r in p.wrecs.values():
        assert r.w_off % 256 == 0 and r.b_off % 256 == 0 and 0 <= r.w_off < r.b_off < len(blob)
        spans.append((r.w_off, r.b_off))
    assert spans == sorted(spans)


def test_letterbox_geometry_matches_reference_golden():
    z = np.load(GOLDEN / "letterbox.npz")
    for c in json.loads(str(z["cases"])):
        kw = dict(c["kw"])
        new_shape = tuple(kw.pop("new_shape"))
        H2, W2, nw, nh, left, top, _ = LetterBox(new_shape, **kw).geometry(tuple(c["shape"]))
        assert [H2, W2, 3] == c["out_shape"], c
        assert [top, top + nh, left, left + nw] == c["box"], c


def test_shard_bounds():
    assert [e - s for s, e in shard_bounds(70, 8)] == [9, 9, 9, 9, 9, 9, 8, 8]  # SAHI config 5
    assert [e - s for s, e in shard_bounds(256, 8)] == [32] * 8                 # config 3
    b = shard_bounds(5, 8)
    assert b[0] == (0, 1) and b[-1] == (5, 5) and sum(e - s for s, e in b) == 5


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from bs_yolo_amd.parallel import gather_detections, gather_detections_async, shard_bounds
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=rank, world_size=world)
n_items = int(sys.argv[3])
s, e = shard_bounds(n_items, world)[rank]
det = torch.zeros(e - s, 4, 6)
counts = torch.zeros(e - s, dtype=torch.int32)
for i in range(s, e):
    det[i - s, :, 0] = i          # item id in column 0
    det[i - s, :, 4] = rank
    counts[i - s] = i % 5
d, c = gather_detections(det, counts, n_items)
assert d.shape == (n_items, 4, 6) and c.shape == (n_items,)
assert torch.equal(d[:, 0, 0], torch.arange(n_items, dtype=torch.float32)), d[:, 0, 0]
assert torch.equal(c, (torch.arange(n_items) % 5).to(torch.int32))
exp_rank = torch.cat([torch.full((b - a,), float(k)) for k, (a, b) in enumerate(shard_bounds(n_items, world))])
assert torch.equal(d[:, 0, 4], exp_rank)
# two gathers in flight (bench.py overlaps the gather of step k with the forward of step k+1)
h1 = gather_detections_async(det, counts, n_items)
h2 = gather_detections_async(det + 1, counts + 1, n_items)
d1, c1 = h1.wait()
d2, c2 = h2.wait()
assert torch.equal(d1, d) and torch.equal(c1, c) and torch.equal(d2, d + 1) and torch.equal(c2, c + 1)
assert c2.dtype == torch.int32 and d2.shape == (n_items, 4, 6)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("n_items", [8, 7])
def test_gather_detections_gloo_world2(tmp_path, n_items):
    """world_size-2 rehearsal of the N>1 path on CPU (gloo): even and uneven shards come back in global order."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(29500 + os.getpid() % 1000 + n_items)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1")
        procs.append(subprocess.Popen([sys.executable, str(script), str(ROOT), port, str(n_items)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


REF = Path("/root/reference")


@pytest.mark.skipif(not REF.exists(), reason="reference tree only exists in the build container")
def test_plugin_reads_a_live_reference_model():
    """accelerate()'s inputs -- model.yaml + state_dict() of a live reference DetectionModel, before and after
    BaseModel.fuse() -- produce the same packed weights as the oracle's parameters."""
    code = r'''
import sys, types, os, importlib.metadata as md
os.environ.update(YOLO_OFFLINE="true", YOLO_AUTOINSTALL="false", YOLO_CONFIG_DIR="/tmp/yolocfg", YOLO_VERBOSE="false")
sys.dont_write_bytecode = True
os.makedirs("/tmp/yolocfg", exist_ok=True)
class D(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"): raise AttributeError(k)
        return D(self.__name__ + "." + k)
    def __call__(self, *a, **k): return None
for n in ("cv2", "pywt", "pywt.data", "seaborn", "cpuinfo"): sys.modules[n] = D(n)
v = md.version; md.version = lambda n: "0.20.0" if n == "torchvision" else v(n)
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, "/root/reference")
import torch, yaml
from ultralytics.nn.tasks import DetectionModel
from bs_yolo_amd.plugin import cfg_of, model_bn_eps
from bs_yolo_amd.plan import Plan
from bs_yolo_amd.weights import pack_plan_weights
from oracle.yolo_ref import synth_param, Model, synth_params
d = yaml.safe_load(open("/root/reference/ultralytics/cfg/models/11/yolo11-seg.yaml"))
d["head"][-1] = [[16, 19, 22], 1, "Detect", ["nc"]]; d["scale"] = "n"; d["nc"] = 80
m = DetectionModel(d, ch=3, nc=80, verbose=False).eval()
for k, t in m.state_dict().items():
    if not k.endswith("num_batches_tracked"): t.copy_(synth_param(k, t.shape, 0))
cfg = cfg_of(m)
assert abs(model_bn_eps(m) - 1e-3) < 1e-12
p1 = Plan(cfg, 1, 64, 64); b1 = pack_plan_weights(p1, {k: v for k, v in m.state_dict().items()}, model_bn_eps(m))
po = Plan(cfg, 1, 64, 64); bo = pack_plan_weights(po, synth_params(Model("yolo11", "n", 80, "detect"), 0))
assert b1 == bo, "unfused reference state_dict packs differently from the oracle parameters"
m.fuse(verbose=False)
p2 = Plan(cfg, 1, 64, 64); b2 = pack_plan_weights(p2, {k: v for k, v in m.state_dict().items()})
import numpy as np
a1 = np.frombuffer(b1, dtype=np.float16).astype(np.float32); a2 = np.frombuffer(b2, dtype=np.float16).astype(np.float32)
assert len(b1) == len(b2) and np.nanmax(np.abs(a1 - a2)) < 2e-3, "fused state_dict packs differently"
# the BS-YOLO graph itself (cfg/models/11/yolo11.yaml, nc = 12): the live model's yaml + state_dict pack exactly like the
# oracle's parameters, un-fused and fused
mb = DetectionModel("/root/reference/ultralytics/cfg/models/11/yolo11.yaml", ch=3, verbose=False).eval()
for k, t in mb.state_dict().items():
    if not k.endswith("num_batches_tracked"): t.copy_(synth_param(k, t.shape, 0))
cb = cfg_of(mb)
assert cb["nc"] == 12
q1 = Plan(cb, 1, 64, 64); c1 = pack_plan_weights(q1, {k: v for k, v in mb.state_dict().items()}, model_bn_eps(mb))
qo = Plan(cb, 1, 64, 64); co = pack_plan_weights(qo, synth_params(Model("bsyolo11", "n", 12, "detect"), 0))
assert c1 == co, "BS-YOLO: reference state_dict packs differently from the oracle parameters"
assert q1.wrecs["model.15"].coef == qo.wrecs["model.15"].coef and q1.wrecs["model.15"].coef[0] > 0
mb.fuse(verbose=False)
q2 = Plan(cb, 1, 64, 64); c2 = pack_plan_weights(q2, {k: v for k, v in mb.state_dict().items()})
assert len(c1) == len(c2)
print("ok")
'''
    r = subprocess.run([sys.executable, "-c", code, str(ROOT)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.skipif(not REF.exists(), reason="reference tree only exists in the build container")
def test_hooks_survive_the_reference_predictor_and_validator_end_to_end(tmp_path):
    """VERDICT r3 "missing 1": the reference's OWN predictor drives the hooks.  DetectionPredictor.setup_model -> AutoBackend(nn_module)
    .fuse() / .float() (nn/autobackend.py:136-147) -> warm-up forward -> stream_inference (engine/predictor.py:219-317: preprocess ->
    inference -> postprocess) -> Results (models/yolo/detect/predict.py:23-41), with `accelerate`, `install_nms`, `install_preprocess`
    installed, and DetectionValidator.postprocess / _process_batch (models/yolo/detect/val.py:93-103, 209-228) with `install_nms` /
    `install_val_metrics`.  No GPU here: `plugin._on_device` lets CPU tensors through to RECORDING stand-ins that return the oracle's
    results (the library's own kernels are exercised by the GPU suite); what this pins is the plumbing -- the hooks are still bound
    after AutoBackend's fuse() / float(), the engine is called with the tensor the contract promises, the reference accepts the tuple
    it gets back (head.py:74), the NMS hook sees the predict / val keyword sets, Results come out, and the boxes equal the un-hooked
    reference pipeline's."""
    code = r"""
import sys, types, os, importlib.metadata as md
os.environ.update(YOLO_OFFLINE="true", YOLO_AUTOINSTALL="false", YOLO_CONFIG_DIR="/tmp/yolocfg", YOLO_VERBOSE="false")
sys.dont_write_bytecode = True
os.makedirs("/tmp/yolocfg", exist_ok=True)
ROOT, TMP = sys.argv[1], sys.argv[2]
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import letterbox_ref, postproc_ref, val_ref
from oracle import yolo_ref as R
class D(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"): raise AttributeError(k)
        return D(self.__name__ + "." + k)
    def __call__(self, *a, **k): return None
for n in ("pywt", "pywt.data", "seaborn", "cpuinfo"): sys.modules[n] = D(n)
cv2 = D("cv2"); cv2.INTER_LINEAR = 1; cv2.BORDER_CONSTANT = 0
cv2.resize = lambda img, dsize, interpolation=1: letterbox_ref.resize_linear_u8(img, dsize)
def _border(img, top, bottom, left, right, borderType, value=(114, 114, 114)):
    h, w = img.shape[:2]
    out = np.empty((h + top + bottom, w + left + right, img.shape[2]), dtype=img.dtype); out[:] = np.asarray(value, dtype=img.dtype)
    out[top:top + h, left:left + w] = img
    return out
cv2.copyMakeBorder = _border
sys.modules["cv2"] = cv2
tv = types.ModuleType("torchvision"); tv.ops = types.ModuleType("torchvision.ops"); tv.ops.nms = postproc_ref.greedy_nms; tv.__version__ = "0.20.0"
sys.modules["torchvision"] = tv; sys.modules["torchvision.ops"] = tv.ops
v = md.version; md.version = lambda n: "0.20.0" if n == "torchvision" else v(n)
sys.path.insert(0, "/root/reference")
import yaml
from ultralytics.nn.tasks import DetectionModel
from ultralytics.models.yolo.detect import DetectionPredictor, DetectionValidator
from ultralytics.utils import ops as rops
from bs_yolo_amd import plugin

def build():
    d = yaml.safe_load(open("/root/reference/ultralytics/cfg/models/11/yolo11-seg.yaml"))
    d["head"][-1] = [[16, 19, 22], 1, "Detect", ["nc"]]; d["scale"] = "n"; d["nc"] = 80
    m = DetectionModel(d, ch=3, nc=80, verbose=False).eval()
    for k, t in m.state_dict().items():
        if not k.endswith("num_batches_tracked"): t.copy_(R.synth_param(k, t.shape, 0))
    for seq in m.model[-1].cv3: seq[-1].bias.data.add_(-6.0)   # a few dozen anchors above conf 0.25 on the noise image below
    return m

rng = np.random.default_rng(4)
im0 = rng.integers(0, 256, (360, 500, 3), dtype=np.uint8)
args = dict(imgsz=640, conf=0.25, iou=0.7, device="cpu", save=False, verbose=False, half=False, batch=1, project=TMP, name="p")

# ---- the un-hooked reference pipeline -----------------------------------------------------------------------------------
ref_pred = DetectionPredictor(overrides=dict(args))
ref_res = ref_pred(source=[im0], model=build())
ref_boxes = ref_res[0].boxes.data.clone()
assert 3 <= ref_boxes.shape[0] <= 300, ref_boxes.shape

# ---- the same predictor with every hook installed; engine / NMS / letterbox = recording stand-ins over the oracle -------------
log = {"engine": [], "nms": [], "lb": [], "built": []}
oracle = R.Model("yolo11", "n", 80, "detect")
class StubEngine:
    def __init__(self, cfg, sd, device=0, bn_eps=1e-3, precision="fp16"):
        self.precision = precision
        log["built"].append((precision, bn_eps, sorted(sd)[:2], len(sd)))
        # the weights accelerate() hands over are read AFTER AutoBackend's fuse(): every conv comes either fused (conv.weight + conv.bias)
        # or with its BatchNorm (this fork's fuse() leaves the Conv class of nn/Addmodules/conv.py alone: isinstance against
        # nn/modules/conv.py's Conv, tasks.py:210) -- weights.fold_conv_bn takes both
        assert all((k[:-len("conv.weight")] + "conv.bias" in sd) or (k[:-len("conv.weight")] + "bn.weight" in sd) for k in sd if k.endswith(".conv.weight") and ".dfl." not in k)
        self.P = {k: v.clone() for k, v in R.synth_params(oracle, 0).items()}
        for k in self.P:
            if ".cv3." in k and k.endswith(".2.bias"): self.P[k] = self.P[k] - 6.0
    def __call__(self, x):
        assert x.dim() == 4 and x.shape[1] == 3 and x.is_contiguous() and x.dtype == torch.float32 and 0.0 <= float(x.min()) and float(x.max()) <= 1.0
        log["engine"].append(tuple(x.shape))
        with torch.inference_mode():
            y, raws = oracle.forward(self.P, x)
        return y, raws
    def close(self): pass
plugin.YoloEngine = StubEngine
plugin._on_device = lambda t: True
plugin._device_is_gpu = lambda d: True
def nms_stub(prediction, *a, **k):
    log["nms"].append((type(prediction).__name__, a, dict(k)))
    return postproc_ref.non_max_suppression(prediction[0] if isinstance(prediction, (list, tuple)) else prediction, *a, **{kk: vv for kk, vv in k.items() if kk in ("conf_thres", "iou_thres", "classes", "agnostic", "multi_label", "max_det", "nc", "max_nms", "max_wh", "in_place")})
plugin._nms.non_max_suppression = nms_stub
import bs_yolo_amd.letterbox as HLB
def lb_stub(ims, imgsz, half=False, pt=True, stride=32, device="cpu"):
    log["lb"].append((len(ims), tuple(imgsz), half, pt, stride))
    return letterbox_ref.preprocess(ims, imgsz, half=half, pt=pt, stride=stride)
HLB.preprocess = lb_stub

model = build()
plugin.accelerate(model)
hook = model.forward
fused = []
_fuse = model.fuse
model.fuse = lambda *a, **k: (fused.append(1), _fuse(*a, **k))[1]
pred = DetectionPredictor(overrides=dict(args))
pred.setup_model(model)                                     # AutoBackend(weights=<nn.Module>): .to() / .fuse() / .float()
inner = pred.model.model
assert inner is model and hasattr(inner, "_bsy_state") and inner.forward == hook, "accelerate's hook did not survive AutoBackend.__init__"
assert fused == [1] and all(p.dtype == torch.float32 for p in inner.parameters()), "AutoBackend did not fuse() / float() the model"
plugin.install_nms(rops)
plugin.install_preprocess(pred)
y = pred.model.warmup(imgsz=(1, 3, 640, 640))               # a no-op on the CPU device (autobackend.py warmup); the forward it would make:
out = pred.model(torch.zeros(1, 3, 64, 96))
assert isinstance(out, (list, tuple)) and len(out) == 2 and tuple(out[0].shape) == (1, 84, 126) and len(out[1]) == 3
res = pred(source=[im0])
st = inner._bsy_state
assert st["fp32_calls"] == 2 and st["engine_calls"] == 0 and st["fallbacks"] == 0 and st["rebuilds"] == 1, st
assert log["built"][0][0] == "fp32x" and abs(log["built"][0][1] - 1e-3) < 1e-12
assert log["engine"] == [(1, 3, 64, 96), (1, 3, 480, 640)], log["engine"]    # 360 x 500 -> auto letterbox 480 x 640 (minimal rectangle)
assert log["lb"] == [(1, (640, 640), False, True, 32)], log["lb"]
kind, a, k = log["nms"][0]
assert kind in ("list", "tuple") and a[:2] == (0.25, 0.7) and k.get("max_det") == 300 and "classes" in k and "agnostic" in k, (kind, a, k)
boxes = res[0].boxes.data
assert type(res[0]).__name__ == "Results" and res[0].orig_shape == (360, 500)
assert boxes.shape == ref_boxes.shape, (boxes.shape, ref_boxes.shape)
assert torch.equal(boxes[:, 5], ref_boxes[:, 5]) and float((boxes[:, :5] - ref_boxes[:, :5]).abs().max()) < 2e-3, (boxes[:, :5] - ref_boxes[:, :5]).abs().max()

# ---- validator: postprocess (val keyword set) and _process_batch through install_val_metrics ---------------------------------
val = DetectionValidator(save_dir=__import__("pathlib").Path(TMP) / "v", args=dict(imgsz=640, device="cpu", conf=0.001, iou=0.6, plots=False, save_json=False))
val.nc, val.lb = 80, []
with torch.inference_mode():
    yv, rawv = oracle.forward(StubEngine(None, {"model.0.conv.bias": 0}).P, torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(1)))
n0 = len(log["nms"])
dets = val.postprocess([yv.clone(), rawv])
kind, a, k = log["nms"][n0]
assert len(dets) == 2 and k.get("multi_label") is True and k.get("labels") == [] and k.get("max_det") == 300 and a[:2] == (0.001, 0.6), (a, k)
import bs_yolo_amd.val as HV
calls = []
def pb_stub(detections, gt_bboxes, gt_cls, iouv):
    calls.append((tuple(detections.shape), tuple(gt_bboxes.shape)))
    return torch.from_numpy(val_ref.process_batch(detections.numpy(), gt_bboxes.numpy(), gt_cls.numpy(), iouv.numpy()))
HV.process_batch = pb_stub
val.iouv = torch.linspace(0.5, 0.95, 10)
orig_pb = val._process_batch
plugin.install_val_metrics(val)
d0 = torch.cat(dets)[:50]
ng = min(5, d0.shape[0])
assert ng >= 1
gt = d0[:ng, :4].clone() + 1.0
tp_hook = val._process_batch(d0, gt, d0[:ng, 5].clone())
tp_ref = orig_pb(d0, gt, d0[:ng, 5].clone())
assert calls == [((d0.shape[0], 6), (ng, 4))], calls
assert tp_hook.shape == (d0.shape[0], 10) and tp_hook.dtype == torch.bool, (tp_hook.shape, tp_hook.dtype)
assert torch.equal(tp_hook, tp_ref.to(tp_hook.device)), (tp_hook.int().sum(0), tp_ref.int().sum(0))
print("ok", boxes.shape[0], "detections")
"""
    r = subprocess.run([sys.executable, "-c", code, str(ROOT), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]


# ---- build / ISA guards (need hipcc's binutils only, no GPU) ----------------------------------------------------------------
def test_isa_has_no_packed_f32():
    """The concurrency-hazard mitigation of DESIGN.md section 7, enforced on the SHIPPED code objects: no kernel of
    libbsyolo_hip.so carries packed f32 arithmetic (v_pk_fma/mul/add_f32) -- hence none can pair it with SDWA f16->f32
    converts -- and no conv / fused-conv kernel has a private (scratch) frame."""
    sys.path.insert(0, str(ROOT / "tools"))
    import isa_scan
    bad, n, scratch = isa_scan.scan()
    assert n >= 130, f"only {n} kernels found: the disassembly failed"
    assert not bad, f"packed f32 arithmetic in {bad}"
    fam = ("conv_mfma_kernel", "conv3x3_patch", "conv1x1_persist", "conv1x1_wres", "c3k2_fused", "bneck_fused", "stem_fused", "dwpw_fused", "conv_first_mfma")
    conv_scratch = [k for k in scratch if any(f in k for f in fam)]
    assert not conv_scratch, f"scratch frames in {conv_scratch}"


def test_build_signature_tracks_flags_and_headers(tmp_path):
    """bs_yolo_amd/build.py rebuilds an object when its command line or ANY header of csrc/ / include/ changes (ADVICE r1:
    the correctness workaround is a flag, and image_conv.h was not tracked)."""
    from bs_yolo_amd import build as Bd
    src = tmp_path / "a.hip"
    hdr = tmp_path / "h.h"
    src.write_text("int f();")
    hdr.write_text("// v1")
    s0 = Bd._signature(["hipcc", "gfx950", "-fno-slp-vectorize", "a.hip"], src, [hdr])
    assert s0 == Bd._signature(["hipcc", "gfx950", "-fno-slp-vectorize", "a.hip"], src, [hdr])
    assert s0 != Bd._signature(["hipcc", "gfx950", "a.hip"], src, [hdr])
    hdr.write_text("// v2")
    assert s0 != Bd._signature(["hipcc", "gfx950", "-fno-slp-vectorize", "a.hip"], src, [hdr])
    hdrs = sorted(p.name for p in Bd.CSRC.glob("*.h"))
    assert "image_conv.h" in hdrs and "common.h" in hdrs


def test_stale_library_is_refused(monkeypatch):
    """bs_yolo_amd.lib refuses an in-tree library that was not built from the current sources (bs_yolo_amd/build.py stale_sources: object
    signatures = flags + source + every header): a source that stops compiling must not leave every test running yesterday's .so."""
    import importlib
    import bs_yolo_amd.build as BLD
    assert BLD.stale_sources() == []          # the tree the tests run on is up to date
    real = BLD._signature
    monkeypatch.setattr(BLD, "_signature", lambda cmd, src, headers: real(cmd, src, headers) + ("x" if src.name == "nms.hip" else ""))
    assert BLD.stale_sources() == ["nms.hip"]
    import bs_yolo_amd.lib as LL
    with pytest.raises(ImportError, match="not built from the current sources"):
        LL._load()
    monkeypatch.setenv("BSY_ALLOW_STALE_LIB", "1")
    assert LL._load() is not None
    importlib.reload(BLD)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself (it exited asking for
    torch.distributed.run).  Without a GPU every rank stops at bench.py's own "needs a ROCm GPU" check -- which proves that
    the child launcher ran bench.py under WORLD_SIZE = 2 (the parent never gets that far: it only relays)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    if torch.cuda.is_available():
        pytest.skip("on a GPU box this would start a real 2-rank run")
    assert r.returncode != 0
    # at least one child rank got as far as the check (the launcher tears the other down when the first exits) and saw WORLD_SIZE = 2
    import re
    assert re.search(r"bench\.py needs a ROCm GPU.*\(rank [01] of 2\)", r.stderr), r.stderr[-2000:]


def test_weights_fingerprint_accepts_inference_tensors():
    """plugin._weights_version on parameters created under torch.inference_mode() (what fuse() produces inside the
    predictor's smart_inference_mode, engine/predictor.py:219): they have no version counter -- reading `_version` raises --
    so the fingerprint falls back to their storage pointers; ordinary tensors still count in-place updates."""
    from bs_yolo_amd import plugin
    with torch.inference_mode():
        conv = torch.nn.Conv2d(4, 4, 1)
    assert all(p.is_inference() for p in conv.parameters())
    with pytest.raises(RuntimeError):
        conv.weight._version
    f0 = plugin._weights_version(conv)
    assert f0 == plugin._weights_version(conv)
    conv.half()
    assert plugin._weights_version(conv) != f0      # new storages
    lin = torch.nn.Linear(3, 3)
    f1 = plugin._weights_version(lin)
    with torch.no_grad():
        lin.weight.mul_(2.0)
    assert plugin._weights_version(lin) != f1       # in-place update of an ordinary tensor
