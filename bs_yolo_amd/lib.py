"""ctypes binding of libbsyolo_hip.so (C ABI declared in include/bsyolo.h).

The library is the product: importing this module fails loudly when the shared object is missing or does not
export every symbol of the header -- there is no CPU / PyTorch fallback behind it.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import torch  # noqa: F401  -- FIRST: the process must hold ONE HIP runtime (torch's libamdhip64); loading ours before
#                               torch would bring in a second copy that sees no device

PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["BSY_LIB"]) if os.environ.get("BSY_LIB") else PKG / "libbsyolo_hip.so"  # BSY_LIB: A/B builds (tools/)

BSY_F16, BSY_F32, BSY_U8 = 0, 1, 2
BSY_EXT_BASE = 0x100000
(OP_CONV_FIRST, OP_CONV, OP_DWCONV, OP_SPPF_POOL, OP_ATTN, OP_DECODE, OP_RAW_NCHW, OP_NHWC2NCHW, OP_STEM, OP_BNECK, OP_DWCONV_G, OP_COPY, OP_GAP, OP_MSCA_MIX, OP_MUL, OP_ELA, OP_DWPW, OP_MSCA_SPATIAL, OP_C3K2, OP_S2D, OP_PMSFA_TAIL, OP_CHAIN) = range(22)

SYMBOLS = [
    "bsy_engine_create", "bsy_engine_destroy", "bsy_engine_load_weights", "bsy_plan_create", "bsy_plan_create_arena", "bsy_engine_arena_bytes", "bsy_plan_set_tuning", "bsy_plan_destroy",
    "bsy_plan_run", "bsy_plan_graph_launch", "bsy_plan_profile", "bsy_plan_copy_buffer", "bsy_plan_check_guards", "bsy_plan_autotune", "bsy_plan_get_tuning", "bsy_plan_get_tuning_alt", "bsy_plan_check_tuning", "bsy_plan_autotune_in_place", "bsy_conv2d", "bsy_conv2d_f32", "bsy_conv2d_f32x", "bsy_conv_first_f32", "bsy_attention_f32", "bsy_conv_packed_dims", "bsy_conv_first", "bsy_stem_fused", "bsy_stem_fused_supported", "bsy_bottleneck_fused", "bsy_bottleneck_fused_supported", "bsy_c3k2_fused", "bsy_c3k2_fused_supported", "bsy_dwconv", "bsy_dwpw_fused", "bsy_dwpw_fused_supported", "bsy_ela", "bsy_ela_scratch_bytes", "bsy_dwconv3x3",
    "bsy_sppf_pool", "bsy_attention", "bsy_detect_decode", "bsy_nms_workspace_bytes", "bsy_nms", "bsy_scale_boxes",
    "bsy_letterbox", "bsy_process_mask", "bsy_process_mask_native", "bsy_scale_masks", "bsy_val_match", "bsy_slice_tiles", "bsy_sahi_merge_workspace_bytes",
    "bsy_sahi_merge", "bsy_ap_workspace_bytes", "bsy_ap_per_class", "bsy_last_error", "bsy_version", "bsy_sizeof_op",
]


class View(C.Structure):
    _fields_ = [("buf", C.c_int32), ("ld", C.c_int32), ("coff", C.c_int32), ("C", C.c_int32)]


NO_VIEW = (-1, 0, 0, 0)


class Op(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("OH", C.c_int32), ("OW", C.c_int32),
        ("src0", View), ("src1", View),
        ("up0", C.c_int32), ("up1", C.c_int32),
        ("dst", View), ("res", View),
        ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("act", C.c_int32), ("out_f32", C.c_int32),
        ("dst_scale", C.c_int32), ("dst_dy", C.c_int32), ("dst_dx", C.c_int32),
        ("w_off", C.c_int64), ("b_off", C.c_int64),
        ("heads", C.c_int32), ("key_dim", C.c_int32), ("head_dim", C.c_int32),
        ("scale", C.c_float),
        ("nl", C.c_int32), ("nc", C.c_int32), ("nm", C.c_int32), ("A", C.c_int32),
        ("box", View * 3), ("cls", View * 3), ("msk", View * 3),
        ("lvl_h", C.c_int32 * 3), ("lvl_w", C.c_int32 * 3),
        ("lvl_stride", C.c_float * 3),
        ("in_dtype", C.c_int32), ("out_dtype", C.c_int32),
        ("level", C.c_int32),
        ("lane", C.c_int32), ("tuned_cfg", C.c_int32), ("join", C.c_int32),
        ("mid_c", C.c_int32),
        ("w2_off", C.c_int64), ("b2_off", C.c_int64),
        ("aux_off", C.c_int64 * 18),
        ("prec", C.c_int32), ("reserved0", C.c_int32), ("ksplit", C.c_int32), ("reserved1", C.c_int32),
    ]


class BsyError(RuntimeError):
    pass


def _load() -> C.CDLL:
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m bs_yolo_amd.build` (hipcc, gfx950). "
            "bs_yolo_amd has no CPU fallback.")
    if not os.environ.get("BSY_LIB") and os.environ.get("BSY_ALLOW_STALE_LIB") != "1":
        from .build import stale_sources
        stale = stale_sources()
        if stale:
            raise ImportError(f"{LIB_PATH} was not built from the current sources ({', '.join(stale)} changed or failed to compile): run "
                              "`python -m bs_yolo_amd.build` (BSY_ALLOW_STALE_LIB=1 loads it anyway)")
    lib = C.CDLL(str(LIB_PATH))
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise ImportError(f"{LIB_PATH} does not export {missing}; rebuild it")
    if lib.bsy_sizeof_op() != C.sizeof(Op):
        raise ImportError(f"{LIB_PATH}: bsy_op is {lib.bsy_sizeof_op()} bytes in the library, {C.sizeof(Op)} in bs_yolo_amd/lib.py; rebuild it")
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    lib.bsy_last_error.restype = C.c_char_p
    lib.bsy_engine_create.argtypes = [i32, C.POINTER(vp)]
    lib.bsy_engine_destroy.argtypes = [vp]
    lib.bsy_engine_destroy.restype = None
    lib.bsy_engine_load_weights.argtypes = [vp, vp, C.c_size_t]
    lib.bsy_plan_create.argtypes = [vp, C.POINTER(Op), i32, C.POINTER(i64), i32, C.POINTER(vp)]
    lib.bsy_plan_create_arena.argtypes = [vp, C.POINTER(Op), i32, C.POINTER(i64), C.POINTER(i64), i32, i64, C.POINTER(vp)]
    lib.bsy_engine_arena_bytes.argtypes = [vp]
    lib.bsy_engine_arena_bytes.restype = C.c_size_t
    lib.bsy_plan_set_tuning.argtypes = [vp, C.POINTER(C.c_int32), i32]
    lib.bsy_plan_destroy.argtypes = [vp]
    lib.bsy_plan_destroy.restype = None
    lib.bsy_plan_run.argtypes = [vp, C.POINTER(vp), i32, vp]
    lib.bsy_plan_graph_launch.argtypes = [vp, C.POINTER(vp), i32, vp, C.POINTER(i32)]
    lib.bsy_plan_autotune.argtypes = [vp, C.POINTER(vp), i32, vp]
    lib.bsy_plan_get_tuning.argtypes = [vp, C.POINTER(C.c_int32), i32]
    lib.bsy_plan_get_tuning_alt.argtypes = [vp, i32, C.POINTER(C.c_int32), i32]
    lib.bsy_plan_check_tuning.argtypes = [vp, C.POINTER(vp), i32, C.POINTER(C.c_int32), i32]
    lib.bsy_plan_autotune_in_place.argtypes = [vp, C.POINTER(vp), i32, vp, i32]
    lib.bsy_plan_copy_buffer.argtypes = [vp, i32, vp, C.c_size_t]
    lib.bsy_plan_check_guards.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    lib.bsy_plan_profile.argtypes = [vp, C.POINTER(vp), i32, vp, C.POINTER(f32)]
    lib.bsy_conv2d.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, i32, vp]
    lib.bsy_conv2d_f32.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, i32, vp]
    lib.bsy_conv2d_f32x.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, i32, vp]
    lib.bsy_conv_first_f32.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.bsy_attention_f32.argtypes = [vp, i32, i32, i32, i32, i32, i32, f32, vp, i32, i32, vp]
    lib.bsy_conv_packed_dims.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    lib.bsy_conv_first.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.bsy_stem_fused.argtypes = [vp, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, vp, i32, i32, vp]
    lib.bsy_stem_fused_supported.argtypes = [i32, i32, i32, i32]
    lib.bsy_val_match.argtypes = [vp, i32, vp, i32, i32, vp, vp, vp, i32, C.POINTER(C.c_float), i32, vp, vp]
    lib.bsy_bottleneck_fused.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, vp]
    lib.bsy_bottleneck_fused_supported.argtypes = [i32, i32]
    lib.bsy_c3k2_fused.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]
    lib.bsy_c3k2_fused_supported.argtypes = [i32, i32, i32]
    lib.bsy_dwconv.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, i32, i32, vp]
    lib.bsy_ela_scratch_bytes.argtypes = [i32, i32, i32, i32]
    lib.bsy_dwpw_fused.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.bsy_dwpw_fused_supported.argtypes = [i32, i32]
    lib.bsy_ela_scratch_bytes.restype = C.c_size_t
    lib.bsy_ela.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, C.POINTER(C.c_float), vp, vp, i32, vp]
    lib.bsy_dwconv3x3.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, vp, i32, i32, vp, i32, vp]
    lib.bsy_sppf_pool.argtypes = [vp, i32, i32, i32, i32, i32, vp]
    lib.bsy_attention.argtypes = [vp, i32, i32, i32, i32, i32, i32, f32, vp, i32, vp]
    lib.bsy_detect_decode.argtypes = [C.POINTER(vp), C.POINTER(i32), C.POINTER(vp), C.POINTER(i32), C.POINTER(vp),
                                      C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(f32), i32, i32, i32,
                                      i32, vp, i32, vp]
    lib.bsy_nms_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    lib.bsy_nms_workspace_bytes.restype = C.c_size_t
    lib.bsy_nms.argtypes = [vp, i32, i32, i32, i32, i32, f32, f32, vp, i32, i32, i32, i32, i32, f32, i32, vp, vp, vp,
                            C.c_size_t, vp]
    lib.bsy_scale_boxes.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.bsy_process_mask.argtypes = [vp, i32, i32, i32, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp]
    lib.bsy_process_mask_native.argtypes = [vp, i32, i32, i32, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp]
    lib.bsy_scale_masks.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.bsy_letterbox.argtypes = [vp, vp, vp, i32, i32, i32, vp, i32, vp]
    lib.bsy_slice_tiles.argtypes = [vp, i32, i32, i32, vp, i32, i32, i32, i32, vp, i32, vp]
    lib.bsy_ap_workspace_bytes.argtypes = [i32, i32]
    lib.bsy_ap_workspace_bytes.restype = C.c_size_t
    lib.bsy_ap_per_class.argtypes = [vp, vp, vp, i32, i32, vp, vp, i32, vp, vp, C.c_double, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.bsy_sahi_merge_workspace_bytes.argtypes = [i32, i32]
    lib.bsy_sahi_merge_workspace_bytes.restype = C.c_size_t
    lib.bsy_sahi_merge.argtypes = [vp, vp, vp, i32, i32, i32, i32, f32, i32, i32, f32, f32, vp, vp, i32, vp, C.c_size_t, vp]
    return lib


lib = _load()


def check(rc: int) -> None:
    if rc != 0:
        raise BsyError(f"libbsyolo_hip: {lib.bsy_last_error().decode(errors='replace')} (status {rc})")


def dtype_code(t) -> int:
    import torch
    if t == torch.float16:
        return BSY_F16
    if t == torch.float32:
        return BSY_F32
    if t == torch.uint8:
        return BSY_U8
    raise TypeError(f"unsupported dtype {t}")
