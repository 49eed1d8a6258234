"""Build libbsyolo_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m bs_yolo_amd.build [--force]
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libbsyolo_hip.so"
ARCH = "gfx950"
COMMON = ["-fno-slp-vectorize"]
# per-file extra flags: NMS / letterbox decisions must round exactly like the fp32 CPU reference (no FMA contraction);
# COMMON -fno-slp-vectorize: the SLP vectoriser turns adjacent scalar f32 operations into packed v_pk_*_f32 instructions;
# fed by SDWA f16 converts that form returned wrong odd elements beside MFMA kernels of other streams (csrc/common.h,
# DESIGN.md section 7), and it buys nothing measurable anywhere (the headline forward: 3.572 vs 3.580 ms, three runs each)
SOURCES = {
    "err.cpp": [],
    "conv_mfma.hip": [],
    "conv_first.hip": [],
    "stem_fused.hip": [],
    "bneck_fused.hip": [],
    "c3k2_fused.hip": [],
    "chain1x1.hip": [],
    "bsyolo_ops.hip": [],
    "pmsfa_fused.hip": [],
    "elementwise.hip": [],
    "attention.hip": [],
    "detect.hip": [],
    "nms.hip": ["-ffp-contract=off"],
    "letterbox.hip": ["-ffp-contract=off"],
    "masks.hip": ["-ffp-contract=off"],
    "val_match.hip": ["-ffp-contract=off"],
    "val_ap.hip": ["-ffp-contract=off"],
    "sahi.hip": ["-ffp-contract=off"],
    "ref32.hip": ["-ffp-contract=off", "-fno-vectorize"],  # loop vectoriser: packed f32 math in attn32_kernel otherwise
    "conv32_mfma.hip": ["-ffp-contract=off", "-fno-vectorize"],  # the same flags: its SiLU must be ref32.hip's bit for bit
    "conv32x_mfma.hip": ["-fno-vectorize"],  # fp32x mode: split-f16 operands on the fp16 matrix pipe
    "attention32x.hip": ["-fno-vectorize"],
    "engine.hip": [],
}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(out: Path, deps) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def _signature(cmd, src: Path, headers) -> str:
    """What an object file was built from: the exact command line plus the contents of its source and of EVERY header of
    csrc/ and include/ (a flag change -- the correctness workaround -fno-slp-vectorize is one -- or an edit of any header
    rebuilds; mtimes are not consulted)."""
    h = hashlib.sha256()
    h.update("\0".join(cmd).encode())
    for f in (src, *headers):
        h.update(b"\0" + f.name.encode() + b"\0")
        h.update(f.read_bytes())
    return h.hexdigest()


def stale_sources() -> list:
    """Sources of SOURCES whose object file was NOT built from their current text (or from the current flags / headers): the in-tree
    library then is not the one the tree describes.  Empty when the objects directory does not travel with the library (nothing to
    compare) or everything is up to date.  bs_yolo_amd/lib.py refuses a stale library at import: a source that no longer compiles
    otherwise leaves yesterday's .so in place and every test and measurement silently runs it (round 4: two experiments did)."""
    objdir = CSRC / "build"
    if not objdir.is_dir() or not LIB.exists() or not any(objdir.glob("*.sig")):
        return []  # (no signature travelled with the library: nothing to compare)
    headers = sorted(CSRC.glob("*.h")) + sorted((PKG.parent / "include").glob("*.h"))
    # the compiler's path is part of a signature; the box that loads the library may name it differently than the one that built it
    compilers = {"/opt/rocm/bin/hipcc", "hipcc"}
    try:
        compilers.add(_hipcc())
    except RuntimeError:
        pass
    out = []
    try:
        for src, extra in SOURCES.items():
            if not (CSRC / src).exists():
                out.append(src)
                continue
            stamp = objdir / (src.rsplit(".", 1)[0] + ".sig")
            obj = stamp.with_suffix(".o")
            if not stamp.exists() or not obj.exists():  # (build_library removes the stamp before it compiles)
                out.append(src)
                continue
            have = stamp.read_text()
            if not any(have == _signature([cc, ARCH, *COMMON, *extra, src], CSRC / src, headers) for cc in compilers):
                out.append(src)
    except OSError:
        return []  # the guard must never be the reason an intact library does not load
    return out


def build_library(force: bool = False, verbose: bool = False) -> Path:
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)
    headers = sorted(CSRC.glob("*.h")) + sorted((PKG.parent / "include").glob("*.h"))
    hipcc = _hipcc()
    jobs = []
    for src, extra in SOURCES.items():
        obj = objdir / (src.rsplit(".", 1)[0] + ".o")
        stamp = obj.with_suffix(".sig")
        # paths relative to the package: the signature must not depend on where the tree is checked out
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip", *COMMON, *extra, "-c",
               str(CSRC / src), "-o", str(obj)]
        sig = _signature([hipcc, ARCH, *COMMON, *extra, src], CSRC / src, headers)
        if force or not obj.exists() or not stamp.exists() or stamp.read_text() != sig:
            jobs.append((cmd, stamp, sig))
    def run(job):
        cmd, stamp, sig = job if isinstance(job, tuple) else (job, None, None)
        if verbose:
            print(" ".join(cmd), flush=True)
        if stamp is not None and stamp.exists():
            stamp.unlink()
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if stamp is not None:
            stamp.write_text(sig)
    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [str(objdir / (s.rsplit(".", 1)[0] + ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *objs])
    return LIB


if __name__ == "__main__":
    p = build_library(force="--force" in sys.argv, verbose=True)
    print("built", p, p.stat().st_size, "bytes")
