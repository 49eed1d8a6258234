"""Scan the gfx950 ISA of every kernel of the library for the instruction pattern DESIGN.md section 7 retired: an f16 -> f32
convert with an SDWA half select (v_cvt_f32_f16_sdwa) in a kernel that also uses packed f32 arithmetic (v_pk_fma / mul /
add_f32) -- and, stricter, any packed f32 arithmetic outside msca_spatial_kernel.  Prints the offenders and exits 1 if there
is one.  Runs on the CPU (hipcc cross-compiles); ~3 minutes.
Usage: python tools/isa_scan.py"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bs_yolo_amd.build import ARCH, COMMON, CSRC, SOURCES, _hipcc  # noqa: E402

bad = []
for src, extra in SOURCES.items():
    if not src.endswith(".hip"):
        continue
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-x", "hip", *COMMON, *extra, "-S", "--cuda-device-only",
               f"-I{ROOT / 'include'}", "-o", f.name, str(CSRC / src)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(f"{src}: {r.stderr[-400:]}")
        asm = Path(f.name).read_text()
    n = 0
    for m in re.finditer(r"^(_Z\w+):.*?s_endpgm", asm, re.S | re.M):
        n += 1
        body = m.group(0)
        sd = len(re.findall(r"v_cvt_f32_f16_sdwa", body))
        pk = len(re.findall(r"v_pk_(?:fma|mul|add)_f32", body))
        if sd and pk:
            bad.append((src, m.group(1), sd, pk))
        elif pk and "msca_spatial_kernel" not in m.group(1):  # the one kernel with hand-written packed FMAs (f32 LDS operands; it
            bad.append((src, m.group(1), sd, pk))              # never runs beside other kernels)
    print(f"{src}: {n} kernels scanned")
for b in bad:
    print("OFFENDER %s %s: %d SDWA f16->f32 converts beside %d packed f32 ops" % b)
sys.exit(1 if bad else 0)
