"""Scan the gfx950 ISA of every kernel the built library ships for the instruction pattern DESIGN.md section 7 retired: an
f16 -> f32 convert with an SDWA half select (v_cvt_f32_f16_sdwa) in a kernel that also uses packed f32 arithmetic
(v_pk_fma / mul / add_f32) -- and, stricter, any packed f32 arithmetic at all.  Works on the object files of
bs_yolo_amd/csrc/build (what libbsyolo_hip.so was linked from): device code object out of .hip_fatbin ->
llvm-objdump -d -> one record per kernel.  Needs no GPU; seconds.  tests/test_host_logic.py runs `scan()` as a CPU test.
Usage: python tools/isa_scan.py   (prints the offenders, exits 1 if there is one)"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LLVM = Path("/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
PACKED_OK = ()  # kernels allowed to carry hand-written packed f32 arithmetic: none


def disassemble(obj: Path) -> str:
    with tempfile.TemporaryDirectory() as td:
        fat, co = Path(td) / "fat.bin", Path(td) / "dev.co"
        r = subprocess.run([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(obj)], capture_output=True, text=True)
        if r.returncode:
            if "not found" in r.stderr:  # a translation unit without kernels (engine.hip)
                return ""
            raise RuntimeError(r.stderr)
        subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}",
                        f"--output={co}"], check=True, capture_output=True)
        return subprocess.run([str(LLVM / "llvm-objdump"), "-d", str(co)], check=True, capture_output=True, text=True).stdout


def kernels(asm: str):
    """(symbol, body) per function of a disassembly."""
    parts = re.split(r"^[0-9a-f]+ <(_Z\w+)>:$", asm, flags=re.M)
    return list(zip(parts[1::2], parts[2::2]))


def scan(objdir: Path = ROOT / "bs_yolo_amd" / "csrc" / "build"):
    """-> (offenders, number of kernels scanned, {symbol: (scratch bytes proxy: count of scratch_ instructions)})."""
    from bs_yolo_amd.build import SOURCES
    bad, n, scratch = [], 0, {}
    for src in SOURCES:
        if not src.endswith(".hip"):
            continue
        obj = objdir / (src.rsplit(".", 1)[0] + ".o")
        if not obj.exists():
            raise FileNotFoundError(f"{obj}: build the library first (python -m bs_yolo_amd.build)")
        for sym, body in kernels(disassemble(obj)):
            n += 1
            sd = len(re.findall(r"v_cvt_f32_f16_sdwa", body))
            pk = len(re.findall(r"v_pk_(?:fma|mul|add)_f32", body))
            sc = len(re.findall(r"\bscratch_(?:load|store)", body))
            if sc:
                scratch[sym] = sc
            if pk and not any(k in sym for k in PACKED_OK):
                bad.append((src, sym, sd, pk))
    return bad, n, scratch


if __name__ == "__main__":
    sys.path.insert(0, str(ROOT))
    bad, n, scratch = scan()
    print(f"{n} kernels scanned; {len(scratch)} use scratch memory")
    for s, c in sorted(scratch.items()):
        print(f"  scratch: {s} ({c} scratch instructions)")
    for b in bad:
        print("OFFENDER %s %s: %d SDWA f16->f32 converts, %d packed f32 ops" % b)
    sys.exit(1 if bad else 0)
