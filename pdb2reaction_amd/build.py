"""Build libumx.so for gfx950 with hipcc (in-tree, so the .so travels with the repository snapshot)."""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libumx.so")
SOURCES = ["umx_api.hip"]
ABI_HEADER = os.path.normpath(os.path.join(HERE, "..", "include", "umx.h"))
# -fno-slp-vectorize: hipcc's SLP vectoriser turns adjacent float operations into packed-fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 /
# v_pk_fma_f32).  On gfx950 with this toolchain (ROCm 7.2) a kernel made of them is TIMING-SENSITIVE: alone on the chip it is bitwise
# reproducible, but while other kernels share its SIMDs -- a second stream, another process -- single waves come out 0.1-1 % off (round 3:
# k_norm_bwd, reproduced stand-alone in csrc/norm_bwd_repro.hip: 27-788 of 40 000 launches beside four busy processes, 0 of 80 000 without
# the packed instructions; NOTES.md section 5 item 14).  The flags are part of the source digest.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize"]


def dependencies() -> list:
    """Every file the library is compiled from: all of csrc/*.h and csrc/*.hip (umx_api.hip includes the headers; a
    hand-kept list once missed the default forward GEMM, so the stale git-ignored .so shipped) plus the ABI header."""
    deps = sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip")))
    deps = [d for d in deps if os.path.basename(d) not in ("gemm_bench.hip", "overlap_bench.hip", "func_bias.hip", "norm_bwd_repro.hip", "corun_probe.hip", "mfma_bias.hip", "mfma_probe.hip", "mfma_rate.hip", "f8_inner_sum.hip")]     # stand-alone dev benchmarks, not part of the library
    return deps + [ABI_HEADER]


def source_digest() -> str:
    """sha256 over names and contents of every file the library is compiled from.  It is compiled INTO the library
    (``umx_build_digest()``), written next to it (``libumx.so.digest``) and recorded with every committed PMC summary, so a
    stale prebuilt .so or a stale traffic figure is detected by content rather than by mtime."""
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode() + b"\0")
    for d in dependencies():
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def find_hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    if any(os.path.getmtime(d) > t for d in dependencies()):
        return True
    try:                                   # content check: a checkout can reset mtimes without changing the prebuilt .so
        with open(OUT + ".digest") as f:
            return f.read().strip() != source_digest()
    except OSError:
        return True


LLVM_BIN = "/opt/rocm/lib/llvm/bin"
PACKED_FP32 = r"\bv_pk_(mul|add|fma)_f32\b"


def device_disassembly(lib_path: str = OUT) -> str:
    """llvm-objdump -d of the gfx950 code object inside a built library: the .hip_fatbin section is cut out (llvm-objcopy), the
    offload bundle unbundled (clang-offload-bundler) and the device ELF disassembled -- the code that actually ships, every kernel
    of every header, not a re-compilation of some of them."""
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib_path, os.path.join(td, "copy.so")], check=True)
        subprocess.run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        return subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout


def check_no_packed_fp32(lib_path: str = OUT) -> int:
    """Fail when the shipped device code contains a packed-fp32 VALU instruction (timing-dependent results on gfx950 while other
    kernels share the SIMDs, NOTES.md section 5 item 14; profiles/r04_k_norm_bwd_isa_diff.txt): the SLP vectoriser is off, but the
    loop vectoriser, explicit float2 arithmetic or a compiler update could bring them back.  Returns the number of kernels checked."""
    import re

    text = device_disassembly(lib_path)
    hits = re.findall(PACKED_FP32, text)
    kernels = len(re.findall(r"^[0-9a-f]+ <[^>]+>:", text, flags=re.M))
    if hits:
        raise RuntimeError(f"{lib_path}: {len(hits)} packed-fp32 instructions (v_pk_mul/add/fma_f32) in the gfx950 code object -- "
                           "build.FLAGS must keep them out (-fno-slp-vectorize); see NOTES.md section 5 item 14")
    if kernels == 0:
        raise RuntimeError(f"{lib_path}: no kernel found in the gfx950 code object (disassembly failed?)")
    return kernels


def build_library(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    digest = source_digest()
    cmd = [find_hipcc(), *FLAGS, "-fPIC", "-shared", "-I", CSRC,
           f'-DUMX_SRC_DIGEST="{digest}"', *[os.path.join(CSRC, s) for s in SOURCES], "-o", OUT]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    check_no_packed_fp32(OUT)                    # on the linked library itself; raises and leaves no digest file behind
    with open(OUT + ".digest", "w") as f:
        f.write(digest + "\n")
    return OUT


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(OUT)
