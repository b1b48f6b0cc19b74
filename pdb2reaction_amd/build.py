"""Build libumx.so for gfx950 with hipcc (in-tree, so the .so travels with the repository snapshot)."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libumx.so")
SOURCES = ["umx_api.hip"]
ABI_HEADER = os.path.normpath(os.path.join(HERE, "..", "include", "umx.h"))


def dependencies() -> list:
    """Every file the library is compiled from: all of csrc/*.h and csrc/*.hip (umx_api.hip includes the headers; a
    hand-kept list once missed the default forward GEMM, so the stale git-ignored .so shipped) plus the ABI header."""
    deps = sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip")))
    deps = [d for d in deps if os.path.basename(d) != "gemm_bench.hip"]     # stand-alone benchmark, not part of the library
    return deps + [ABI_HEADER]


def find_hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in dependencies())


def build_library(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    cmd = [find_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", CSRC,
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", OUT]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(OUT)
