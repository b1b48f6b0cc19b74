"""Bit-exact model of the gfx950 16-bit matrix cores (tools/mfma_emul.c) held against raw hardware results (csrc/mfma_probe.hip).

    bash tools/gpu_mfma_probe.sh                      (on the MI355X, through gpurun: writes gpurun_out/mfma_probe/*.out.bin)
    python tools/mfma_model.py gpurun_out/mfma_probe  (here: regenerates the same operand tiles, runs the C model, counts mismatching BITS)

Also imported by tools/cpu_mfma_gemm_bias.py and tests/test_mfma_model.py (`load_lib`, `gemm_bf16x3`)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import mfma_probe_cases as G  # noqa: E402

_LIB = None


def load_lib():
    """gcc-compile tools/mfma_emul.c into build/libmfma_emul.so (once per source change) and load it."""
    global _LIB
    if _LIB is not None:
        return _LIB
    src, out = os.path.join(HERE, "mfma_emul.c"), os.path.join(ROOT, "build", "libmfma_emul.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-fopenmp", "-o", out, src, "-lm"], check=True)
    lib = C.CDLL(out)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
    lib.mfma_tiles.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.mfma_pass8.argtypes = [C.c_float, fp, fp, C.c_int]
    lib.mfma_pass8.restype = C.c_float
    lib.mfma_dot.argtypes = [C.c_float, fp, fp, C.c_int, C.c_int]
    lib.mfma_dot.restype = C.c_float
    lib.gemm_bf16x3.argtypes = [fp, fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, ip, C.c_int, ip, C.c_int, ip, C.c_int, ip, C.c_int]
    _LIB = lib
    return lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


# accumulation programs of the engine's six-product forward GEMM (umx_gemm_q.h): (qa, qb, accumulator) per k-tile, folds per k-tile / at the end
SCHEMES = {
    # every product into one accumulator, smallest first (LS = 0; UMX_LOW_SEP=0)
    "plain": dict(prog=[(0, 2, 0), (1, 1, 0), (2, 0, 0), (0, 1, 0), (1, 0, 0), (0, 0, 0)], fold_step=[], fold_end=[]),
    # LS = 1 (256 x 256 tiles): the 2^-16-order products chain from zero every k-tile and are folded in with a float32 add
    "ls1": dict(prog=[(2, 0, 1), (1, 1, 1), (0, 2, 1), (1, 0, 0), (0, 1, 0), (0, 0, 0)], fold_step=[(1, 0)], fold_end=[]),
    # LS = 2 (256 x 128 tiles): a second accumulator for the whole k loop
    "ls2": dict(prog=[(0, 2, 1), (1, 1, 1), (2, 0, 1), (0, 1, 0), (1, 0, 0), (0, 0, 0)], fold_step=[], fold_end=[(1, 0)]),
}


def gemm_bf16x3(A, W, bias=None, scheme="plain", cols=None, alt_rows=True):
    """The engine's bf16x3 forward GEMM Y = bias + A . W^T on the modelled matrix core; returns float32 [M, len(cols)]."""
    lib = load_lib()
    A = np.ascontiguousarray(A, np.float32); W = np.ascontiguousarray(W, np.float32)
    M, K = A.shape
    N = W.shape[0]
    assert W.shape[1] == K and K % 16 == 0
    cols = np.arange(N, dtype=np.int32) if cols is None else np.ascontiguousarray(cols, np.int32)
    sch = SCHEMES[scheme] if isinstance(scheme, str) else scheme
    prog = np.ascontiguousarray(np.array(sch["prog"], np.int32).reshape(-1))
    fs = np.ascontiguousarray(np.array(sch["fold_step"], np.int32).reshape(-1)); fe = np.ascontiguousarray(np.array(sch["fold_end"], np.int32).reshape(-1))
    Y = np.empty((M, len(cols)), np.float32)
    rs = np.where(np.arange(M) % 2 == 1, -1.0, 1.0).astype(np.float32) if alt_rows else None
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    lib.gemm_bf16x3(_fp(A), _fp(W), _fp(b) if b is not None else None, _fp(rs) if rs is not None else None, _fp(Y), M, N, K, _ip(cols), len(cols),
                    _ip(prog), len(prog) // 3, _ip(fs) if len(fs) else None, len(fs) // 2, _ip(fe) if len(fe) else None, len(fe) // 2)
    return Y


def check(probe_dir: str) -> int:
    lib = load_lib()
    total_bad = 0
    for kind, sig in (("bf16_32", 8), ("f16_32", 11), ("bf16_16", 8)):
        sets = G.make_sets(kind, 100 + G.KINDS.index(kind))
        for name, (A, B, C0) in sets.items():
            path = os.path.join(probe_dir, f"{name}.{kind}.out.bin")
            if not os.path.exists(path):
                print(f"{kind:8s} {name:7s} (no hardware result at {path})")
                continue
            if kind.startswith("f16"):
                Av, Bv = A.astype(np.float16).astype(np.float32), B.astype(np.float16).astype(np.float32)
            else:
                Av, Bv = G.from_bf16_bits(G.to_bf16_bits(A)), G.from_bf16_bits(G.to_bf16_bits(B))
            Av = np.ascontiguousarray(Av); Bv = np.ascontiguousarray(Bv); C0 = np.ascontiguousarray(C0, np.float32)
            T, steps, R, K = A.shape
            hw = np.fromfile(path, np.float32).reshape(T, R, R)
            out = np.empty_like(hw)
            lib.mfma_tiles(_fp(Av), _fp(Bv), _fp(C0), _fp(out), T, steps, R, K, sig)
            bad = int((out.view(np.uint32) != hw.view(np.uint32)).sum())
            total_bad += bad
            ex = C0.astype(np.float64) + np.einsum("tsik,tsjk->tij", Av.astype(np.float64), Bv.astype(np.float64))
            rne_bad = int((ex.astype(np.float32) != hw).sum())
            print(f"{kind:8s} {name:7s} {hw.size:7d} dot products x {steps} step(s): model != hardware in {bad} results;   "
                  f"(correctly rounded exact sum != hardware in {rne_bad})")
    return total_bad


if __name__ == "__main__":
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "mfma_probe")
    bad = check(d)
    print("TOTAL mismatches:", bad)
    sys.exit(1 if bad else 0)
