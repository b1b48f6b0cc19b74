"""Replay ONE output element of an engine GEMM on csrc/mfma_probe.hip MFMA by MFMA (round 6): tile t of the probe input holds the first t + 1
instructions of the element's accumulation chain (the rest zero operands: an MFMA whose products are all zero returns its accumulator), so the
hardware's running sum after every instruction can be held against the model's -- to find the instruction where they part.

    python tools/mfma_chain_replay.py make <dump.npz> <out dir>      (build container: writes chain_*.bf16_32.in.bin + chain_meta.npz)
    build/mfma_probe bf16_32 <in.bin> <out.bin>                       (MI355X)
    python tools/mfma_chain_replay.py check <out dir>                 (build container)"""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mfma_model as MM  # noqa: E402
import mfma_probe_cases as G  # noqa: E402
from pdb2reaction_amd import weights as W  # noqa: E402

ORDER = [(0, 2), (1, 1), (2, 0), (0, 1), (1, 0), (0, 0)]            # the one-accumulator product order of umx_gemm_q.h (LS = 0)


def bf(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def planes_a(x):                                                     # qf_split2: nearest bf16, three times
    p0 = bf(x); p1 = bf(x - p0); p2 = bf(x - p0 - p1)
    return [p0, p1, p2]


def planes_w(w):                                                     # umx_load_weights with aligned planes: quantum 2^(e_max - 12) per group of 8, planes 0 and 1
    out = [np.zeros_like(w) for _ in range(3)]
    rem = w.astype(np.float32).copy()
    for q in range(3):
        for g in range(0, len(w), 8):
            r = rem[g:g + 8]
            gm = np.abs(r).max()
            lead = r.copy()
            if q < 2 and gm > 0:
                quantum = np.float32(2.0 ** (np.floor(np.log2(gm)) - 12))
                lead = (np.rint(r / quantum) * quantum).astype(np.float32)
            out[q][g:g + 8] = bf(lead)
        rem = (rem - out[q]).astype(np.float32)
    return out


CASES = [("conv1_m1_re_P", "y1", 768, 512, "so2_conv_1.so2_m_conv.0.fc.weight", 0, 1310, 64),      # re . Wa
         ("conv1_m1_re_S", "y1", 1280, 512, "so2_conv_1.so2_m_conv.0.fc.weight", 256, 1310, 64)]    # im . Wb


def make(dump, out):
    os.makedirs(out, exist_ok=True)
    d = np.load(dump)
    w = W.make_synthetic_weights(1)
    meta = {}
    for name, src, off, K, wname, wrow0, row, col in CASES:
        a = d[src][row, off:off + K].astype(np.float32)              # (un-negated operand; row parity even -> sign +1)
        sg = -1.0 if row % 2 else 1.0
        ap = planes_a((sg * a).astype(np.float32))
        wp = planes_w(w[f"blocks.1.edge_wise.{wname}"][wrow0 + col].astype(np.float32))
        ops = [(kt, qa, qb) for kt in range(K // 16) for qa, qb in ORDER]
        T = len(ops)
        A = np.zeros((T, T, 32, 16), np.float32); B = np.zeros((T, T, 32, 16), np.float32)
        for t in range(T):
            for s_, (kt, qa, qb) in enumerate(ops[:t + 1]):
                A[t, s_, 0] = ap[qa][kt * 16:kt * 16 + 16]; B[t, s_, 0] = wp[qb][kt * 16:kt * 16 + 16]
        C0 = np.zeros((T, 32, 32), np.float32)
        G.write(os.path.join(out, f"chain_{name}.bf16_32.in.bin"), A, B, C0, "bf16_32")
        meta[name + ".ap"] = np.stack(ap); meta[name + ".wp"] = np.stack(wp)
    np.savez(os.path.join(out, "chain_meta.npz"), **meta)
    print("wrote", out)


def check(out):
    import ctypes as C
    lib = MM.load_lib()
    fp = C.POINTER(C.c_float)
    meta = np.load(os.path.join(out, "chain_meta.npz"))
    for name, src, off, K, wname, wrow0, row, col in CASES:
        ap, wp = meta[name + ".ap"], meta[name + ".wp"]
        ops = [(kt, qa, qb) for kt in range(K // 16) for qa, qb in ORDER]
        hw = np.fromfile(os.path.join(out, f"chain_{name}.bf16_32.out.bin"), np.float32).reshape(len(ops), 32, 32)[:, 0, 0]
        acc = np.float32(0.0)
        first = None
        for t, (kt, qa, qb) in enumerate(ops):
            a = np.ascontiguousarray(ap[qa][kt * 16:kt * 16 + 16]); b = np.ascontiguousarray(wp[qb][kt * 16:kt * 16 + 16])
            prev = acc
            acc = np.float32(lib.mfma_dot(C.c_float(float(acc)), a.ctypes.data_as(fp), b.ctypes.data_as(fp), 16, 8))
            if acc != hw[t] and first is None:
                first = t
                print(f"{name}: model and hardware part at instruction {t} (k-tile {kt}, planes {qa},{qb}): acc before {prev!r}, model {acc!r}, hardware {hw[t]!r}")
                print("   a =", a.tolist()); print("   b =", b.tolist())
                for hh in (0, 8):
                    r = lib.mfma_pass8(C.c_float(float(prev)) if hh == 0 else C.c_float(float(r0)), a[hh:].ctypes.data_as(fp), b[hh:].ctypes.data_as(fp), 8)
                    if hh == 0:
                        r0 = r
                    print(f"   model after pass k = {hh}..{hh + 7}: {np.float32(r)!r}")
                acc = hw[t]                     # continue from the hardware's value: are there more?
        print(f"{name}: {len(ops)} instructions, final model {acc!r} hardware {hw[-1]!r}; first difference at {first}")


if __name__ == "__main__":
    if sys.argv[1] == "make":
        make(sys.argv[2], sys.argv[3])
    else:
        check(sys.argv[2])
