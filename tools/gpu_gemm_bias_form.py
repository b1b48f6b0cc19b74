"""What FORM does the systematic error of the split-precision GEMM outputs have?  For rad / hg / msg of layer 0 (engine vs the staged float64
oracle) regress the difference d = a - r on [1, r, |r|] per column block (the m = 0 / m = 1 / m = 2 sub-GEMMs): a constant is a one-sided
(floor-like) rounding, a coefficient on r a magnitude truncation (toward zero), on |r| a one-sided error that scales with the accumulator.

    UMX_PRECISION=bf16x3 python tools/gpu_gemm_bias_form.py [n_atoms]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from pdb2reaction_amd import synth, weights as W  # noqa: E402
from pdb2reaction_amd.engine import Engine  # noqa: E402
from oracle.staged import Staged  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
torch.set_num_threads(16)
w = W.make_synthetic_weights(0)
z, pos = synth.make_cluster(n)
pos32 = pos.astype(np.float32)
st = Staged(w)
st.forward(z, pos32.astype(np.float64))
T = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in st.t.items()}
eng = Engine(0)
eng.load_weights(w)
eng.set_system(z)
eng.debug_keep(True)
eng.energy_forces(pos32, forces=False)
ne = len(T["src"])
print(f"mode {eng.precision_mode()}  N = {n}  edges = {ne}")


def fit(name, a, r, blocks):
    a = a.reshape(ne, -1).astype(np.float64); r = np.asarray(r, dtype=np.float64).reshape(ne, -1)
    for bname, lo, hi in blocks:
        aa, rr = a[:, lo:hi].reshape(-1), r[:, lo:hi].reshape(-1)
        d = aa - rr
        x = np.stack([np.ones_like(rr), rr, np.abs(rr)], axis=1)
        coef, *_ = np.linalg.lstsq(x, d, rcond=None)
        print(f"{name:8s} {bname:14s} rms(r) {np.sqrt((rr * rr).mean()):.3f} mean(r) {rr.mean():+.3f}  mean d {d.mean():+.2e}  fit: const {coef[0]:+.2e}  *r {coef[1]:+.2e}  *|r| {coef[2]:+.2e}   rms d {np.sqrt((d * d).mean()):.2e}")


for i in (0, 3):
    hg = np.concatenate([T[f"gate.{i}"], T[f"hpre.{i}"].reshape(ne, -1)], axis=1)
    fit(f"rad.{i}", eng.debug_fetch(f"rad.{i}"), T[f"rad.{i}"], [("all (K=128)", 0, 1536)])
    fit(f"hg.{i}", eng.debug_fetch(f"hg.{i}"), hg, [("m0 (K=768)", 0, 640), ("m1 (K=512)", 640, 1152), ("m2 (K=256)", 1152, 1408)])
    fit(f"msg.{i}", eng.debug_fetch(f"msg.{i}"), T[f"msg.{i}"], [("m0 (K=384)", 0, 384), ("m0 row l0", 0, 128), ("m0 rows l1,l2", 128, 384), ("m1 (K=256)", 384, 896), ("m2 (K=128)", 896, 1152)])
