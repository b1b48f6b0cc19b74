"""CPU emulation (round 3): can the fourth forward product a0 x w2 (third fp16 weight plane) be replaced by a per-column constant?
The energy error of two-plane weights is sum_kn w2[k,n] dE/dW[k,n] to first order -- a fixed number per geometry that grows with the
number of atoms.  Replacing a[e,k] by its column mean abar[k] turns the product into a bias vector abar @ w2 (a GEMV); this measures
how much of the error that removes.  Same matmul patch as tools/precision_study2.py, fp16 planes scaled by 2^(11 q)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W, synth
from oracle.staged import Staged

torch.set_num_threads(8)
orig = torch.Tensor.__matmul__
MODE = {"m": None}


def split16(x, terms):
    parts, r = [], x
    for q in range(terms):
        s = 2.0 ** (11 * q)
        p = (r * s).to(torch.float16).to(torch.float64) / s
        parts.append(p)
        r = r - p
    return parts


def mm(a, b):
    m = MODE["m"]
    if m is None or a.dim() != 2 or b.dim() != 2 or a.shape[0] < 1000:        # edge-level GEMMs only (rows = edges)
        return orig(a, b)
    pa, pb = split16(a, 2), split16(b, 3)
    out = orig(pa[0], pb[0]) + orig(pa[0], pb[1]) + orig(pa[1], pb[0])
    if m == "4":
        out = out + orig(pa[0], pb[2])
    elif m == "3+mean":
        out = out + orig(pa[0].mean(0, keepdim=True), pb[2])
    elif m == "3+mean-all":                                                   # also the (a1, w1) term's mean
        out = out + orig(pa[0].mean(0, keepdim=True), pb[2]) + orig(pa[1].mean(0, keepdim=True), pb[1])
    return out


torch.Tensor.__matmul__ = mm
w = W.make_synthetic_weights(0)
rmsd = 1.5
for n in [int(a) for a in sys.argv[1:]] or [120, 250, 500]:
    z, pos = synth.make_cluster(n)
    pos = pos.astype(np.float32).astype(np.float64)
    st = Staged(w)
    MODE["m"] = None
    e0 = float(st.forward(z, pos))
    line = [f"N={n:5d} edges={len(st.t['src']):7d}"]
    for m in ("4", "3", "3+mean", "3+mean-all"):
        MODE["m"] = m
        e = float(st.forward(z, pos))
        line.append(f"{m}: dE={(e - e0) * rmsd:+.2e} ({(e - e0) * rmsd / n:+.1e}/atom)")
    print(" | ".join(line), flush=True)
