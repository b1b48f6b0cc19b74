"""CPU emulation: which of the six bf16 plane products of the forward GEMMs can be dropped?  (round 2)
Energy error vs exact float64 for product sets of the 3-plane x 3-plane split, at several N (is the error ~N or ~sqrt N?).
Patches Tensor.__matmul__ as tools/precision_study.py does (all dense linears of oracle/staged.py go through `@`)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W, synth
from oracle.staged import Staged

torch.set_num_threads(8)
orig = torch.Tensor.__matmul__
SETS = {
    "6 products (shipping)": [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)],
    "drop (1,1)": [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0)],
    "drop (a2,w0)": [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2)],
    "drop (a0,w2)": [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0)],
    "3 products (bf16x3)": [(0, 0), (0, 1), (1, 0)],
}
ACTIVE = {"set": None}


def split(x, terms):
    parts, r = [], x
    for _ in range(terms):
        p = r.to(torch.bfloat16).to(torch.float64)
        parts.append(p)
        r = r - p
    return parts


def mm(a, b):
    ps = ACTIVE["set"]
    if ps is None or a.dim() != 2 or b.dim() != 2:
        return orig(a, b)
    pa, pb = split(a, 3), split(b, 3)
    out = 0
    for i, j in ps:
        out = out + orig(pa[i], pb[j]).to(torch.float32).to(torch.float64)     # fp32 accumulators
    return out


torch.Tensor.__matmul__ = mm
w = W.make_synthetic_weights(0)
rmsd = 1.5
for n in [int(a) for a in sys.argv[1:]] or [120, 250, 500]:
    z, pos = synth.make_cluster(n)
    pos = pos.astype(np.float32).astype(np.float64)
    st = Staged(w)
    ACTIVE["set"] = None
    e0 = float(st.forward(z, pos))
    line = [f"N={n:5d} edges={len(st.t['src']):7d}"]
    for name, ps in SETS.items():
        ACTIVE["set"] = ps
        e = float(st.forward(z, pos))
        line.append(f"{name}: dE={(e - e0) * rmsd:+.2e} ({(e - e0) * rmsd / n:+.1e}/atom)")
    print(" | ".join(line), flush=True)
