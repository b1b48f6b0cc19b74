"""Where does a SYSTEMATIC (same-sign, ~N) energy error enter?  Every forward stage of the engine against the staged float64 oracle:
gain = <a, r> / <r, r> - 1 (a multiplicative bias; float32 storage noise averages out over the ~1e6-1e7 elements of a stage, so
1e-9 is resolvable), mean signed difference, rms difference.

    python3 tools/gpu_stage_bias.py [n_atoms] [weights seed]   (environment: UMX_PRECISION, UMX_NODE_F64, ...)"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from pdb2reaction_amd import synth, weights as W  # noqa: E402
from pdb2reaction_amd.engine import Engine  # noqa: E402
from oracle.staged import Staged  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.set_num_threads(16)
wseed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
w = W.make_synthetic_weights(wseed)
z, pos = synth.make_cluster(n)
pos32 = pos.astype(np.float32)
st = Staged(w)
em = st.forward(z, pos32.astype(np.float64))
T = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in st.t.items()}
rmsd = float(w["normalizer.rmsd"][0])
e_ref = float(em) * rmsd + float(np.asarray(w["element_refs"], np.float64)[z].sum())
eng = Engine(0)
eng.load_weights(w)
eng.set_system(z)
eng.debug_keep(True)
e, _ = eng.energy_forces(pos32, forces=False)
ne = len(T["src"])
print(f"mode {eng.precision_mode()}  weights seed {wseed}  N = {n}  edges = {ne}   dE = {e[0] - e_ref:+.3e} eV ({(e[0] - e_ref) / n:+.2e} eV/atom)")


def cmp(name, ref):
    try:
        a = eng.debug_fetch(name).astype(np.float64)
    except Exception as ex:
        print(f"{name:10s} missing ({str(ex)[:40]})")
        return
    r = np.asarray(ref, dtype=np.float64).reshape(-1)
    if a.size != r.size:
        print(f"{name:10s} size {a.size} vs {r.size}")
        return
    d = a - r
    print(f"{name:10s} gain {np.dot(a, r) / np.dot(r, r) - 1.0:+.2e}   mean diff {d.mean():+.2e}   rms diff {np.sqrt((d * d).mean()):.2e}   rms ref {np.sqrt((r * r).mean()):.2e}")


fr = eng.debug_fetch("frame").reshape(ne, 36).astype(np.float64)
d = fr[:, 34] - T["env"]
print(f"{'env':10s} gain {np.dot(fr[:, 34], T['env']) / np.dot(T['env'], T['env']) - 1:+.2e}   mean diff {d.mean():+.2e}")
cmp("rad.deg", T["rad.deg"]); cmp("x0", T["x0"])
for i in range(4):
    cmp(f"xn.{i}", T[f"xn.{i}"]); cmp(f"rad.{i}", T[f"rad.{i}"])
    hg = np.concatenate([T[f"gate.{i}"], T[f"hpre.{i}"].reshape(ne, -1)], axis=1)
    cmp(f"hg.{i}", hg); cmp(f"msg.{i}", T[f"msg.{i}"]); cmp(f"xmid.{i}", T[f"xmid.{i}"])
    cmp(f"xn2.{i}", T[f"xn2.{i}"]); cmp(f"gspre.{i}", T[f"gspre.{i}"]); cmp(f"ffh.{i}", T[f"ffh.{i}"]); cmp(f"x.{i}", T[f"x.{i}"])
cmp("pre1", T["pre1"]); cmp("pre2", T["pre2"]); cmp("e_node", T["e_node"])
en = eng.debug_fetch("e_node").astype(np.float64)
print(f"sum(e_node) diff {(en - T['e_node'].reshape(-1)).sum() * rmsd:+.3e} eV")
