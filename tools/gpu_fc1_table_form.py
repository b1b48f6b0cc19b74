"""fc1 of the radial MLPs: is the element-table add ("float32 MFMA sum + table constant of the element pair", rounded once from double) visible as a
coherent offset?  The engine's h1pre against the float64 staged oracle, grouped by (source element, target element): number of (pair, column)
cells whose mean error is more than 4 standard errors from zero, and the first-order energy the cells' means carry.

    python3 tools/gpu_fc1_table_form.py [n_atoms] [weights seed]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from pdb2reaction_amd import synth, weights as W  # noqa: E402
from pdb2reaction_amd.engine import Engine  # noqa: E402
from oracle.staged import Staged  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 700
wseed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.set_num_threads(16)
w = W.make_synthetic_weights(wseed)
z, pos = synth.make_cluster(n)
pos32 = pos.astype(np.float32)
st = Staged(w)
st.forward(z, pos32.astype(np.float64))
st.backward()
T = st.t
rmsd = float(w["normalizer.rmsd"][0])
src, dst = np.asarray(T["src"]), np.asarray(T["dst"])
ne = len(src)
pair = z[src].astype(np.int64) * 200 + z[dst]
eng = Engine(0)
eng.load_weights(w)
eng.set_system(z)
eng.debug_keep(True)
eng.energy_forces(pos32)
print(f"mode {eng.precision_mode()}  weights seed {wseed}  N = {n}  edges = {ne}  element pairs = {len(np.unique(pair))}")
for tag in ["deg", "0", "1", "2", "3"]:
    ref = T[f"h1pre.{tag}"].numpy().reshape(ne, -1)
    d = eng.debug_fetch(f"h1pre.{tag}").astype(np.float64).reshape(ne, -1) - ref
    g = T[f"g_h1pre.{tag}"].numpy().reshape(ne, -1)
    cells = sig = 0
    carried_mean = 0.0
    for pv in np.unique(pair):
        m = pair == pv
        if m.sum() < 200:
            continue
        dm, gm = d[m], g[m]
        t = dm.mean(0) / (dm.std(0) / np.sqrt(m.sum()) + 1e-300)
        cells += dm.shape[1]
        sig += int((np.abs(t) > 4).sum())
        carried_mean += float((gm.sum(0) * dm.mean(0)).sum()) * rmsd          # what the per-(pair, column) MEAN errors carry into the energy
    tot = float((g * d).sum()) * rmsd
    print(f"h1pre.{tag:3s}: {sig} of {cells} (pair, column) cells with |mean error| > 4 standard errors (pairs with >= 200 edges); rms error {np.sqrt((d * d).mean()):.2e}; "
          f"energy carried: total {tot:+.2e} eV, by the cell means {carried_mean:+.2e} eV")
