"""WHERE does the energy error enter?  Linear response along successive CUTS of the network.

All of the energy flows through the node state at every cut x0 -> xmid.0 -> x.0 -> xmid.1 -> ... -> x.3 -> pre1 -> pre2 -> e_node, so for
each cut S the first-order energy error carried by the state so far is c_S = <dE/dS (float64 oracle, oracle/staged.py), S_engine - S_oracle>;
the INCREMENT c_S - c_(S-1) is what the half-block between the two cuts adds (edge-wise block: x.(i-1) -> xmid.i; atom-wise block:
xmid.i -> x.i; readout: x.3 -> pre1 -> pre2 -> e_node).  Noise of a cut: sqrt(sum (g d)^2) is printed beside it.

    python3 tools/gpu_energy_cuts.py [n_atoms] [weights seed]      (environment: UMX_PRECISION, UMX_NODE_F64, UMX_ALT_ROWS)"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from pdb2reaction_amd import synth, weights as W  # noqa: E402
from pdb2reaction_amd.engine import Engine  # noqa: E402
from oracle.staged import Staged, silu_grad  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 700
wseed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.set_num_threads(16)
w = W.make_synthetic_weights(wseed)
z, pos = synth.make_cluster(n)
pos32 = pos.astype(np.float32)
st = Staged(w)
em = st.forward(z, pos32.astype(np.float64))
st.backward()
T = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in st.t.items()}
p = {k: np.asarray(v, dtype=np.float64) for k, v in w.items()}
rmsd = float(w["normalizer.rmsd"][0])
e_ref = float(em) * rmsd + float(p["element_refs"][z].sum())
eng = Engine(0)
eng.load_weights(w)
eng.set_system(z)
eng.debug_keep(True)
e, _ = eng.energy_forces(pos32)
print(f"mode {eng.precision_mode()}  weights seed {wseed}  N = {n}  edges = {len(T['src'])}   dE = {e[0] - e_ref:+.3e} eV ({(e[0] - e_ref) / n:+.2e} eV/atom)")
# gradients of the MODEL energy with respect to every cut
g_pre2 = p["energy_block.4.weight"].reshape(1, -1) * silu_grad(torch.as_tensor(T["pre2"])).numpy()
g_pre1 = (g_pre2 @ p["energy_block.2.weight"]) * silu_grad(torch.as_tensor(T["pre1"])).numpy()
ne = len(T["src"])


def carried(name, g, ref=None):
    a = eng.debug_fetch(name).astype(np.float64)
    d = a - np.asarray(T[name] if ref is None else ref, dtype=np.float64).reshape(-1)
    return np.asarray(g, dtype=np.float64).reshape(-1) * d


cuts = [("x0", T["g_xin.0"])]
for i in range(4):
    cuts.append((f"xmid.{i}", T[f"g_xmid.{i}"]))
    cuts.append((f"x.{i}", T[f"g_xin.{i + 1}"] if i < 3 else T["g_xfinal"]))
cuts += [("pre1", g_pre1), ("pre2", g_pre2), ("e_node", np.ones(n))]
prev = 0.0
for name, g in cuts:
    if name.startswith("xmid."):
        # inside the edge-wise block of layer i: the residual branch x.(i-1) (gradient g_xmid.i) + one of xn.i / hg.i / msg.i is a complete cut
        i = int(name[-1])
        res = carried("x0" if i == 0 else f"x.{i - 1}", T[f"g_xmid.{i}"]).sum() * rmsd
        g_hg = np.concatenate([T[f"g_gate.{i}"], T[f"g_hpre.{i}"].reshape(ne, -1)], axis=1)
        hg_ref = np.concatenate([T[f"gate.{i}"], T[f"hpre.{i}"].reshape(ne, -1)], axis=1)
        sub = [("norm_1 -> xn", res + carried(f"xn.{i}", T[f"g_xn.{i}"]).sum() * rmsd),
               ("gather/rotate/radial/conv-1 -> hg", res + carried(f"hg.{i}", g_hg, hg_ref).sum() * rmsd),
               ("gate/conv-2 -> msg", res + carried(f"msg.{i}", T[f"g_msg.{i}"]).sum() * rmsd)]
        p2 = prev
        for label, c in sub:
            print(f"      layer {i}: {label:36s} carries {c:+.3e} eV   increment {c - p2:+.3e} eV")
            p2 = c
        # the radial MLP alone (geometry -> h1pre -> h2pre -> rad.i): what its error contributes through conv-1, layer by layer of the MLP,
        # with the significance of the per-EDGE contributions (mean / standard error over the edges)
        for nm, gk in ((f"h1pre.{i}", f"g_h1pre.{i}"), (f"h2pre.{i}", f"g_h2pre.{i}"), (f"rad.{i}", f"g_rad.{i}")):
            pe = carried(nm, T[gk]).reshape(ne, -1).sum(1) * rmsd
            print(f"      layer {i}: (radial MLP alone) {nm:10s} carries {pe.sum():+.3e} eV   per edge: mean {pe.mean():+.2e} std {pe.std():.2e} (mean / standard error = {pe.mean() / (pe.std() / np.sqrt(ne)):+.1f})")
    gd = carried(name, g)
    c = gd.sum() * rmsd
    # per-atom contributions: is the cut's error a sum of independent per-atom terms (noise ~ sqrt N) or a common offset?
    per_atom = gd.reshape(n, -1).sum(1) * rmsd
    print(f"cut {name:8s} carries {c:+.3e} eV   increment {c - prev:+.3e} eV   per atom: mean {per_atom.mean():+.2e}  std {per_atom.std():.2e}  "
          f"(mean / standard error = {per_atom.mean() / (per_atom.std() / np.sqrt(n) + 1e-300):+.1f})")
    if name == "x0":
        for nm, gk in (("h1pre.deg", "g_h1pre.deg"), ("h2pre.deg", "g_h2pre.deg"), ("rad.deg", "g_rad.deg")):
            pe = carried(nm, T[gk]).reshape(ne, -1).sum(1) * rmsd
            print(f"      (edge-degree radial MLP alone) {nm:10s} carries {pe.sum():+.3e} eV   per edge: mean {pe.mean():+.2e} std {pe.std():.2e} (mean / standard error = {pe.mean() / (pe.std() / np.sqrt(ne)):+.1f})")
    prev = c
