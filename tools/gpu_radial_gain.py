"""Is the error of the fused radial head's small linears a GAIN (every product shrunk / inflated by the same relative amount)?
Regress (engine - float64 oracle) of h1pre / h2pre on the part of the oracle value that comes out of the matrix pipe:
h1pre - (element tables + bias) = W1g . gauss,   h2pre - b2 = W2 . a1.      python3 tools/gpu_radial_gain.py [n_atoms] [weights seed]"""
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
T = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in st.t.items()}
p = {k: np.asarray(v, dtype=np.float64) for k, v in w.items()}
eng = Engine(0)
eng.load_weights(w)
eng.set_system(z)
eng.debug_keep(True)
eng.energy_forces(pos32, forces=False)
ne = len(T["src"])
print(f"mode {eng.precision_mode()}  weights seed {wseed}  N = {n}  edges = {ne}")
for tag, prefix in [("deg", "edge_degree_embedding.rad_func")] + [(str(i), f"blocks.{i}.edge_wise.so2_conv_1.rad_func") for i in range(4)]:
    w1 = p[f"{prefix}.fc1.weight"]
    lin1 = T["gauss"] @ w1[:, :64].T                                  # the fp32-MFMA part of fc1
    lin2 = T[f"h2pre.{tag}"] - p[f"{prefix}.fc2.bias"][None, :]       # ... of fc2
    for name, lin in ((f"h1pre.{tag}", lin1), (f"h2pre.{tag}", lin2)):
        a = eng.debug_fetch(name).astype(np.float64).reshape(ne, -1)
        d = a - T[name]
        gain = (d * lin).sum() / (lin * lin).sum()
        # significance: per-edge estimates of the gain
        ge = (d * lin).sum(1) / (lin * lin).sum(1)
        print(f"{name:10s} gain on the matrix-pipe part {gain:+.3e}  (per-edge mean {ge.mean():+.3e} +- {ge.std() / np.sqrt(ne):.1e});  mean diff {d.mean():+.2e}  rms diff {np.sqrt((d * d).mean()):.2e}  rms of the part {np.sqrt((lin * lin).mean()):.2e}")
