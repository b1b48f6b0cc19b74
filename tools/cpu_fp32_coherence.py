"""Is the COHERENT energy error of a float32-accumulated radial fc2 (tools/gpu_energy_cuts.py: h1pre -> h2pre) a property of the matrix
cores, or of float32 accumulation as such?  CPU emulation on the float64 oracle's own tensors: a1 = SiLU(LN(h1pre)) rounded to float32,
h2 = a1 . W2^T + b2 accumulated in float32 (i) by a sequential fma-free chain over k and (ii) by numpy's float32 matmul (BLAS: blocked,
FMA), against the float64 product; contribution to the energy = <dE/dh2pre (float64), h2_f32 - h2_f64> per edge.

    python tools/cpu_fp32_coherence.py [n_atoms] [weights seed]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from pdb2reaction_amd import synth, weights as W  # noqa: E402
from oracle.staged import Staged, ln_silu_fwd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 700
wseed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.set_num_threads(8)
w = W.make_synthetic_weights(wseed)
z, pos = synth.make_cluster(n)
st = Staged(w)
st.forward(z, pos.astype(np.float32).astype(np.float64))
st.backward()
T, p = st.t, st.p
rmsd = float(w["normalizer.rmsd"][0])
ne = len(T["src"])
print(f"weights seed {wseed}  N = {n}  edges = {ne}")
for tag, prefix in [("deg", "edge_degree_embedding.rad_func")] + [(str(i), f"blocks.{i}.edge_wise.so2_conv_1.rad_func") for i in range(4)]:
    a1 = ln_silu_fwd(T[f"h1pre.{tag}"], p[f"{prefix}.ln1.weight"], p[f"{prefix}.ln1.bias"]).numpy()
    w2, b2 = p[f"{prefix}.fc2.weight"].numpy(), p[f"{prefix}.fc2.bias"].numpy()
    g = T[f"g_h2pre.{tag}"].numpy()
    exact = a1 @ w2.T + b2
    a32, w32, b32 = a1.astype(np.float32), w2.astype(np.float32), b2.astype(np.float32)
    ref32in = a32.astype(np.float64) @ w2.T + b2                      # exact arithmetic on the float32-rounded inputs
    seq = np.zeros((ne, w32.shape[0]), dtype=np.float32)
    for k in range(a32.shape[1]):
        seq = (seq + a32[:, k:k + 1] * w32[None, :, k]).astype(np.float32)     # product rounded, sum rounded: the plainest float32 chain
    seq = (seq + b32[None, :]).astype(np.float32)
    blas = (a32 @ w32.T + b32).astype(np.float32)
    for label, h in (("inputs rounded to float32, exact product", ref32in), ("sequential float32 chain", seq.astype(np.float64)), ("numpy float32 matmul", blas.astype(np.float64))):
        pe = (g * (h - exact)).sum(1) * rmsd
        print(f"h2pre.{tag:3s} {label:42s} carries {pe.sum():+.3e} eV   per edge mean {pe.mean():+.2e} std {pe.std():.2e}  (mean / standard error {pe.mean() / (pe.std() / np.sqrt(ne)):+.1f})")
