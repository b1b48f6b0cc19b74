"""The radial fc3 GEMM ALONE: what form does its error have?  (round 5, after the bias-first accumulators)

The engine's own fc2 output (h2pre, float32, captured) is pushed through LayerNorm + SiLU + fc3 in float64 on the CPU; the difference d
between the engine's `rad` and that is the error of the LN/SiLU tail (float32, the same kernel in every mode) plus the fc3 GEMM of the
precision mode.  Printed per radial MLP: the first-order energy it carries, <dE/d rad (float64 oracle), d>, for all rows and for the even
and odd rows separately (odd rows are stored negated: a one-sided matrix-core rounding shows up with OPPOSITE signs on the two halves, a
sign-symmetric one with the same sign), and a least-squares fit of d on [1, r, |r|, b3(column)].

    UMX_PRECISION=bf16x3 python3 tools/gpu_fc3_error_form.py [n_atoms] [weights seed]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from pdb2reaction_amd import synth, weights as W  # noqa: E402
from pdb2reaction_amd.engine import Engine  # noqa: E402
from oracle.staged import Staged, ln_silu_fwd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 700
wseed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.set_num_threads(16)
w = W.make_synthetic_weights(wseed)
z, pos = synth.make_cluster(n)
pos32 = pos.astype(np.float32)
st = Staged(w)
st.forward(z, pos32.astype(np.float64))
st.backward()
T, p = st.t, st.p
rmsd = float(w["normalizer.rmsd"][0])
ne = len(T["src"])
eng = Engine(0)
eng.load_weights(w)
eng.set_system(z)
eng.debug_keep(True)
eng.energy_forces(pos32)
print(f"mode {eng.precision_mode()}  weights seed {wseed}  N = {n}  edges = {ne}")
even = (np.arange(ne) % 2) == 0
for tag, prefix in [("deg", "edge_degree_embedding.rad_func")] + [(str(i), f"blocks.{i}.edge_wise.so2_conv_1.rad_func") for i in range(4)]:
    h2 = torch.as_tensor(eng.debug_fetch(f"h2pre.{tag}").astype(np.float64).reshape(ne, -1))
    a2 = ln_silu_fwd(h2, p[f"{prefix}.ln2.weight"], p[f"{prefix}.ln2.bias"]).numpy()
    w3, b3 = p[f"{prefix}.fc3.weight"].numpy(), p[f"{prefix}.fc3.bias"].numpy()
    r = a2 @ w3.T + b3
    d = eng.debug_fetch(f"rad.{tag}").astype(np.float64).reshape(ne, -1) - r
    g = T[f"g_rad.{tag}"].numpy().reshape(ne, -1)
    pe = (g * d).sum(1) * rmsd
    for label, m in (("all rows", np.ones(ne, bool)), ("even rows", even), ("odd rows", ~even)):
        x = pe[m]
        print(f"rad.{tag:3s} {label:9s} carries {x.sum():+.3e} eV  per edge mean {x.mean():+.2e} std {x.std():.2e}  (mean / standard error {x.mean() / (x.std() / np.sqrt(len(x))):+.1f})")
    for label, m in (("even rows", even), ("odd rows", ~even)):
        rr, dd = r[m].reshape(-1), d[m].reshape(-1)
        bb = np.broadcast_to(b3[None, :], r[m].shape).reshape(-1)
        xs = np.stack([np.ones_like(rr), rr, np.abs(rr), bb], axis=1)
        coef, *_ = np.linalg.lstsq(xs, dd, rcond=None)
        print(f"        {label:9s} d ~ {coef[0]:+.2e} {coef[1]:+.2e} r {coef[2]:+.2e} |r| {coef[3]:+.2e} b3     rms d {np.sqrt((dd * dd).mean()):.2e}  rms r {np.sqrt((rr * rr).mean()):.3f}")
    # per-column mean error against its standard error: how many columns are significantly off, and does the column offset follow the bias?
    cm, cs = d.mean(0), d.std(0) / np.sqrt(ne)
    sig = np.abs(cm / cs) > 4
    print(f"        columns with |mean d| > 4 standard errors: {int(sig.sum())} of {d.shape[1]};  corr(mean d, b3) = {np.corrcoef(cm, b3)[0, 1]:+.2f}  corr(mean d, |b3|) = {np.corrcoef(cm, np.abs(b3))[0, 1]:+.2f}"
          f"  corr(mean d, mean r) = {np.corrcoef(cm, r.mean(0))[0, 1]:+.2f}  corr(mean d, mean |r|) = {np.corrcoef(cm, np.abs(r).mean(0))[0, 1]:+.2f}")
