"""Which cut of the 16-bit matrix core makes the COHERENT part of a bf16x3 GEMM's error, and which accumulation scheme avoids it?  (round 6)

The float64 staged oracle provides, for one cluster, the operands of the forward GEMM sites (radial fc3, SO(2) conv-1 / conv-2 m = 0) and
dE/d(output); the bit-exact matrix-core model (tools/mfma_emul.c) evaluates the engine's six-product GEMM under an accumulation scheme.
Printed per site and scheme: the first-order energy error the GEMM's rounding carries (sum over edges of <dE/dy, y_model - y_exact>), its
significance (mean / standard error over edges: a random error stays within +-3, a coherent one grows with sqrt(edges)), and the number of
output columns whose mean error is off by more than 4 standard errors.

    python tools/cpu_mfma_gemm_bias.py [n_atoms=300] [weights seed=1] [columns per site=128]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import mfma_model as MM  # noqa: E402
from oracle.staged import Staged, ln_silu_fwd  # noqa: E402
from pdb2reaction_amd import synth, weights as W  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
wseed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ncol = int(sys.argv[3]) if len(sys.argv) > 3 else 128
only = sys.argv[4].split(",") if len(sys.argv) > 4 and sys.argv[4] != "all" else None
quick = len(sys.argv) > 5 and sys.argv[5] == "quick"
plain_abl = len(sys.argv) > 5 and sys.argv[5] == "plain"
torch.set_num_threads(8)
w = W.make_synthetic_weights(wseed)
z, pos = synth.make_cluster(n)
st = Staged(w)
t0 = time.time()
st.forward(z, pos.astype(np.float32).astype(np.float64))
st.backward()
T, p = st.t, st.p
rmsd = float(w["normalizer.rmsd"][0])
ne = len(T["src"])
print(f"# weights seed {wseed}  N = {n}  edges = {ne}  (oracle {time.time() - t0:.0f} s)")
C, H = W.SPHERE_CHANNELS, W.HIDDEN_CHANNELS


def sites():
    for tag, prefix in [("deg", "edge_degree_embedding.rad_func")] + [(str(i), f"blocks.{i}.edge_wise.so2_conv_1.rad_func") for i in range(W.NUM_LAYERS)]:
        a2 = ln_silu_fwd(T[f"h2pre.{tag}"], p[f"{prefix}.ln2.weight"], p[f"{prefix}.ln2.bias"]).numpy()
        yield f"fc3.{tag}", a2, p[f"{prefix}.fc3.weight"].numpy(), p[f"{prefix}.fc3.bias"].numpy(), T[f"g_rad.{tag}"].numpy().reshape(ne, -1)
    for i in range(W.NUM_LAYERS):
        b = f"blocks.{i}.edge_wise"
        x0 = (T[f"xrot.{i}"][:, 0:3, :].reshape(ne, -1) * T[f"rad.{i}"][:, : 3 * 2 * C]).numpy()
        g = torch.cat([T[f"g_gate.{i}"], T[f"g_hpre.{i}"][:, 0:3].reshape(ne, -1)], dim=1).numpy()
        yield f"c1m0.{i}", x0, p[f"{b}.so2_conv_1.fc_m0.weight"].numpy(), p[f"{b}.so2_conv_1.fc_m0.bias"].numpy(), g
        yield (f"c2m0.{i}", T[f"hid.{i}"][:, 0:3].reshape(ne, -1).numpy(), p[f"{b}.so2_conv_2.fc_m0.weight"].numpy(),
               p[f"{b}.so2_conv_2.fc_m0.bias"].numpy(), T[f"g_msg.{i}"][:, 0:3].reshape(ne, -1).numpy())


SCHEMES = dict(MM.SCHEMES)
lib = MM.load_lib()
import ctypes  # noqa: E402
abl = ctypes.c_int.in_dll(lib, "mfma_ablate")
dem_a, dem_w = ctypes.c_int.in_dll(lib, "gemm_dem_a"), ctypes.c_int.in_dll(lib, "gemm_dem_w")
rng = np.random.default_rng(0)
for name, A, Wt, bias, g in sites():
    if only and not any(name.startswith(o) for o in only):
        continue
    A32 = A.astype(np.float32); W32 = Wt.astype(np.float32); b32 = bias.astype(np.float32)
    N = W32.shape[0]
    cols = np.sort(rng.choice(N, size=min(ncol, N), replace=False)).astype(np.int32)
    exact = A32.astype(np.float64) @ W32[cols].astype(np.float64).T + b32[cols].astype(np.float64)
    gs = g[:, cols] * rmsd * (N / len(cols))
    print(f"{name:8s} M = {ne}  N = {N} ({len(cols)} columns)  K = {A32.shape[1]}   rms y = {np.sqrt((exact ** 2).mean()):.3f}")
    base = None
    runs = [("plain", 0, 0, 0), ("ls1", 0, 0, 0), ("ls2", 0, 0, 0)] + [("ls2", 1 << k, 0, 0) for k in range(4)] + [("ls2", 15, 0, 0)]
    runs += [("ls2", 0, 12, 12), ("ls2", 0, 12, 0), ("ls2", 0, 0, 12), ("ls1", 0, 12, 12), ("plain", 0, 12, 12), ("ls2", 0, 11, 11), ("ls2", 0, 13, 11)]     # aligned planes: leading-plane quantum 2^(e_max - da) / 2^(e_max - dw)
    if quick:
        runs = [r for r in runs if r[1] in (0, 1)]
    if plain_abl:             # which cut makes the one-accumulator scheme's column offsets?
        runs = [("ls2", 0, 12, 12)] + [("plain", ab, 12, 12) for ab in (0, 1, 2, 4, 8, 6, 14)]
    for sch, ab, da, dw in runs:
        abl.value = ab
        dem_a.value, dem_w.value = da, dw
        t0 = time.time()
        y = MM.gemm_bf16x3(A32, W32, b32, SCHEMES[sch], cols=cols)
        d = y.astype(np.float64) - exact
        pe = (gs * d).sum(1)
        cm, cs = d.mean(0), d.std(0) / np.sqrt(ne)
        line = (f"    {sch:6s} ablate {ab:2d} align {da}/{dw}: carries {pe.sum():+.3e} eV  ({pe.sum() / n:+.2e} eV/atom)  mean / s.e. {pe.mean() / (pe.std() / np.sqrt(ne)):+5.1f}"
                f"   columns off by > 4 s.e.: {int((np.abs(cm / cs) > 4).sum())}/{len(cols)}   rms d {np.sqrt((d * d).mean()):.2e}")
        if sch == "ls2" and ab == 0:
            base = pe
        elif base is not None:
            # PAIRED difference against the engine's scheme on the same data: the contribution of exactly the cut that was replaced
            # (or of the scheme change), free of the noise of everything else
            dd = base - pe
            line += f"   | ls2 - this: {dd.sum():+.3e} eV ({dd.sum() / n:+.2e} eV/atom), mean / s.e. {dd.mean() / (dd.std() / np.sqrt(ne) + 1e-300):+5.1f}"
        print(line + f"  ({time.time() - t0:.0f} s)", flush=True)
    abl.value = 0
    dem_a.value = dem_w.value = 0
