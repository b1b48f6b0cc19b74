"""Single-image capacity on one GPU: (i) the c5 image (20 000 atoms, fits in one piece) evaluated in 3 forced
partitions against the float64 golden; (ii) images of 26 000 ... atoms that do NOT fit in one piece: partitions chosen by the engine,
time, sum of forces."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from pdb2reaction_amd import synth, weights as W
from pdb2reaction_amd.engine import Engine

w = W.make_synthetic_weights(0)
g = np.load(os.path.join("tests", "golden", "c5_n20000.npz"))
for parts in (0, 3):
    if parts:
        os.environ["UMX_FORCE_PARTS"] = str(parts)
    eng = Engine(0); eng.load_weights(w); eng.set_system(g["z"])
    eng.energy_forces(g["pos"][None])
    t = time.perf_counter(); e, f = eng.energy_forces(g["pos"][None]); dt = time.perf_counter() - t
    print(f"c5 forced parts={parts}: used {eng.last_partitions()}  dE = {e[0] - g['energy'][0]:+.2e} eV  max|dF| = {np.abs(f[0] - g['forces']).max():.2e} eV/A  {dt * 1e3:.0f} ms", flush=True)
    eng.close()
os.environ.pop("UMX_FORCE_PARTS", None)
for n in [int(a) for a in sys.argv[1:]] or [26000, 32000, 38000]:
    z, pos = synth.make_cluster(n)
    eng = Engine(0); eng.load_weights(w); eng.set_system(z)
    try:
        eng.energy_forces(pos[None])
        t = time.perf_counter(); e, f = eng.energy_forces(pos[None]); dt = time.perf_counter() - t
        ne, md = eng.graph_stats()
        print(f"N = {n}: edges {ne}  partitions {eng.last_partitions()}  E = {e[0]:.3f} eV  |sum F| = {np.abs(f[0].astype(np.float64).sum(0)).max():.2e}  max|F| = {np.abs(f).max():.2f}  {dt * 1e3:.0f} ms "
              f"({30.98e6 * ne / dt / 1e12:.0f} alg-TFLOP/s)", flush=True)
    except Exception as exc:
        print(f"N = {n}: {type(exc).__name__}: {str(exc)[:300]}", flush=True)
    eng.close()
