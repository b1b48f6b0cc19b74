"""c5-size sanity: 20 000 atoms, 1 and 2 images (chunking), invariants only (no oracle at this size)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W, synth
from pdb2reaction_amd.engine import Engine
eng = Engine(0); eng.load_weights(W.make_synthetic_weights(0))
z, imgs, _ = synth.make_images(20000, 2)
eng.set_system(z)
t = time.time(); e, f = eng.energy_forces(imgs[:1]); t1 = time.time() - t
print("N=20000 K=1: edges/maxdeg", eng.graph_stats(), "E", e, f"first call {t1:.2f}s", "finite", np.isfinite(f).all(), "sumF", np.abs(f.astype(np.float64).sum(1)).max(), "max|F|", np.abs(f).max())
t = time.time(); e2, f2 = eng.energy_forces(imgs); t2 = time.time() - t
print(f"K=2: {t2:.2f}s  image0 identical to single:", e2[0] == e[0], np.array_equal(f2[0], f[0]))
t = time.time(); e3, _ = eng.energy_forces(imgs[:1], forces=False); print(f"energy only {time.time()-t:.2f}s", e3[0] == e[0])
