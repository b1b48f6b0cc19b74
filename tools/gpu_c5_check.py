"""c5 size (BASELINE configs[4]: ~20 000 atoms x 8 images): one batched E+F of all 8 images, timing + invariants (no oracle at
this size).  Used under rocprofv3 for profiles/rNN_c5_kernel_stats_*.csv (share of the O(N^2) radius-graph kernels)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W, synth
from pdb2reaction_amd.engine import Engine
eng = Engine(0); eng.load_weights(W.make_synthetic_weights(0))
z, imgs, _ = synth.make_images(20000, 8)
eng.set_system(z)
t = time.time(); e1, f1 = eng.energy_forces(imgs[:1]); t1 = time.time() - t
print("N=20000 K=1: edges/maxdeg", eng.graph_stats(), f"first call {t1:.2f}s", "finite", np.isfinite(f1).all(), "sumF", np.abs(f1.astype(np.float64).sum(1)).max())
t = time.time(); e, f = eng.energy_forces(imgs); t8 = time.time() - t
print(f"K=8: {t8:.3f}s  edges", eng.graph_stats()[0], " image0 identical to single:", e[0] == e1[0], np.array_equal(f[0], f1[0]),
      f" {8 * 20000 / t8:.0f} image*atom/s  sumF {np.abs(f.astype(np.float64).sum(1)).max():.2e}")
