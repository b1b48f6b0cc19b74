"""Wall time per batched E+F call for the BASELINE configs (host-pointer entry, includes PCIe copies)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W, synth
from pdb2reaction_amd.engine import Engine
eng = Engine(0); eng.load_weights(W.make_synthetic_weights(0))
for name, n, k in (("c1", 50, 8), ("c2", 500, 12), ("c3", 2000, 16), ("c4-string", 2000, 24), ("c5", 20000, 8)):
    z, imgs, _ = synth.make_images(n, k); eng.set_system(z); eng.reserve_images(k)
    eng.energy_forces(imgs)
    reps = 20 if n <= 500 else 2
    t = time.time()
    for _ in range(reps): e, f = eng.energy_forces(imgs)
    dt = (time.time() - t) / reps
    ne, md = eng.graph_stats()
    print(f"{name}: N={n} K={k} edges={ne} -> {dt*1e3:.1f} ms per E+F of all images ({1/dt:.2f} evaluations/s, {k*n/dt:.3e} image-atom/s, {30.98e6*ne/dt/1e12:.1f} alg-TFLOP/s)", flush=True)
