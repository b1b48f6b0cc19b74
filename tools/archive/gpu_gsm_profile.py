"""Where does a growing-string run spend its wall time?  (dev)  usage: python tools/gpu_gsm_profile.py [atoms] [max_nodes] [cycles]
Prints per-phase wall time (calculator set-up, first evaluation, the cycles) and a cProfile of the cycles after the first."""
import cProfile
import importlib
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pdb2reaction_amd import synth  # noqa: E402
from pdb2reaction_amd.gsm import GrowingStringDriver  # noqa: E402

U = importlib.import_module("pdb2reaction_amd.uma_pysis")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
nodes = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 40
z, imgs, frozen = synth.make_images(n, 2)
elem = [synth.SYMBOLS[int(a)] for a in z]
t0 = time.perf_counter()
calc = U.uma_pysis(model="synthetic", freeze_atoms=list(frozen))
r, p = (imgs[0] * U.ANG2BOHR).reshape(-1), (imgs[1] * U.ANG2BOHR).reshape(-1)
calc.get_forces(elem, r)
t1 = time.perf_counter()
print(f"calculator + first single evaluation: {t1 - t0:.2f} s")
stamps = []
edges = []
orig = calc.get_forces_batch


def timed(el, c):
    a = time.perf_counter()
    out = orig(el, c)
    stamps.append((len(c), time.perf_counter() - a))
    edges.append(calc._core.engine.graph_stats()[0] if hasattr(calc._core, "engine") else -1)
    return out


calc.get_forces_batch = timed
drv = GrowingStringDriver(elem, r, p, calc, gs_kw={"max_nodes": nodes, "climb": True}, stopt_kw={"max_cycles": cycles})
pr = cProfile.Profile()
t2 = time.perf_counter()
pr.enable()
res = drv.run()
pr.disable()
t3 = time.perf_counter()
ev = sum(s[1] for s in stamps)
print(f"{res.cycles} cycles, {res.force_evaluations} image evaluations in {t3 - t2:.2f} s = {(t3 - t2) / res.cycles * 1e3:.1f} ms/cycle; "
      f"inside get_forces_batch {ev:.2f} s ({ev / (t3 - t2) * 100:.0f} %)")
for k in sorted({s[0] for s in stamps}):
    ts = [s[1] for s in stamps if s[0] == k]
    print(f"  batches of {k:2d} images: {len(ts):3d} calls, first {ts[0] * 1e3:7.1f} ms, median {np.median(ts) * 1e3:7.1f} ms")
print("per call (images: ms): " + " ".join(f"{k}:{t * 1e3:.0f}" for k, t in stamps))
print("edges per call: " + " ".join(str(e) for e in edges))
pstats.Stats(pr).sort_stats("cumulative").print_stats(8)
