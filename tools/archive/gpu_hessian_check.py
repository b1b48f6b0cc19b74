"""FD-Hessian throughput check on the engine: N atoms, `n_active` movable, 2*3*n_active displaced geometries batched through
get_forces_batch (reference uma_pysis.py:595-686).  usage: python tools/gpu_hessian_check.py [atoms] [active]"""
import importlib
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pdb2reaction_amd import synth  # noqa: E402

U = importlib.import_module("pdb2reaction_amd.uma_pysis")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
act = int(sys.argv[2]) if len(sys.argv) > 2 else 60
z, pos = synth.make_cluster(n)
elem = [synth.SYMBOLS[int(a)] for a in z]
order = np.argsort(np.linalg.norm(pos, axis=1))
freeze = sorted(int(i) for i in order[act:])
calc = U.uma_pysis(model="synthetic", freeze_atoms=freeze, out_hess_torch=False, return_partial_hessian=True)
x = (pos * U.ANG2BOHR).reshape(-1)
calc.get_forces(elem, x)                      # load weights
t0 = time.perf_counter()
calc.get_hessian(elem, x)                     # first call also sizes the workspace for the 64-geometry batches
dt_first = time.perf_counter() - t0
t0 = time.perf_counter()
res = calc.get_hessian(elem, x)
dt = time.perf_counter() - t0
print(f"first call {dt_first:.2f} s (incl. workspace allocation), steady state {dt:.2f} s")
h = np.asarray(res["hessian"])
print(f"atoms {n} active {act}: Hessian {h.shape} in {dt:.2f} s = {2 * 3 * act / dt:.1f} displaced geometries/s "
      f"({2 * 3 * act * n / dt / 1e3:.1f} k image-atoms/s)")
print("symmetric:", float(np.abs(h - h.T).max()), " max |H|:", float(np.abs(h).max()))
assert h.shape == (3 * act, 3 * act) and np.isfinite(h).all() and np.abs(h - h.T).max() < 1e-12
