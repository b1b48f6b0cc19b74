"""Where does the pre-alignment + staged scan spend its wall time?  (dev)  usage: python tools/gpu_prestep_profile.py [atoms] [mobile images]"""
import cProfile
import importlib
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pdb2reaction_amd import prestep as PS, synth  # noqa: E402

U = importlib.import_module("pdb2reaction_amd.uma_pysis")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
m = int(sys.argv[2]) if len(sys.argv) > 2 else 4
z, imgs, frozen = synth.make_images(n, m + 1)
elem = [synth.SYMBOLS[int(a)] for a in z]
anchors = list(frozen)
calc = U.uma_pysis(model="synthetic")
ref = imgs[0] * U.ANG2BOHR
rng = np.random.default_rng(0)
mobs = []
for k in range(1, m + 1):
    q = imgs[k] * U.ANG2BOHR
    q[anchors] += 0.4 * rng.normal(size=(len(anchors), 3))            # anchors off by ~0.35 A: a handful of 0.1 A scan steps
    mobs.append(q + np.array([2.0, -1.0, 0.5]))
calc.get_forces(elem, ref.reshape(-1))
stamps = []
orig = calc.get_forces_batch


def timed(el, c):
    a = time.perf_counter()
    out = orig(el, c)
    stamps.append((len(c), time.perf_counter() - a))
    return out


calc.get_forces_batch = timed
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
out, res = PS.align_and_refine_sequence(calc, elem, [ref] + mobs, [anchors] * (m + 1), step_A=0.1, per_step_cycles=10, final_cycles=30, thresh="gau_loose")
pr.disable()
dt = time.perf_counter() - t0
ev = sum(s[1] for s in stamps)
print(f"{m} mobile images of {n} atoms: {dt:.2f} s, {len(stamps)} batched calls ({sum(s[0] for s in stamps)} image evaluations), "
      f"inside get_forces_batch {ev:.2f} s ({100 * ev / dt:.0f} %); scan steps {[r['scan']['n_steps'] for r in res]}")
for k in sorted({s[0] for s in stamps}):
    ts = [s[1] for s in stamps if s[0] == k]
    print(f"  batches of {k:2d}: {len(ts):4d} calls, first {ts[0] * 1e3:7.1f} ms, median {np.median(ts) * 1e3:7.1f} ms")
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
