"""Determinism check: the same batch evaluated `reps` times in one engine (and, with a second engine evaluating beside it on its own stream in
another thread, under SIMD sharing) must give bitwise the same energies and forces every time.
    UMX_PRECISION=<mode> python3 tools/gpu_repeat_bitwise.py [atoms=2000] [images=4] [reps=20]"""
import sys
import threading

import numpy as np

sys.path.insert(0, ".")
from pdb2reaction_amd import synth, weights as W  # noqa: E402
from pdb2reaction_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
w = W.make_synthetic_weights(0)
z, imgs, _ = synth.make_images(n, k)
p32 = np.asarray(imgs, dtype=np.float32)
a, b = Engine(0), Engine(0)
for e in (a, b):
    e.load_weights(w); e.set_system(z)
print("mode", a.precision_mode())
e0, f0 = a.energy_forces(p32)
bad = 0
for i in range(reps):
    e, f = a.energy_forces(p32)
    bad += int(not (np.array_equal(e, e0) and np.array_equal(f, f0)))
print(f"alone: {reps} repeats, {bad} differing from the first")
stop = threading.Event()


def other():
    q = p32[::-1].copy()
    while not stop.is_set():
        b.energy_forces(q)


t = threading.Thread(target=other); t.start()
bad2 = 0
worst = 0.0
for i in range(reps):
    e, f = a.energy_forces(p32)
    same = np.array_equal(e, e0) and np.array_equal(f, f0)
    bad2 += int(not same)
    if not same:
        worst = max(worst, float(np.abs(f - f0).max()))
stop.set(); t.join()
print(f"beside a second engine: {reps} repeats, {bad2} differing from the first (max |dF| {worst:.2e})")
sys.exit(1 if (bad or bad2) else 0)
