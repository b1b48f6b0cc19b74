"""One BASELINE config through the host-pointer entry: wall time per batched E+F of all images (optionally under rocprofv3).

    python3 tools/gpu_eval_config.py c2 [reps]
"""
import sys
import time

sys.path.insert(0, ".")
from pdb2reaction_amd import synth, weights as W  # noqa: E402
from pdb2reaction_amd.engine import Engine  # noqa: E402

CONFIGS = {"c1": (50, 8), "c2": (500, 12), "c3": (2000, 16), "c3-shard": (2000, 2), "c4-string": (2000, 24), "c5": (20000, 8)}
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
n, k = CONFIGS[name]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else (20 if n <= 500 else 2)
eng = Engine(0)
eng.load_weights(W.make_synthetic_weights(0))
z, imgs, _ = synth.make_images(n, k)
eng.set_system(z)
eng.reserve_images(k)                      # steady-state timing of fixed batches: workspace for the whole batch, allocated once
eng.energy_forces(imgs)
eng.energy_forces(imgs)
t = time.perf_counter()
for _ in range(reps):
    e, f = eng.energy_forces(imgs)
dt = (time.perf_counter() - t) / reps
ne, md = eng.graph_stats()
print(f"{name}: N={n} K={k} edges={ne} -> {dt * 1e3:.2f} ms per E+F of all images ({reps} reps; {k * n / dt:.3e} image-atom/s, "
      f"{30.98e6 * ne / dt / 1e12:.1f} alg-TFLOP/s)", flush=True)
