"""Wall time of repeated batched E+F calls through the host-pointer entry: same positions vs positions that change every call,
growing / shrinking batch sizes (dev; the pattern a growing-string run produces).  usage: python tools/gpu_batch_latency.py [atoms]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pdb2reaction_amd import synth, weights as W  # noqa: E402
from pdb2reaction_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
z, imgs, _ = synth.make_images(n, 12)
eng = Engine(0)
eng.load_weights(W.make_synthetic_weights(0))
eng.set_system(z)
rng = np.random.default_rng(0)


def run(label, batches):
    out = []
    for b in batches:
        t = time.perf_counter()
        eng.energy_forces(b)
        out.append((time.perf_counter() - t) * 1e3)
    print(f"{label}: " + " ".join(f"{t:.0f}" for t in out), flush=True)


x10 = imgs[:10].copy()
run("1 image", [imgs[:1]] * 3)
run("10 images, same positions", [x10] * 8)
run("10 images, jitter 1e-3 A per call", [x10 + rng.normal(size=x10.shape).astype(np.float32) * 1e-3 for _ in range(8)])
run("10 images, same again", [x10] * 4)
run("12 images", [imgs] * 4)
run("10 images after 12", [x10] * 6)
shrink = [x10 * (1.0 + 1e-4 * i) for i in range(8)]            # expanding cluster: fewer edges every call
run("10 images, slowly expanding (edge count falls)", shrink)
run("10 images, contracting again (edge count rises)", shrink[::-1])
eng.close()
