"""Round-5 A/B measurements that need ONE box (alternating runs inside one process / one gpurun call):

    python3 tools/gpu_r5_ab.py lanes      # UMX_STREAMS=1 vs 2: c3 (16 images), the 2-image shard of the 8-GPU run, c2, c4-string
    python3 tools/gpu_r5_ab.py variant    # grid feed-forward vs spectral: ms per E+F at c3 / c2
    python3 tools/gpu_r5_ab.py neigh      # c3 with max_neigh in {300, 70, 50, 30}: edges, ms per E+F (the cap is the checkpoint's to choose)

Wall time of the device-resident batched E+F (host-pointer entry: + two small PCIe copies), after two warm-up calls.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pdb2reaction_amd import synth, weights as W  # noqa: E402
from pdb2reaction_amd.engine import Engine  # noqa: E402

WEIGHTS = W.make_synthetic_weights(0)


def timed(n, k, reps, env=None, max_neigh=None, weights=None):
    for key, val in (env or {}).items():
        os.environ[key] = val
    eng = Engine(0)
    eng.load_weights(weights if weights is not None else WEIGHTS)
    z, imgs, _ = synth.make_images(n, k)
    eng.set_system(z, max_neigh=max_neigh)
    eng.reserve_images(k)
    e0, f0 = eng.energy_forces(imgs)
    eng.energy_forces(imgs)
    t = time.perf_counter()
    for _ in range(reps):
        e, f = eng.energy_forces(imgs)
    dt = (time.perf_counter() - t) / reps
    ne, md = eng.graph_stats()
    eng.close()
    for key in (env or {}):
        os.environ.pop(key, None)
    return dt * 1e3, ne, md, e, f


what = sys.argv[1] if len(sys.argv) > 1 else "lanes"
if what == "lanes":
    for name, n, k, reps in (("c3", 2000, 16, 4), ("c3-shard", 2000, 2, 12), ("c2", 500, 12, 20), ("c4-string", 2000, 24, 3), ("c1", 50, 8, 40)):
        ref = None
        for rnd in range(2):
            for lanes in ("1", "2"):
                ms, ne, md, e, f = timed(n, k, reps, {"UMX_STREAMS": lanes})
                same = ""
                if ref is None:
                    ref = (e.copy(), f.copy())
                else:
                    same = f"  bitwise == first run: E {bool((e == ref[0]).all())} F {bool((f == ref[1]).all())}"
                print(f"{name}: UMX_STREAMS={lanes} round {rnd}: {ms:.2f} ms per E+F of {k} images ({ne} edges){same}", flush=True)
elif what == "neigh":
    for mn in (300, 70, 50, 30, 300):
        ms, ne, md, e, f = timed(2000, 16, 3, None, mn)
        print(f"c3: max_neigh={mn}: {ms:.2f} ms per E+F of 16 images, {ne} directed edges, max degree {md}, E[0] = {e[0]:.6f} eV", flush=True)
elif what == "variant":          # the grid feed-forward beside the spectral one (c3, c2): ms per E+F, float64-accumulated vs fp32-MFMA grid GEMMs
    wg = W.make_synthetic_weights(0, ff_type="grid")
    for name, n, k, reps in (("c3", 2000, 16, 3), ("c2", 500, 12, 10)):
        for rnd in range(2):
            for label, wts, env in (("spectral", None, None), ("grid (float64-accumulated grid GEMMs)", wg, None), ("grid (UMX_GRID_F64=0: fp32 MFMA)", wg, {"UMX_GRID_F64": "0"})):
                ms, ne, md, e, f = timed(n, k, reps, env, None, wts)
                print(f"{name}: {label}: round {rnd}: {ms:.2f} ms per E+F of {k} images ({ne} edges), E[0] = {e[0]:.6f} eV", flush=True)
elif what == "neigh1":           # one setting (under rocprofv3 --kernel-trace --stats): argv[2] = max_neigh
    mn = int(sys.argv[2])
    ms, ne, md, e, f = timed(2000, 16, 3, None, mn)
    print(f"c3: max_neigh={mn}: {ms:.2f} ms per E+F of 16 images, {ne} directed edges, max degree {md}", flush=True)
