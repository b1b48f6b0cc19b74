"""Why does a batched call take 97 ms inside a growing-string run and 55 ms in a tight loop?  (dev)  Variants by argv[1]:
plain | torch (torch.cuda initialised first) | sleep (5 ms of host sleep between calls) | numpy (5 ms of numpy work between calls) |
calc (through uma_pysis.get_forces_batch) | calcnumpy (calc + numpy work: the calculator caps the BLAS pools) |
numpy<N> (numpy with the pools limited to N threads)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode in ("torch", "calc", "calcnumpy"):
    import torch
    torch.zeros(1, device="cuda")
from pdb2reaction_amd import synth, weights as W  # noqa: E402
from pdb2reaction_amd.engine import Engine  # noqa: E402

z, imgs, _ = synth.make_images(500, 12)
x10 = imgs[:10].copy()
if mode in ("calc", "calcnumpy"):
    import importlib
    U = importlib.import_module("pdb2reaction_amd.uma_pysis")
    calc = U.uma_pysis(model="synthetic")
    elem = [synth.SYMBOLS[int(a)] for a in z]
    call = lambda x: calc.get_forces_batch(elem, x.astype(np.float64) * U.ANG2BOHR)
else:
    eng = Engine(0)
    eng.load_weights(W.make_synthetic_weights(0))
    eng.set_system(z)
    call = lambda x: eng.energy_forces(x)
call(imgs)
if mode.startswith("numpy") and mode != "numpy":
    from threadpoolctl import threadpool_limits
    nthr = int(mode[5:])
    threadpool_limits(limits=nthr)
    mode = "numpy"
    label = f"numpy, BLAS limited to {nthr} thread(s)"
else:
    label = mode
import os
print("cpus:", os.cpu_count(), "affinity:", len(os.sched_getaffinity(0)), open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "no cpu.max")
out = []
a = np.random.default_rng(0).normal(size=(600, 600))
for i in range(12):
    if mode == "sleep":
        time.sleep(0.005)
    if mode in ("numpy", "calcnumpy"):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.005:
            a @ a
    t = time.perf_counter()
    call(x10)
    out.append((time.perf_counter() - t) * 1e3)
print(f"{label}: " + " ".join(f"{t:.0f}" for t in out), flush=True)
