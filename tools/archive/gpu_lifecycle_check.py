"""Lifecycle soak: engines created / destroyed repeatedly, systems of different sizes re-bound on one engine, device memory watched.
usage: python tools/gpu_lifecycle_check.py"""
import sys
import numpy as np
import torch

sys.path.insert(0, ".")
from pdb2reaction_amd import synth, weights as W  # noqa: E402
from pdb2reaction_amd.engine import Engine  # noqa: E402

w = W.make_synthetic_weights(0)
torch.cuda.init()
base = torch.cuda.mem_get_info()[0]
held = []
for rep in range(6):                                   # create / evaluate / destroy
    eng = Engine(0); eng.load_weights(w)
    z, imgs, _ = synth.make_images(200 + 50 * rep, 3, seed=rep)
    eng.set_system(z)
    e, f = eng.energy_forces(imgs)
    assert np.isfinite(e).all() and np.isfinite(f).all()
    eng.close()
    left = base - torch.cuda.mem_get_info()[0]
    print(f"rep {rep}: after close {left / 2**20:.1f} MiB held")
    held = held + [left] if rep else [left]            # (the first use keeps code objects / runtime pools: watch the GROWTH)
    assert left - held[0] < 32 * 2**20, held
eng = Engine(0); eng.load_weights(w)
ref = {}
for rnd in range(3):                                   # one engine, systems of changing size, results reproducible bit for bit
    for n in (40, 700, 150, 1200, 9):
        z, imgs, _ = synth.make_images(n, 2, seed=n)
        eng.set_system(z, charge=(-1 if n == 150 else 0), spin=(2 if n == 150 else 1))
        e, f = eng.energy_forces(imgs)
        key = n
        if key in ref:
            assert np.array_equal(ref[key][0], e) and np.array_equal(ref[key][1], f), (rnd, n)
        ref[key] = (e, f)
    print(f"round {rnd}: held {(base - torch.cuda.mem_get_info()[0]) / 2**30:.2f} GiB")
eng.close()
print("after final close:", (base - torch.cuda.mem_get_info()[0]) / 2**20, "MiB")
print("OK")
