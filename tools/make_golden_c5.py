#!/usr/bin/env python3
"""float64 oracle energy and forces of ONE c5-size image (20 000 atoms, 1.6 M directed edges) -> tests/golden/c5_n20000.npz.

Image 0 of synth.make_images(20000, 8) (BASELINE configs[4]); oracle/chunked.py with the blocked radius graph; about 25 minutes on
8 cores and ~10 GB.  Positions are stored as the float32 values the engine receives."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from pdb2reaction_amd import synth, weights as W  # noqa: E402
from oracle.chunked import ChunkedForces  # noqa: E402

torch.set_num_threads(8)
z, imgs, frozen = synth.make_images(20000, 8)
p32 = imgs[0].astype(np.float32)
cf = ChunkedForces(W.make_synthetic_weights(0), chunk=12288)
t0 = time.time()
e, f = cf.energy_forces(z, p32.astype(np.float64), log=lambda m: print(f"  {m}  ({time.time() - t0:.0f} s)", flush=True))
print(f"c5[0]: E = {e!r}  max|F| = {np.abs(f).max():.4f}  sum F = {np.abs(f.sum(0)).max():.2e}  ({time.time() - t0:.0f} s)", flush=True)
np.savez_compressed("tests/golden/c5_n20000.npz", z=z.astype(np.int32), pos=p32, energy=np.array([e]), forces=f[None], image_index=np.array([0]),
                    charge=0, spin=1, task="omol", weights_seed=0)
print("wrote tests/golden/c5_n20000.npz", flush=True)
