#!/usr/bin/env python3
"""float64 oracle energy and forces of ONE c5-size image (20 000 atoms, 1.6 M directed edges) -> tests/golden/c5_n20000*.npz.

    python tools/make_golden_c5.py                       # image 0 of synth.make_images(20000, 8) (BASELINE configs[4]), weights seed 0 -> c5_n20000.npz
    python tools/make_golden_c5.py g1 20260230 0 3       # another GEOMETRY: cluster seed 20260230, image 3, weights seed 0 -> c5_n20000_g1.npz
    python tools/make_golden_c5.py w1 20260130 1 0       # another WEIGHT SET: the BASELINE geometry, synthetic weights seed 1 -> c5_n20000_w1.npz

oracle/chunked.py with the blocked radius graph; 13-25 minutes on 8 cores and ~10 GB.  Positions are stored as the float32 values the
engine receives.  The two extra fixtures (round 5) widen the evidence behind the 1e-4 eV energy bound at 20 000 atoms:
the sign-alternating operand rows cancel a one-sided matrix-core rounding error statistically, so one geometry and one weight set is thin."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from pdb2reaction_amd import synth, weights as W  # noqa: E402
from oracle.chunked import ChunkedForces  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else ""
geom_seed = int(sys.argv[2]) if len(sys.argv) > 2 else synth.DEFAULT_SEED
wseed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
image = int(sys.argv[4]) if len(sys.argv) > 4 else 0
torch.set_num_threads(int(sys.argv[5]) if len(sys.argv) > 5 else 8)
z, imgs, frozen = synth.make_images(20000, 8, seed=geom_seed)
p32 = imgs[image].astype(np.float32)
cf = ChunkedForces(W.make_synthetic_weights(wseed), chunk=12288)
t0 = time.time()
e, f = cf.energy_forces(z, p32.astype(np.float64), log=lambda m: print(f"  {m}  ({time.time() - t0:.0f} s)", flush=True))
print(f"c5[{image}] (cluster seed {geom_seed}, weights seed {wseed}): E = {e!r}  max|F| = {np.abs(f).max():.4f}  sum F = {np.abs(f.sum(0)).max():.2e}  ({time.time() - t0:.0f} s)", flush=True)
name = "tests/golden/c5_n20000" + (f"_{tag}" if tag else "") + ".npz"
np.savez_compressed(name, z=z.astype(np.int32), pos=p32, energy=np.array([e]), forces=f[None], image_index=np.array([image]),
                    charge=0, spin=1, task="omol", weights_seed=wseed, cluster_seed=geom_seed)
print("wrote", name, flush=True)
