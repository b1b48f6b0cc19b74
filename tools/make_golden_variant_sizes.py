#!/usr/bin/env python3
"""float64 oracle energy and forces of ONE image at a BASELINE size for a MODEL VARIANT (round 6, end): the grid feed-forward at 2000 and 20 000 atoms.

    python tools/make_golden_variant_sizes.py grid 2000 [weights seed = 0]     -> tests/golden/c3_n2000_grid_w<seed>.npz      (image 5 of make_images(2000, 16))
    python tools/make_golden_variant_sizes.py grid 20000 [weights seed = 0]    -> tests/golden/c5_n20000_grid_w<seed>.npz     (image 0 of make_images(20000, 8))

The variant parity tests (tests/test_gpu_variants.py) stop at 500 atoms; whether a coherent per-atom error hides in the grid form's node-level GEMMs
only shows at the sizes where 1e-4 eV is 5e-9 eV per atom.  oracle/chunked.py (its atom-wise block is staged.atomwise_fwd / _bwd: both forms)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from pdb2reaction_amd import synth, weights as W  # noqa: E402
from oracle.chunked import ChunkedForces  # noqa: E402

variant, n = sys.argv[1], int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
assert variant == "grid" and n in (2000, 20000)
torch.set_num_threads(8)
k, image = (16, 5) if n == 2000 else (8, 0)
z, imgs, _ = synth.make_images(n, k)
p32 = imgs[image].astype(np.float32)
w = W.make_synthetic_weights(seed, ff_type="grid")
t0 = time.time()
e, f = ChunkedForces(w, chunk=12288).energy_forces(z, p32.astype(np.float64), log=lambda m: print(f"  {m}  ({time.time() - t0:.0f} s)", flush=True))
print(f"{variant} N = {n} weights seed {seed}: E = {e!r}  max|F| = {np.abs(f).max():.4f}  sum F = {np.abs(f.sum(0)).max():.2e}  ({time.time() - t0:.0f} s)", flush=True)
name = f"tests/golden/{'c3_n2000' if n == 2000 else 'c5_n20000'}_{variant}_w{seed}.npz"
np.savez_compressed(name, z=z.astype(np.int32), pos=p32[None] if n == 2000 else p32, energy=np.array([e]), forces=f[None], image_index=np.array([image]),
                    charge=0, spin=1, task="omol", weights_seed=seed, variant=variant)
print("wrote", name, flush=True)
