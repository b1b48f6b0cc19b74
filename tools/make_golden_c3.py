#!/usr/bin/env python3
"""float64 oracle ENERGIES AND FORCES at the BASELINE sizes c3 / c4 (2000 atoms) -> tests/golden/c3c4_n2000.npz.

Uses oracle/chunked.py (the hand-derived reverse pass of oracle/staged.py with edge tensors processed in chunks and
re-derived in the reverse pass; equals autograd through oracle/escn_md_oracle.py to 3e-15 on small systems), so a 2000-atom
image needs a few GB instead of the ~200 GB of autograd state.  Contents:

  c3: images 0 and 9 of synth.make_images(2000, 16)                 -> energy, forces
  c4: images 0 (identical to c3's image 0) and 13 of make_images(2000, 24) -> energy, forces
  c4 Hessian: central-difference columns of the ORACLE forces at c4 image 0 for the DOF listed in `hess_dof`, on the
      float32-rounded displaced geometries the engine sees (h = 1e-3 A, reference uma_pysis.py:600), eV/A^2.

Positions are stored as the float32 values the engine receives (AtomicData.pos is float32, uma_pysis.py:312-322); the oracle
evaluates exactly those values in float64.  ~4 min per evaluation on 8 cores; run from the repository root.
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from pdb2reaction_amd import synth, weights as W  # noqa: E402
from oracle.chunked import ChunkedForces  # noqa: E402

torch.set_num_threads(8)
H_STEP = 1.0e-3
w = W.make_synthetic_weights(0)
cf = ChunkedForces(w, chunk=12288)
t00 = time.time()


def ef(z, p32, tag):
    t = time.time()
    e, f = cf.energy_forces(z, p32.astype(np.float64))
    print(f"{tag}: E = {e!r}  max|F| = {np.abs(f).max():.4f}  sum F = {np.abs(f.sum(0)).max():.2e}  ({time.time() - t:.0f} s, total {time.time() - t00:.0f} s)", flush=True)
    return e, f


z, img16, frozen = synth.make_images(2000, 16)
_, img24, _ = synth.make_images(2000, 24)
assert np.array_equal(img16[0], img24[0])
out = dict(z=z.astype(np.int32), frozen=np.asarray(frozen, dtype=np.int32), charge=0, spin=1, task="omol", weights_seed=0, fd_step=H_STEP)
c3_idx, c4_idx = [0, 9], [0, 13]
c3_pos = img16[c3_idx].astype(np.float32)
c4_pos = img24[c4_idx].astype(np.float32)
res = {}
for tag, k, p in (("c3[0]", ("c3", 0), c3_pos[0]), ("c3[9]", ("c3", 1), c3_pos[1]), ("c4[13]", ("c4", 1), c4_pos[1])):
    res[k] = ef(z, p, tag)
res[("c4", 0)] = res[("c3", 0)]
out.update(c3_index=np.array(c3_idx), c3_pos=c3_pos, c3_energy=np.array([res[("c3", i)][0] for i in range(2)]),
           c3_forces=np.stack([res[("c3", i)][1] for i in range(2)]),
           c4_index=np.array(c4_idx), c4_pos=c4_pos, c4_energy=np.array([res[("c4", i)][0] for i in range(2)]),
           c4_forces=np.stack([res[("c4", i)][1] for i in range(2)]))
np.savez_compressed("tests/golden/c3c4_n2000.npz", **out)           # forces first: the Hessian columns take another 15 min
# Hessian columns at c4 image 0: DOF (atom, component) -- the innermost atom and one further out, both active
order = np.argsort(np.einsum("ij,ij->i", img24[0], img24[0]))
dofs = [(int(order[0]), 0), (int(order[40]), 2)]
base64 = img24[0].astype(np.float64)
cols = []
for a, c in dofs:
    fpm = []
    for sgn in (+1.0, -1.0):
        d = base64.copy()
        d[a, c] += sgn * H_STEP
        fpm.append(ef(z, d.astype(np.float32), f"hess dof ({a},{c}) {'+' if sgn > 0 else '-'}")[1])
    cols.append((-(fpm[0] - fpm[1]) / (2.0 * H_STEP)).reshape(-1))
out.update(hess_dof=np.array([3 * a + c for a, c in dofs]), hess_cols=np.stack(cols))
np.savez_compressed("tests/golden/c3c4_n2000.npz", **out)
print("wrote tests/golden/c3c4_n2000.npz", flush=True)
