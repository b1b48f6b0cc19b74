"""A toy analytic ``UMAcore`` for pinning the calculator boundary to the reference's own methods (TEST INFRASTRUCTURE).

Both sides of ``tests/golden/ref_uma_pysis_methods.json`` run on THIS core: ``tools/make_reference_fixtures.py`` puts it behind
the ``ast``-compiled bodies of the reference's ``uma_pysis`` methods (``pdb2reaction/uma_pysis.py:502-780``) and records what they
return; ``tests/test_reference_uma_pysis.py`` puts it behind ``pdb2reaction_amd.uma_pysis`` and compares.

The contract is ``UMAcore.compute``'s (reference ``:330-419``): Angstrom float64 in, ``{"energy": float eV, "forces": (N,3)
float32 eV/A | None, "hessian": (N,3,N,3) float32 torch | None}`` out; positions are quantised to float32 first, as
``AtomicData.pos`` is -- this is what puts the reference's finite-difference noise floor into the fixture.
The potential is a spring + inverse-square pair potential evaluated with scalar IEEE double ``+ - * / sqrt`` in a fixed order,
so the recorded numbers do not depend on a BLAS, a SIMD width or libm.
"""
from __future__ import annotations

import math

import numpy as np
import torch


class ToyPairCore:
    def __init__(self, n_atoms: int, *, seed: int = 0, parallel_predict: bool = False, has_torch_model: bool = True):
        rng = np.random.default_rng(seed)
        self.n = int(n_atoms)
        self.k = rng.uniform(2.0, 9.0, size=(self.n, self.n))          # spring constants, eV/A^2
        self.r0 = rng.uniform(1.0, 2.2, size=(self.n, self.n))         # rest lengths, A
        self.c = rng.uniform(0.1, 0.8, size=(self.n, self.n))          # inverse-square strengths, eV A^2
        self.parallel_predict = bool(parallel_predict)                 # reference UMAcore attributes read by get_hessian (:736)
        self.has_torch_model = bool(has_torch_model)
        self.device = torch.device("cpu")
        self.calls = 0
        self.seen = []                                                 # every geometry handed in (float64, as received)

    # ------------------------------------------------------------------
    def _evaluate(self, coord_ang, want_hessian: bool):
        pos = np.asarray(coord_ang, dtype=np.float64).reshape(self.n, 3).astype(np.float32).astype(np.float64)
        n = self.n
        e = 0.0
        f = [[0.0, 0.0, 0.0] for _ in range(n)]
        h = np.zeros((n, 3, n, 3), dtype=np.float64) if want_hessian else None
        for i in range(n):
            for j in range(i + 1, n):
                d = [float(pos[i, a]) - float(pos[j, a]) for a in range(3)]
                r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]
                r = math.sqrt(r2)
                k, r0, c = float(self.k[i, j]), float(self.r0[i, j]), float(self.c[i, j])
                e += 0.5 * k * (r - r0) * (r - r0) + c / r2
                de = k * (r - r0) - 2.0 * c / (r2 * r)                  # dE/dr
                for a in range(3):
                    g = de * d[a] / r
                    f[i][a] -= g
                    f[j][a] += g
                if want_hessian:
                    d2e = k + 6.0 * c / (r2 * r2)                       # d2E/dr2
                    for a in range(3):
                        for b in range(3):
                            uu = d[a] * d[b] / r2
                            blk = d2e * uu + (de / r) * ((1.0 if a == b else 0.0) - uu)
                            h[i, a, i, b] += blk
                            h[j, a, j, b] += blk
                            h[i, a, j, b] -= blk
                            h[j, a, i, b] -= blk
        return e, np.asarray(f, dtype=np.float64).astype(np.float32), h

    def compute(self, coord_ang, *, forces: bool = False, hessian: bool = False):
        if hessian and (self.parallel_predict or not self.has_torch_model):
            raise RuntimeError(
                "Analytical Hessian is not available when predictor workers > 1 "
                "or when predictor.model is not exposed. Use FiniteDifference Hessian."
            )
        self.calls += 1
        self.seen.append(np.array(coord_ang, dtype=np.float64, copy=True))
        e, f, h = self._evaluate(coord_ang, hessian)
        out = {"energy": float(e), "forces": f if (forces or hessian) else None, "hessian": None}
        if hessian:
            out["hessian"] = torch.from_numpy(h.astype(np.float32))     # model dtype, (N,3,N,3), like reference :411-417
        return out

    def compute_batch(self, coords_ang, *, forces: bool = True):
        """The batched entry point this repo adds (``pdb2reaction_amd.uma_pysis.UMAcore.compute_batch``)."""
        c = np.asarray(coords_ang, dtype=np.float64).reshape(-1, self.n, 3)
        res = [self.compute(c[k], forces=forces) for k in range(c.shape[0])]
        return {"energy": np.asarray([r["energy"] for r in res], dtype=np.float64),
                "forces": np.stack([r["forces"] for r in res]) if forces else None}


def toy_geometry(n_atoms: int, seed: int = 0) -> np.ndarray:
    """(N,3) Angstrom float64: a jittered chain with neighbour distances of 1.2-2.0 A (no close contacts)."""
    rng = np.random.default_rng(1000 + seed)
    base = np.stack([1.45 * np.arange(n_atoms), 0.6 * (np.arange(n_atoms) % 2), 0.35 * (np.arange(n_atoms) % 3)], axis=1)
    return base + rng.uniform(-0.18, 0.18, size=(n_atoms, 3))
