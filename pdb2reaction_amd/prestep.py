"""Pre-step helpers either side of the string loop: rigid alignment and the harmonic
distance-restraint wrapper, restated from the reference with vectorised / batched implementations.

* ``kabsch_R_t``: row-vector Kabsch fit ``Q @ R + t ~ P`` with the improper-rotation fix -- reference
  ``align_freeze_atoms.py:128-145``.
* ``HarmonicBias``: wraps a calculator and adds ``0.5 k (|r_i - r_j| - d0)^2`` wells (k in eV/A^2, targets in Angstrom,
  coordinates in Bohr, results in Hartree / Hartree/Bohr) -- reference ``opt.py:286-343`` (``HarmonicBiasCalculator``):
  out-of-range pairs and coincident atoms are skipped, unknown attributes are forwarded to the wrapped calculator.
  Added here: ``get_forces_batch`` so biased strings keep the one-launch-per-cycle path.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from ._calculator_base import ANG2BOHR
from .hessian import EV_PER_ANG2_TO_AU


def kabsch_R_t(P: np.ndarray, Q: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """R (3x3), t (3,) minimising ||(Q @ R + t) - P|| over proper rotations, P and Q of shape (N, 3)."""
    p = np.asarray(P, dtype=float)
    q = np.asarray(Q, dtype=float)
    if p.shape != q.shape or p.ndim != 2 or p.shape[1] != 3:
        raise ValueError("Kabsch expects P, Q with shape (N, 3).")
    cp, cq = p.mean(axis=0), q.mean(axis=0)
    u, _, vt = np.linalg.svd((p - cp).T @ (q - cq))
    rot = vt.T @ u.T
    if np.linalg.det(rot) < 0.0:          # reflection: flip the axis of the smallest singular value
        vt = vt.copy()
        vt[-1] *= -1.0
        rot = vt.T @ u.T
    return rot, cp - cq @ rot


def align_onto(reference: np.ndarray, mobile: np.ndarray, subset: Optional[Sequence[int]] = None) -> np.ndarray:
    """`mobile` rigidly fitted onto `reference` (optionally using only the atoms in `subset` for the fit)."""
    ref = np.asarray(reference, dtype=float).reshape(-1, 3)
    mob = np.asarray(mobile, dtype=float).reshape(-1, 3)
    idx = slice(None) if subset is None else np.asarray(list(subset), dtype=int)
    rot, t = kabsch_R_t(ref[idx], mob[idx])
    return mob @ rot + t


class HarmonicBias:
    """Wrap a calculator with harmonic distance restraints (reference ``HarmonicBiasCalculator``, ``opt.py:286-343``)."""

    def __init__(self, base_calc, k: float = 10.0, pairs: Optional[List[Tuple[int, int, float]]] = None):
        self.base = base_calc
        self.k_evAA = float(k)
        self.k_au_bohr2 = self.k_evAA * EV_PER_ANG2_TO_AU
        self._pairs: List[Tuple[int, int, float]] = []
        self.set_pairs(list(pairs or []))

    def set_pairs(self, pairs: List[Tuple[int, int, float]]) -> None:
        self._pairs = [(int(i), int(j), float(t)) for (i, j, t) in pairs]

    def _bias(self, coords_bohr: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """Vectorised over a batch: coords (K, N, 3) Bohr -> (E_bias (K,), F_bias (K, N, 3))."""
        c = np.asarray(coords_bohr, dtype=float)
        k_img, n = c.shape[0], c.shape[1]
        e = np.zeros(k_img)
        f = np.zeros_like(c)
        valid = [(i, j, t) for (i, j, t) in self._pairs if 0 <= i < n and 0 <= j < n]
        if not valid:
            return e, f
        i_idx = np.array([p[0] for p in valid]); j_idx = np.array([p[1] for p in valid])
        target = np.array([p[2] for p in valid]) * ANG2BOHR
        rij = c[:, i_idx, :] - c[:, j_idx, :]                           # (K, P, 3)
        d = np.linalg.norm(rij, axis=2)
        ok = d >= 1e-14
        diff = np.where(ok, d - target[None, :], 0.0)
        e = 0.5 * self.k_au_bohr2 * (diff ** 2).sum(axis=1)
        fi = -self.k_au_bohr2 * diff[..., None] * rij / np.maximum(d, 1e-14)[..., None]
        for p in range(len(valid)):                                      # a pair list is short; atoms may repeat
            f[:, i_idx[p], :] += fi[:, p, :]
            f[:, j_idx[p], :] -= fi[:, p, :]
        return e, f

    # ---- calculator protocol ---------------------------------------------------------------------------
    def get_forces(self, elem, coords):
        cb = np.asarray(coords, dtype=float).reshape(1, -1, 3)
        base = self.base.get_forces(elem, cb[0])
        eb, fb = self._bias(cb)
        return {"energy": float(base["energy"]) + float(eb[0]), "forces": np.asarray(base["forces"], dtype=float).reshape(-1) + fb[0].reshape(-1)}

    def get_energy(self, elem, coords):
        cb = np.asarray(coords, dtype=float).reshape(1, -1, 3)
        return {"energy": float(self.base.get_energy(elem, cb[0])["energy"]) + float(self._bias(cb)[0][0])}

    def get_forces_batch(self, elem, coords_batch):
        c = np.asarray(coords_batch, dtype=float)
        k = c.shape[0]
        base = self.base.get_forces_batch(elem, c.reshape(k, -1))
        eb, fb = self._bias(c.reshape(k, -1, 3))
        return {"energy": np.asarray(base["energy"], dtype=float) + eb, "forces": np.asarray(base["forces"], dtype=float).reshape(k, -1) + fb.reshape(k, -1)}

    def get_energy_and_forces(self, elem, coords):
        res = self.get_forces(elem, coords)
        return res["energy"], res["forces"]

    def get_energy_and_gradient(self, elem, coords):
        res = self.get_forces(elem, coords)
        return res["energy"], -np.asarray(res["forces"], dtype=float).reshape(-1)

    def __getattr__(self, name: str):
        return getattr(self.base, name)
