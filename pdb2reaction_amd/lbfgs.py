"""Batched L-BFGS relaxation: K independent geometries of the same atoms advance together, ONE batched E+F call per
cycle.

The reference relaxes one geometry at a time with pysisyphus' ``LBFGS`` (third-party; call sites
``align_freeze_atoms.py:468-476,494-503`` with ``max_cycles``, ``thresh="gau"``, ``dump=False`` and
``path_opt.py:483-489``) -- K serial device round trips per cycle.  Here every image keeps its own (s, y) history and
its own convergence flag, and the still-active images are evaluated through ``calc.get_forces_batch``.

Restated pysisyphus conventions [3P-UNVERIFIED, from its public documentation]: Cartesian coordinates in Bohr; the
Gaussian-style threshold sets of ``gsm.THRESH`` = (max|F|, rms F, max|step|, rms step), all four needed (the step
criteria are skipped on the very first cycle when the forces already satisfy theirs); the step is scaled down so that
its largest component does not exceed ``max_step`` (0.2 Bohr); frozen atoms never move.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Union

import numpy as np

from ._host import with_small_host_math
from .gsm import THRESH


def _two_loop(g: np.ndarray, s_hist, y_hist, beta: float) -> np.ndarray:
    """-H g by the L-BFGS two-loop recursion (H0 = gamma I, gamma = s.y / y.y of the newest pair, or beta)."""
    q = g.copy()
    alphas = []
    for s, y in zip(reversed(s_hist), reversed(y_hist)):
        rho = 1.0 / float(y @ s)
        a = rho * float(s @ q)
        alphas.append((a, rho))
        q -= a * y
    if s_hist:
        s, y = s_hist[-1], y_hist[-1]
        q *= float(s @ y) / float(y @ y)
    else:
        q *= beta
    for (a, rho), s, y in zip(reversed(alphas), s_hist, y_hist):
        b = rho * float(y @ q)
        q += (a - b) * s
    return -q


class BatchedLBFGS:
    """Relax K geometries (K, N, 3) Bohr with a shared calculator.

    ``calc`` needs ``get_forces_batch(elem, coords (k, 3N)) -> {"energy": (k,), "forces": (k, 3N)}`` (``uma_pysis`` and
    ``HarmonicBias`` have it); a calculator with only ``get_forces`` is called per image.  ``freeze`` = indices of atoms
    held fixed (one list for all images, or one list per image).
    """

    def __init__(self, calc, elem: Sequence[str], coords_bohr: np.ndarray, *, freeze: Optional[Union[Sequence[int], Sequence[Sequence[int]]]] = None,
                 thresh: Union[str, tuple] = "gau", max_cycles: int = 50, max_step: float = 0.2, keep_last: int = 7, beta: float = 1.0):
        self.calc = calc
        if hasattr(calc, "reserve_images"):               # a long run of K-image batches: one workspace allocation (umx_reserve_images)
            calc.reserve_images(np.asarray(coords_bohr).shape[0] if np.asarray(coords_bohr).ndim == 3 else 1)
        self.elem = list(elem)
        x = np.array(coords_bohr, dtype=np.float64)
        if x.ndim == 2:
            x = x[None]
        self.k, self.n = x.shape[0], x.shape[1]
        self.x = x.reshape(self.k, -1)
        self.thresh = THRESH[thresh] if isinstance(thresh, str) else tuple(float(t) for t in thresh)
        self.max_step, self.keep_last, self.beta = float(max_step), int(keep_last), float(beta)
        self.max_cycles = np.broadcast_to(np.asarray(max_cycles, dtype=int), (self.k,)).copy()        # scalar or one budget per image
        self.active = np.ones((self.k, self.n), dtype=bool)
        if freeze is not None and len(freeze) > 0:
            per_image = isinstance(freeze[0], (list, tuple, np.ndarray))
            for i in range(self.k):
                idx = np.asarray(freeze[i] if per_image else freeze, dtype=int)
                idx = idx[(idx >= 0) & (idx < self.n)]
                self.active[i, idx] = False
        self.dof = np.repeat(self.active, 3, axis=1)                       # (K, 3N)
        self.energies = np.full(self.k, np.nan)
        self.forces = np.zeros_like(self.x)
        self.converged = np.zeros(self.k, dtype=bool)
        self.cycles = np.zeros(self.k, dtype=int)
        self.n_calls = 0

    def _eval(self, which: np.ndarray) -> None:
        ids = np.flatnonzero(which)
        if ids.size == 0:
            return
        self.n_calls += 1
        if hasattr(self.calc, "get_forces_batch"):
            res = self.calc.get_forces_batch(self.elem, self.x[ids])
            e, f = np.asarray(res["energy"], dtype=float).reshape(-1), np.asarray(res["forces"], dtype=float).reshape(ids.size, -1)
        else:
            out = [self.calc.get_forces(self.elem, self.x[i]) for i in ids]
            e = np.array([float(o["energy"]) for o in out]); f = np.stack([np.asarray(o["forces"], dtype=float).reshape(-1) for o in out])
        self.energies[ids] = e
        self.forces[ids] = np.where(self.dof[ids], f, 0.0)

    def _check(self, i: int, step: Optional[np.ndarray]) -> bool:
        max_f, rms_f, max_s, rms_s = self.thresh
        m = self.dof[i]
        if not m.any():
            return True
        f = self.forces[i][m]
        ok = np.abs(f).max() <= max_f and np.sqrt(np.mean(f * f)) <= rms_f
        if ok and step is not None:
            s = step[m]
            ok = np.abs(s).max() <= max_s and np.sqrt(np.mean(s * s)) <= rms_s
        return bool(ok)

    @with_small_host_math
    def run(self) -> Dict[str, np.ndarray]:
        s_hist = [[] for _ in range(self.k)]
        y_hist = [[] for _ in range(self.k)]
        last_step = [None] * self.k
        todo = ~self.converged
        self._eval(todo)
        for _ in range(int(self.max_cycles.max()) if self.k else 0):
            for i in np.flatnonzero(todo):
                if self._check(i, last_step[i]):
                    self.converged[i] = True
            todo = ~self.converged & (self.cycles < self.max_cycles)
            if not todo.any():
                break
            x_old, f_old = self.x.copy(), self.forces.copy()
            for i in np.flatnonzero(todo):
                step = _two_loop(-self.forces[i], s_hist[i], y_hist[i], self.beta)      # gradient = -forces
                step = np.where(self.dof[i], step, 0.0)
                if float(step @ self.forces[i]) <= 0.0:                                   # not a descent direction: restart
                    s_hist[i].clear(); y_hist[i].clear()
                    step = self.beta * self.forces[i]
                big = np.abs(step).max()
                if big > self.max_step:
                    step *= self.max_step / big
                self.x[i] += step
                last_step[i] = step
                self.cycles[i] += 1
            self._eval(todo)
            for i in np.flatnonzero(todo):
                s = self.x[i] - x_old[i]
                y = -(self.forces[i] - f_old[i])
                if float(s @ y) > 1e-12 * float(np.linalg.norm(s) * np.linalg.norm(y) + 1e-300):   # curvature condition
                    s_hist[i].append(s); y_hist[i].append(y)
                    if len(s_hist[i]) > self.keep_last:
                        s_hist[i].pop(0); y_hist[i].pop(0)
        for i in np.flatnonzero(~self.converged):            # images that ran out of cycles: final verdict on the last point
            self.converged[i] = self._check(i, last_step[i])
        return {"coords": self.x.reshape(self.k, self.n, 3).copy(), "energies": self.energies.copy(), "forces": self.forces.copy(),
                "cycles": self.cycles.copy(), "converged": self.converged.copy(), "n_calls": self.n_calls}
