"""Batched Growing-String driver.

The reference runs ``GrowingString(images, calc_getter, **GS_KW)`` + ``StringOptimizer(gs, **STOPT_KW).run()`` from
pysisyphus (reference ``path_opt.py:959-977``, ``path_search.py:664-681``); pysisyphus evaluates the images one after
another through the shared calculator.  This driver restates that loop so that ALL images needing an evaluation in a
cycle go through ONE ``calc.get_forces_batch`` call (and, across GPUs, one sharded call + all-gather).

PARITY UNPINNED: pysisyphus is not installed here, so the loop follows the published GSM semantics summarised in
SURVEY.md Appendix B -- frontier growth at ``perp_thresh``, equal-arc ("equi") reparametrisation, perpendicular-force
steps scaled to ``max_step``, climbing image once the fully grown string is below ``climb_rms``, convergence on the
``thresh`` presets of reference ``opt.py:176-187`` -- not pysisyphus' source.  Keyword names and defaults are those of
reference ``GS_KW`` / ``STOPT_KW`` (``path_opt.py:168-200``).  Tangents come from a parametric cubic spline through the
images (Appendix B; central differences for fewer than four images, ``tangent="central"`` forces them); with
``climb_lanczos`` (the reference default, ``path_opt.py:181-182``) the climbing image's tangent is the lowest-curvature mode
from a Lanczos iteration on finite-difference Hessian-vector products once the string is below ``climb_lanczos_rms``
(:func:`lanczos_lowest_mode`; one single-image evaluation per Lanczos step).  Coordinates are Cartesian (no DLC).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional, Sequence

import numpy as np

from ._host import with_small_host_math
from .string import select_hei_index

# reference path_opt.py:168-185
GS_KW: Dict[str, Any] = {
    "fix_first": True, "fix_last": True, "max_nodes": 10, "perp_thresh": 5e-3, "reparam_check": "rms",
    "reparam_every": 1, "reparam_every_full": 1, "param": "equi", "max_micro_cycles": 10, "reset_dlc": True,
    "climb": True, "climb_rms": 5e-4, "climb_lanczos": True, "climb_lanczos_rms": 5e-4, "climb_fixed": False,
    "scheduler": None,
}
# reference path_opt.py:188-200 (+ max_step / thresh from OPT_BASE_KW, opt.py:172-187)
STOPT_KW: Dict[str, Any] = {
    "type": "string", "stop_in_when_full": 300, "align": False, "scale_step": "global", "max_cycles": 300, "dump": False,
    "dump_restart": False, "reparam_thresh": 0.0, "coord_diff_thresh": 0.0, "out_dir": "./result_path_opt/",
    "print_every": 10, "max_step": 0.1, "thresh": "gau_loose",
}
# convergence presets in Hartree/Bohr and Bohr: (max_force, rms_force, max_step, rms_step) -- reference opt.py:176-187
THRESH = {
    "gau_loose": (2.5e-3, 1.7e-3, 1.0e-2, 6.7e-3),
    "gau": (4.5e-4, 3.0e-4, 1.8e-3, 1.2e-3),
    "gau_tight": (1.5e-5, 1.0e-5, 6.0e-5, 4.0e-5),
    "gau_vtight": (2.0e-6, 1.0e-6, 6.0e-6, 4.0e-6),
}


@dataclass
class GSMResult:
    coords: np.ndarray              # (K, 3N) Bohr
    energies: np.ndarray            # (K,) Hartree
    converged: bool
    cycles: int
    fully_grown: bool
    hei_index: int
    force_evaluations: int          # image evaluations (sum over cycles of images in the batch)
    history: List[Dict[str, float]] = field(default_factory=list)


def _tangents_central(x: np.ndarray) -> np.ndarray:
    t = np.empty_like(x)
    t[1:-1] = x[2:] - x[:-2]
    t[0] = x[1] - x[0]
    t[-1] = x[-1] - x[-2]
    return t / np.maximum(np.linalg.norm(t, axis=1, keepdims=True), 1e-30)


def _tangents(x: np.ndarray, kind: str = "spline") -> np.ndarray:
    """Unit tangents at the images.  "spline": derivative of the interpolating parametric cubic spline (not-a-knot) over
    the cumulative chord length -- what a spline through the string gives; needs >= 4 images with
    distinct positions, otherwise (and for "central") central differences with one-sided ends."""
    if kind == "spline" and len(x) >= 4:
        seg = np.linalg.norm(np.diff(x, axis=0), axis=1)
        if np.all(seg > 1e-12):
            from scipy.interpolate import make_interp_spline

            u = np.concatenate([[0.0], np.cumsum(seg)])
            t = make_interp_spline(u, x, k=3)(u, 1)
            return t / np.maximum(np.linalg.norm(t, axis=1, keepdims=True), 1e-30)
    return _tangents_central(x)


def lanczos_lowest_mode(grad_fn: Callable[[np.ndarray], np.ndarray], x: np.ndarray, g0: np.ndarray, guess: np.ndarray, *,
                        dx: float = 5e-3, dl: float = 1e-2, max_cycles: int = 25):
    """Lowest Hessian eigenpair at x from a Lanczos recursion on forward-difference Hessian-vector products
    H q ~ (g(x + dx q) - g(x)) / dx; one gradient per step, started from `guess` (the string tangent).

    Returns (eigenvalue, unit eigenvector, gradient evaluations).  Stops when the lowest Ritz value changes by less than
    `dl` (relative) between two steps, when the Krylov space is exhausted, or after `max_cycles` steps.  The vector is
    oriented along `guess`."""
    n = x.size
    r = np.asarray(guess, dtype=np.float64).reshape(-1).copy()
    beta = float(np.linalg.norm(r))
    if beta < 1e-14:
        raise ValueError("lanczos: zero start vector")
    qs: List[np.ndarray] = []
    alphas: List[float] = []
    betas: List[float] = []
    q_prev = np.zeros(n)
    w_prev: Optional[float] = None
    w_min, v_min = 0.0, r / beta
    steps = 0
    for steps in range(1, min(int(max_cycles), n) + 1):
        q = r / beta
        for qq in qs:                                  # full re-orthogonalisation: the space is small and FD noise is not
            q -= (qq @ q) * qq
        q /= max(float(np.linalg.norm(q)), 1e-30)
        u = (grad_fn(x + dx * q) - g0) / dx - beta * q_prev if qs else (grad_fn(x + dx * q) - g0) / dx
        alpha = float(q @ u)
        r = u - alpha * q
        qs.append(q); alphas.append(alpha)
        t = np.diag(alphas) + np.diag(betas, 1) + np.diag(betas, -1)
        w, v = np.linalg.eigh(t)
        w_min = float(w[0])
        v_min = np.stack(qs, axis=1) @ v[:, 0]
        beta = float(np.linalg.norm(r))
        if w_prev is not None and abs(w_min - w_prev) <= dl * max(abs(w_prev), 1e-12):
            break
        if beta < 1e-10:
            break
        w_prev, q_prev = w_min, q
        betas.append(beta)
    v_min = v_min / max(float(np.linalg.norm(v_min)), 1e-30)
    if float(v_min @ np.asarray(guess, dtype=np.float64).reshape(-1)) < 0.0:
        v_min = -v_min
    return w_min, v_min, steps


def _place(x: np.ndarray, targets: np.ndarray) -> np.ndarray:
    """Points at normalised arc-length positions `targets` (in [0,1]) along the polyline through x."""
    seg = np.linalg.norm(x[1:] - x[:-1], axis=1)
    s = np.concatenate([[0.0], np.cumsum(seg)])
    total = s[-1] if s[-1] > 0 else 1.0
    out = np.empty((len(targets), x.shape[1]))
    for n, tg in enumerate(targets):
        u = tg * total
        i = int(np.clip(np.searchsorted(s, u, side="right"), 1, len(s) - 1))
        w = (u - s[i - 1]) / max(s[i] - s[i - 1], 1e-30)
        out[n] = x[i - 1] * (1 - w) + x[i] * w
    return out


class GrowingStringDriver:
    """Growing string between two endpoints; every cycle issues one batched E+F call.

    calc: object with ``get_forces_batch(atoms, coords[K,3N] Bohr) -> {"energy": (K,), "forces": (K,3N)}``
    (``pdb2reaction_amd.uma_pysis.uma_pysis`` or any stand-in); ``evaluate`` may override how a batch is evaluated
    (e.g. ``ShardedImageEvaluator`` for one-process-per-GPU sharding).
    """

    def __init__(self, atoms: Sequence[str], reactant: np.ndarray, product: np.ndarray, calc: Any = None,
                 evaluate: Optional[Callable[[np.ndarray], Any]] = None, gs_kw: Optional[Dict[str, Any]] = None,
                 stopt_kw: Optional[Dict[str, Any]] = None, log: Optional[Callable[[str], None]] = None):
        self.atoms = list(atoms)
        self.gs = {**GS_KW, **(gs_kw or {})}
        self.opt = {**STOPT_KW, **(stopt_kw or {})}
        if self.gs["param"] != "equi":
            raise NotImplementedError("only param='equi' (equal arc length) is implemented")
        r = np.asarray(reactant, dtype=np.float64).reshape(-1)
        p = np.asarray(product, dtype=np.float64).reshape(-1)
        if r.shape != p.shape or r.size != 3 * len(self.atoms):
            raise ValueError("reactant/product must both be (3N,) for the given atoms")
        self.calc, self._evaluate, self.log = calc, evaluate, (log or (lambda s: None))
        self.max_images = int(self.gs["max_nodes"]) + 2
        if hasattr(calc, "reserve_images"):                  # the string grows to max_images: one workspace allocation instead of one per growth
            calc.reserve_images(self.max_images)
        step = 1.0 / (self.max_images - 1)
        if self.max_images <= 3:
            self.left, self.right = [r], [p]
            if self.max_images == 3:
                self.left.append(0.5 * (r + p))
        else:                                                # endpoints + first frontier node on either side
            self.left = [r, r + step * (p - r)]
            self.right = [p - step * (p - r), p]
        self.energies: Optional[np.ndarray] = None
        self.forces: Optional[np.ndarray] = None
        self.n_eval = 0
        self.lanczos_evals = 0
        self.tangent_kind = str(self.gs.get("tangent", "spline"))
        self._lbfgs_s: List[np.ndarray] = []
        self._lbfgs_y: List[np.ndarray] = []
        self._prev = None

    # ---- helpers -------------------------------------------------------------------------------------
    @property
    def coords(self) -> np.ndarray:
        return np.stack(self.left + self.right)

    @property
    def fully_grown(self) -> bool:
        return len(self.left) + len(self.right) >= self.max_images

    def _set_coords(self, x: np.ndarray):
        nl = len(self.left)
        self.left = [x[i].copy() for i in range(nl)]
        self.right = [x[i].copy() for i in range(nl, len(x))]

    def _eval(self, x: np.ndarray, need: np.ndarray):
        """Batched E+F for the images flagged in `need`."""
        idx = np.nonzero(need)[0]
        if self.energies is None or len(self.energies) != len(x):
            self.energies, self.forces = np.zeros(len(x)), np.zeros_like(x)
        if len(idx) == 0:
            return
        if self._evaluate is not None:
            e, f = self._evaluate(x[idx])
        else:
            res = self.calc.get_forces_batch(self.atoms, x[idx])
            e, f = res["energy"], res["forces"]
        self.energies[idx] = np.asarray(e, dtype=np.float64)
        self.forces[idx] = np.asarray(f, dtype=np.float64).reshape(len(idx), -1)
        self.n_eval += len(idx)

    def _single_forces(self, xq: np.ndarray) -> np.ndarray:
        """Forces of ONE geometry through the same (batched) evaluator -- the serial steps of the Lanczos recursion."""
        if self._evaluate is not None:
            _, f = self._evaluate(xq[None])
        else:
            f = self.calc.get_forces_batch(self.atoms, xq[None])["forces"]
        self.n_eval += 1
        return np.asarray(f, dtype=np.float64).reshape(-1)

    def _reparametrize(self, x: np.ndarray) -> np.ndarray:
        k, nl = len(x), len(self.left)
        if self.fully_grown:
            tg = np.linspace(0.0, 1.0, k)
        else:                                                # growing: left nodes at i*step, right nodes mirrored from the end
            step = 1.0 / (self.max_images - 1)
            tg = np.concatenate([np.arange(nl) * step, 1.0 - np.arange(k - nl - 1, -1, -1) * step])
        out = _place(x, tg)
        out[0], out[-1] = x[0], x[-1]
        return out

    def _grow(self, x: np.ndarray, fperp_rms: np.ndarray) -> bool:
        """Add a node next to a frontier whose perpendicular force is below perp_thresh; True if the string changed."""
        grew = False
        nl = len(self.left)
        for side in ("left", "right"):
            if self.fully_grown:
                break
            fl, fr = self.left[-1], self.right[0]
            gap_nodes = self.max_images - len(self.left) - len(self.right)
            frontier = nl - 1 if side == "left" else nl
            if fperp_rms[frontier] >= self.gs["perp_thresh"]:
                continue
            d = (fr - fl) / (gap_nodes + 1)
            if side == "left":
                self.left.append(fl + d)
            else:
                self.right.insert(0, fr - d)
            grew = True
        return grew

    # ---- main loop -----------------------------------------------------------------------------------
    @with_small_host_math
    def run(self) -> GSMResult:
        gs, opt = self.gs, self.opt
        max_f, rms_f, _, _ = THRESH[opt["thresh"]] if isinstance(opt["thresh"], str) else opt["thresh"]
        history: List[Dict[str, float]] = []
        converged = False
        full_cycles = 0
        need = None
        climbing = False
        cycle = 0
        stale = False
        for cycle in range(1, int(opt["max_cycles"]) + 1):
            x = self.coords
            k = len(x)
            if need is None or len(need) != k:
                need = np.ones(k, dtype=bool)
            self._eval(x, need)
            stale = False                                       # energies/forces belong to the current coordinates
            moving = np.ones(k, dtype=bool)
            moving[0] = not gs["fix_first"]
            moving[-1] = not gs["fix_last"]
            t = _tangents(x, self.tangent_kind)
            f = self.forces
            fpar = np.einsum("ij,ij->i", f, t)[:, None] * t
            fperp = f - fpar
            fperp[~moving] = 0.0
            n_dof = x.shape[1]
            rms_img = np.sqrt((fperp ** 2).sum(1) / n_dof)
            rms_all = float(np.sqrt((fperp[moving] ** 2).mean())) if moving.any() else 0.0
            max_all = float(np.abs(fperp[moving]).max()) if moving.any() else 0.0
            hei = select_hei_index(self.energies)
            full = self.fully_grown
            if full and gs["climb"] and not climbing and rms_all <= gs["climb_rms"] and 0 < hei < k - 1:
                climbing = True
                self._lbfgs_s.clear(); self._lbfgs_y.clear(); self._prev = None
            step_force = fperp.copy()
            if climbing and 0 < hei < k - 1:
                t_ci = t[hei]
                if gs["climb_lanczos"] and rms_all <= gs["climb_lanczos_rms"]:
                    # lowest-curvature direction at the HEI instead of the string tangent (reference GS_KW climb_lanczos)
                    def grad_at(xq):
                        return -self._single_forces(xq)
                    _, t_ci, n_l = lanczos_lowest_mode(grad_at, x[hei], -f[hei], t[hei])
                    self.lanczos_evals += n_l
                step_force[hei] = f[hei] - 2.0 * float(f[hei] @ t_ci) * t_ci   # invert the component along the climbing tangent
            history.append({"cycle": cycle, "images": k, "rms_fperp": rms_all, "max_fperp": max_all, "e_hei": float(self.energies[hei]),
                            "climbing": float(climbing)})
            if cycle % max(int(opt["print_every"]), 1) == 0:
                self.log(f"cycle {cycle:4d} images {k:3d} rms(F_perp) {rms_all:.3e} max {max_all:.3e} E_HEI {self.energies[hei]:.8f}")
            if full:
                full_cycles += 1
                ci_ok = True
                if climbing:
                    ci = step_force[hei]
                    ci_ok = np.abs(ci).max() <= max_f and np.sqrt((ci ** 2).mean()) <= rms_f
                if max_all <= max_f and rms_all <= rms_f and ci_ok and (climbing or not gs["climb"] or not (0 < hei < k - 1)):
                    converged = True
                    break
                if full_cycles > int(opt["stop_in_when_full"]):
                    break
            # ---- step: steepest descent while growing, L-BFGS once fully grown (history reset on any size change)
            g = -step_force[moving].reshape(-1)
            direction = -g
            if full:
                xm = x[moving].reshape(-1)
                if self._prev is not None and len(self._prev[0]) == len(xm):
                    s_, y_ = xm - self._prev[0], g - self._prev[1]
                    if float(s_ @ y_) > 1e-12:
                        self._lbfgs_s.append(s_); self._lbfgs_y.append(y_)
                        self._lbfgs_s, self._lbfgs_y = self._lbfgs_s[-10:], self._lbfgs_y[-10:]
                self._prev = (xm.copy(), g.copy())
                q = g.copy()
                al = []
                for s_, y_ in zip(reversed(self._lbfgs_s), reversed(self._lbfgs_y)):
                    a = float(s_ @ q) / float(y_ @ s_); al.append(a); q -= a * y_
                if self._lbfgs_s:
                    q *= float(self._lbfgs_s[-1] @ self._lbfgs_y[-1]) / float(self._lbfgs_y[-1] @ self._lbfgs_y[-1])
                for (s_, y_), a in zip(zip(self._lbfgs_s, self._lbfgs_y), reversed(al)):
                    b = float(y_ @ q) / float(y_ @ s_); q += (a - b) * s_
                direction = -q
                if float(direction @ g) >= 0:                   # not a descent direction: fall back
                    direction = -g
                    self._lbfgs_s.clear(); self._lbfgs_y.clear()
            biggest = np.abs(direction).max()
            if biggest > opt["max_step"]:                         # scale_step="global"
                direction = direction * (opt["max_step"] / biggest)
            xn = x.copy()
            xn[moving] = x[moving] + direction.reshape(-1, n_dof)
            # ---- reparametrise / grow
            every = gs["reparam_every_full"] if full else gs["reparam_every"]
            self._set_coords(xn)
            stale = True
            changed = False
            if not full:
                changed = self._grow(xn, rms_img)
            if changed or (every and cycle % int(every) == 0):
                xr = self._reparametrize(self.coords)
                if climbing and 0 < hei < len(xr) - 1 and len(xr) == k:
                    xr[hei] = self.coords[hei]                    # the climbing image is not redistributed
                self._set_coords(xr)
            if changed:
                self._lbfgs_s.clear(); self._lbfgs_y.clear(); self._prev = None
                self.energies, self.forces = None, None
                need = None
            else:
                need = np.ones(len(self.coords), dtype=bool)
                need[0] = not gs["fix_first"]
                need[-1] = not gs["fix_last"]
        x = self.coords
        if self.energies is None or len(self.energies) != len(x):
            self._eval(x, np.ones(len(x), dtype=bool))
        elif stale and need is not None and len(need) == len(x):
            # max_cycles exhausted right after a step: the result must pair coordinates with THEIR energies,
            # so the stepped images get one more evaluation instead of returning pre-step energies
            self._eval(x, need)
        return GSMResult(coords=x, energies=self.energies.copy(), converged=converged, cycles=cycle, fully_grown=self.fully_grown,
                         hei_index=select_hei_index(self.energies), force_evaluations=self.n_eval, history=history)
