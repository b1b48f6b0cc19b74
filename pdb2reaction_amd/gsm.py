"""Batched Growing-String driver -- device resident since round 4.

The reference runs ``GrowingString(images, calc_getter, **GS_KW)`` + ``StringOptimizer(gs, **STOPT_KW).run()`` from
pysisyphus (reference ``path_opt.py:959-977``, ``path_search.py:664-681``); pysisyphus evaluates the images one after
another through the shared calculator.  This driver restates that loop so that ALL images needing an evaluation in a
cycle go through ONE batched call (``calc.get_forces_batch``; across GPUs one sharded call + all-gather,
``parallel.EngineStringEvaluator``).

WHERE THE LOOP RUNS.  The string (K x 3N float64 = 768 KB at 2000 atoms x 16 images), the forces, the tangents, the
projection, the L-BFGS history and the reparametrisation are torch tensors on ``device`` -- the engine's GPU when the
evaluator is a device one (``evaluate_device``), the CPU otherwise -- and a cycle crosses PCIe exactly once, for one small
vector of scalars (rms / max of the perpendicular force, per-image rms, energies, the L-BFGS curvature checks: 2K + 8
doubles) that the host needs for its DECISIONS (grow, climb, converged, stop).  The step itself is computed
optimistically on the device BEFORE that read, with the climbing state of the previous cycle; the rare events that
invalidate it (climbing switches on, a rejected curvature pair, a non-descent direction) recompute it.  At the 2-image
shard of the 8-GPU run one cycle is ~50-70 ms of GPU work: the numpy form of this loop (K x 3N doubles to the host and
back, ~40 dot products and a scipy spline per cycle) was no longer noise there.

PARITY UNPINNED: pysisyphus is not installed here, so the loop follows the published GSM semantics summarised in
SURVEY.md Appendix B -- frontier growth at ``perp_thresh``, equal-arc ("equi") reparametrisation, perpendicular-force
steps scaled to ``max_step``, climbing image once the fully grown string is below ``climb_rms``, convergence on the
``thresh`` presets of reference ``opt.py:176-187`` -- not pysisyphus' source.  Keyword names and defaults are those of
reference ``GS_KW`` / ``STOPT_KW`` (``path_opt.py:168-200``); keywords this driver does NOT implement are refused or
warned about at construction when they carry a non-default value (``_check_keywords``).  Tangents come from a parametric
cubic spline (not-a-knot) through the images (Appendix B; central differences for fewer than four images,
``tangent="central"`` forces them); with ``climb_lanczos`` (the reference default, ``path_opt.py:181-182``) the climbing
image's tangent is the lowest-curvature mode from a Lanczos iteration on finite-difference Hessian-vector products once
the string is below ``climb_lanczos_rms`` (:func:`lanczos_lowest_mode`; one single-image evaluation per Lanczos step).
Coordinates are Cartesian (no DLC).
"""
from __future__ import annotations

import warnings
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

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
LBFGS_HISTORY = 10

# Keywords of the reference's dicts that have NO effect in this driver.  A non-default value is either REFUSED (it would change what
# pysisyphus computes: coordinates, parametrisation, the step rule) or WARNED about (it only tunes pysisyphus internals that have no
# counterpart here: DLC micro-cycles, the reparametrisation trigger) -- never swallowed in silence.
_REFUSED = {
    "param": "only param='equi' (equal arc length) is implemented",
    "scheduler": "a dask scheduler is not supported: images are batched through one engine call / sharded over ranks instead",
    "align": "align=True (per-cycle Kabsch alignment of the images) is not implemented; align the endpoints beforehand (prestep.py)",
    "type": "only the string optimiser (type='string') exists here",
    "coord_type": "only Cartesian coordinates (coord_type='cart', the reference default) are implemented; DLC are not",
}
_WARNED = {
    "reparam_check": "the string is re-parametrised every `reparam_every(_full)` cycles unconditionally (no rms / norm trigger)",
    "max_micro_cycles": "there are no DLC micro-cycles in Cartesian coordinates",
    "reset_dlc": "there are no DLC to reset in Cartesian coordinates",
    "reparam_thresh": "no threshold on the re-parametrisation displacement is applied",
    "coord_diff_thresh": "no coordinate-difference convergence criterion is applied",
    "dump": "the driver writes no dump files (formats.write_trj_with_energy writes the final path)",
    "dump_restart": "the driver writes no restart files",
}


def _check_keywords(gs: Dict[str, Any], opt: Dict[str, Any], extra: Optional[Dict[str, Any]] = None) -> None:
    defaults = {**GS_KW, **STOPT_KW, "coord_type": "cart"}
    given = {**gs, **opt, **(extra or {})}
    for key, why in _REFUSED.items():
        if key in given and given[key] != defaults[key]:
            if key == "param":
                raise NotImplementedError(why)
            raise NotImplementedError(f"GrowingStringDriver: {key}={given[key]!r}: {why}")
    if given.get("scale_step", "global") not in ("global", "per_image"):
        raise NotImplementedError(f"GrowingStringDriver: scale_step={given['scale_step']!r}: 'global' (the reference default) or 'per_image'")
    for key, why in _WARNED.items():
        if key in given and given[key] != defaults[key]:
            warnings.warn(f"GrowingStringDriver: {key}={given[key]!r} has no effect here: {why}", RuntimeWarning, stacklevel=3)


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
    timing: Dict[str, float] = field(default_factory=dict)   # seconds: total wall, inside the evaluator, host share = the rest


# ---- tangents, placement, L-BFGS: torch, any device, no host synchronisation ---------------------------------------------------
def _unit(t: torch.Tensor) -> torch.Tensor:
    return t / t.norm(dim=1, keepdim=True).clamp_min(1e-30)


def tangents_central_t(x: torch.Tensor) -> torch.Tensor:
    t = torch.empty_like(x)
    t[1:-1] = x[2:] - x[:-2]
    t[0] = x[1] - x[0]
    t[-1] = x[-1] - x[-2]
    return _unit(t)


def spline_derivative_t(u: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """First derivative at the knots of the interpolating cubic spline with not-a-knot ends through (u_i, x_i), u (K,) increasing,
    x (K, D), K >= 4 -- the same spline ``scipy.interpolate.make_interp_spline(u, x, k=3)`` builds (it is unique).  The K x K
    tridiagonal system for the knot derivatives is assembled densely and solved once for all D columns."""
    k = x.shape[0]
    h = u[1:] - u[:-1]                                           # (K-1,)
    slope = (x[1:] - x[:-1]) / h[:, None]
    a = torch.zeros(k, k, dtype=x.dtype, device=x.device)
    b = torch.empty_like(x)
    i = torch.arange(1, k - 1, device=x.device)
    a[i, i] = 2.0 * (h[:-1] + h[1:])
    a[i, i + 1] = h[:-1]
    a[i, i - 1] = h[1:]
    b[1:-1] = 3.0 * (h[1:, None] * slope[:-1] + h[:-1, None] * slope[1:])
    d0 = h[0] + h[1]
    a[0, 0], a[0, 1] = h[1], d0
    b[0] = ((h[0] + 2.0 * d0) * h[1] * slope[0] + h[0] * h[0] * slope[1]) / d0
    d1 = h[-1] + h[-2]
    a[-1, -1], a[-1, -2] = h[-2], d1
    b[-1] = (h[-1] * h[-1] * slope[-2] + (2.0 * d1 + h[-1]) * h[-2] * slope[-1]) / d1
    return torch.linalg.solve_ex(a, b, check_errors=False)[0]


def tangents_t(x: torch.Tensor, kind: str = "spline") -> torch.Tensor:
    """Unit tangents at the images of a (K, D) string.  "spline": derivative of the interpolating parametric cubic spline
    (not-a-knot) over the cumulative chord length; needs >= 4 images with distinct positions, otherwise
    (and for "central") central differences with one-sided ends.  The degenerate case is selected ON THE DEVICE."""
    if kind == "spline" and x.shape[0] >= 4:
        seg = (x[1:] - x[:-1]).norm(dim=1)
        ok = (seg > 1e-12).all()
        segc = torch.where(ok, seg, torch.ones_like(seg))        # keep the solve finite when it is not going to be used
        u = torch.cat([seg.new_zeros(1), segc.cumsum(0)])
        ts = _unit(spline_derivative_t(u, x))
        return torch.where(ok, ts, tangents_central_t(x))
    return tangents_central_t(x)


def place_t(x: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """Points at normalised arc-length positions `targets` (in [0,1]) along the polyline through x."""
    k = x.shape[0]
    seg = (x[1:] - x[:-1]).norm(dim=1)
    s = torch.cat([seg.new_zeros(1), seg.cumsum(0)])
    total = torch.where(s[-1] > 0, s[-1], torch.ones_like(s[-1]))
    uu = targets * total
    idx = torch.searchsorted(s, uu.contiguous(), right=True).clamp(1, k - 1)
    s0, s1 = s[idx - 1], s[idx]
    w = ((uu - s0) / (s1 - s0).clamp_min(1e-30)).unsqueeze(1)
    return x[idx - 1] * (1.0 - w) + x[idx] * w


def _gram(a: torch.Tensor, b: torch.Tensor, chunk: int = 2048) -> torch.Tensor:
    """a @ b.T for two skinny (m, n) matrices, m ~ 10, n ~ 1e5: as ONE GEMM this is a 10 x 10 output with a 96 000-long reduction,
    which the BLAS runs on a single workgroup (measured 15 ms on the MI355X in float64); cut into n / chunk independent products
    (a batched GEMM over all CUs) and summed, it is ~0.05 ms."""
    m, n = a.shape
    pad = (-n) % chunk
    if pad:
        a = torch.nn.functional.pad(a, (0, pad))
        b = torch.nn.functional.pad(b, (0, pad))
    nb = a.shape[1] // chunk
    return torch.bmm(a.reshape(m, nb, chunk).transpose(0, 1), b.reshape(b.shape[0], nb, chunk).permute(1, 2, 0)).sum(0)


def lbfgs_direction_t(s_hist: torch.Tensor, y_hist: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """-H g with H the L-BFGS inverse Hessian of the (m, n) history (rows oldest -> newest, all pairs with s.y > 0), in the
    COMPACT form (Byrd, Nocedal, Schnabel 1994): H = gamma I + [S gamma Y] [[R^-T (D + gamma Y^T Y) R^-1, -R^-T], [-R^-1, 0]]
    [S^T; gamma Y^T], R = triu(S^T Y), D = diag(S^T Y), gamma = s.y / y.y of the newest pair -- the same vector the two-loop
    recursion gives, in ~10 launches instead of ~12 per pair."""
    m = s_hist.shape[0]
    if m == 0:
        return -g
    sy = _gram(s_hist, y_hist)                                   # (m, m): sy[i, j] = s_i . y_j
    yy = _gram(y_hist, y_hist)
    gamma = sy[-1, -1] / yy[-1, -1]
    r = torch.triu(sy)
    a = s_hist @ g
    b = gamma * (y_hist @ g)
    u = torch.linalg.solve_triangular(r, a[:, None], upper=True)[:, 0]
    w = (torch.diag(torch.diagonal(sy)) + gamma * yy) @ u - b
    p1 = torch.linalg.solve_triangular(r.T, w[:, None], upper=False)[:, 0]
    return -(gamma * g + s_hist.T @ p1 - gamma * (y_hist.T @ u))


def lbfgs_two_loop_t(s_hist: torch.Tensor, y_hist: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """Reference form of :func:`lbfgs_direction_t` (tests): the textbook two-loop recursion."""
    q = g.clone()
    al = []
    for s_, y_ in zip(reversed(list(s_hist)), reversed(list(y_hist))):
        a = (s_ @ q) / (y_ @ s_); al.append(a); q = q - a * y_
    if len(s_hist):
        q = q * ((s_hist[-1] @ y_hist[-1]) / (y_hist[-1] @ y_hist[-1]))
    for (s_, y_), a in zip(zip(s_hist, y_hist), reversed(al)):
        b = (y_ @ q) / (y_ @ s_); q = q + (a - b) * s_
    return -q


def hei_index_t(e: torch.Tensor) -> torch.Tensor:
    """``select_hei_index`` (reference ``path_opt.py:259-273``) on the device: 0-dim long tensor."""
    n = e.shape[0]
    if n < 3:
        return torch.argmax(e)
    inner = e[1:-1]
    peak = (inner > e[:-2]) & (inner > e[2:])
    masked = torch.where(peak, inner, torch.full_like(inner, float("-inf")))
    return 1 + torch.where(peak.any(), torch.argmax(masked), torch.argmax(inner))


# ---- numpy faces of the same functions (tests, tools) --------------------------------------------------------------------------
def _tangents(x: np.ndarray, kind: str = "spline") -> np.ndarray:
    return tangents_t(torch.as_tensor(np.asarray(x, dtype=np.float64)), kind).numpy()


def _place(x: np.ndarray, targets: np.ndarray) -> np.ndarray:
    return place_t(torch.as_tensor(np.asarray(x, dtype=np.float64)), torch.as_tensor(np.asarray(targets, dtype=np.float64))).numpy()


def lanczos_lowest_mode_t(grad_fn: Callable[[torch.Tensor], torch.Tensor], x: torch.Tensor, g0: torch.Tensor, guess: torch.Tensor, *,
                          dx: float = 5e-3, dl: float = 1e-2, max_cycles: int = 25, orient: Optional[torch.Tensor] = None) -> Tuple[float, torch.Tensor, int]:
    """Lowest Hessian eigenpair at x from a Lanczos recursion on forward-difference Hessian-vector products
    H q ~ (g(x + dx q) - g(x)) / dx; one gradient per step, started from `guess` (the string tangent).  Vectors stay on x's device;
    the recursion coefficients (two scalars per step) come to the host, where the small tridiagonal eigenproblem is solved --
    every step is a force evaluation anyway.

    Returns (eigenvalue, unit eigenvector, gradient evaluations).  Stops when the lowest Ritz value changes by less than
    `dl` (relative) between two steps, when the Krylov space is exhausted, or after `max_cycles` steps.  The vector is
    oriented along `orient` (default: `guess`) -- a warm start passes last cycle's mode as `guess` and the string tangent as `orient`."""
    n = x.numel()
    x = x.reshape(-1)
    guess = guess.reshape(-1).to(x.dtype)
    r = guess.clone()
    beta = float(r.norm())
    if beta < 1e-14:
        raise ValueError("lanczos: zero start vector")
    qs: List[torch.Tensor] = []
    alphas: List[float] = []
    betas: List[float] = []
    q_prev = torch.zeros_like(x)
    w_prev: Optional[float] = None
    w_min, v_min = 0.0, r / beta
    steps = 0
    for steps in range(1, min(int(max_cycles), n) + 1):
        q = r / beta
        if qs:                                          # full re-orthogonalisation: the space is small and FD noise is not
            qm = torch.stack(qs)
            for _ in range(1):
                q = q - qm.T @ (qm @ q)
        q = q / q.norm().clamp_min(1e-30)
        u = (grad_fn(x + dx * q).reshape(-1) - g0.reshape(-1)) / dx
        if qs:
            u = u - beta * q_prev
        alpha = float(q @ u)
        r = u - alpha * q
        qs.append(q); alphas.append(alpha)
        t = np.diag(alphas) + np.diag(betas, 1) + np.diag(betas, -1)
        w, v = np.linalg.eigh(t)
        w_min = float(w[0])
        v_min = torch.stack(qs, dim=1) @ torch.as_tensor(v[:, 0], dtype=x.dtype, device=x.device)
        beta = float(r.norm())
        if w_prev is not None and abs(w_min - w_prev) <= dl * max(abs(w_prev), 1e-12):
            break
        if beta < 1e-10:
            break
        w_prev, q_prev = w_min, q
        betas.append(beta)
    v_min = v_min / v_min.norm().clamp_min(1e-30)
    if float(v_min @ (guess if orient is None else orient.reshape(-1).to(x.dtype))) < 0.0:
        v_min = -v_min
    return w_min, v_min, steps


def lanczos_lowest_mode(grad_fn: Callable[[np.ndarray], np.ndarray], x: np.ndarray, g0: np.ndarray, guess: np.ndarray, *,
                        dx: float = 5e-3, dl: float = 1e-2, max_cycles: int = 25):
    """numpy face of :func:`lanczos_lowest_mode_t`."""
    def gt(xq: torch.Tensor) -> torch.Tensor:
        return torch.as_tensor(np.asarray(grad_fn(xq.numpy()), dtype=np.float64))
    w, v, n = lanczos_lowest_mode_t(gt, torch.as_tensor(np.asarray(x, dtype=np.float64)), torch.as_tensor(np.asarray(g0, dtype=np.float64)),
                                    torch.as_tensor(np.asarray(guess, dtype=np.float64)), dx=dx, dl=dl, max_cycles=max_cycles)
    return w, v.numpy(), n


class GrowingStringDriver:
    """Growing string between two endpoints; every cycle issues one batched E+F call.

    calc: object with ``get_forces_batch(atoms, coords[K,3N] Bohr) -> {"energy": (K,), "forces": (K,3N)}``
    (``pdb2reaction_amd.uma_pysis.uma_pysis`` or any stand-in; numpy in, numpy out).  ``evaluate`` overrides how a batch is evaluated
    with the same numpy contract ``evaluate(x[k,3N]) -> (E[k], F[k,3N])``.  ``evaluate_device`` is the device contract:
    ``evaluate_device(x: torch[k,3N] on `device`) -> (E: torch[k], F: torch[k,3N])`` on the same device, nothing crossing PCIe
    (``parallel.EngineStringEvaluator``: the engine's device-pointer entry, sharded over the ranks of a process group).
    """

    def __init__(self, atoms: Sequence[str], reactant: np.ndarray, product: np.ndarray, calc: Any = None,
                 evaluate: Optional[Callable[[np.ndarray], Any]] = None, gs_kw: Optional[Dict[str, Any]] = None,
                 stopt_kw: Optional[Dict[str, Any]] = None, log: Optional[Callable[[str], None]] = None,
                 evaluate_device: Optional[Callable[[torch.Tensor], Any]] = None, device: Optional[torch.device] = None,
                 geom_kw: Optional[Dict[str, Any]] = None, images: Optional[np.ndarray] = None):
        """images: (max_nodes + 2, 3N) Bohr -- start from this FULLY GROWN string (a restart, or a path from another method) instead of
        growing one from the endpoints; its first / last rows must be the reactant / product."""
        self.atoms = list(atoms)
        self.gs = {**GS_KW, **(gs_kw or {})}
        self.opt = {**STOPT_KW, **(stopt_kw or {})}
        _check_keywords(gs_kw or {}, stopt_kw or {}, {k: v for k, v in (geom_kw or {}).items() if k == "coord_type"})
        r = np.asarray(reactant, dtype=np.float64).reshape(-1)
        p = np.asarray(product, dtype=np.float64).reshape(-1)
        if r.shape != p.shape or r.size != 3 * len(self.atoms):
            raise ValueError("reactant/product must both be (3N,) for the given atoms")
        self.calc, self._evaluate, self._evaluate_device, self.log = calc, evaluate, evaluate_device, (log or (lambda s: None))
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        if evaluate_device is None and self.device.type != "cpu":
            raise ValueError("a non-CPU `device` needs `evaluate_device` (a numpy evaluator would pull the string to the host every cycle)")
        self.max_images = int(self.gs["max_nodes"]) + 2
        if hasattr(calc, "reserve_images"):                  # the string grows to max_images: one workspace allocation instead of one per growth
            calc.reserve_images(self.max_images)
        step = 1.0 / (self.max_images - 1)
        if self.max_images <= 3:
            left, right = [r], [p]
            if self.max_images == 3:
                left.append(0.5 * (r + p))
        else:                                                # endpoints + first frontier node on either side
            left = [r, r + step * (p - r)]
            right = [p - step * (p - r), p]
        if images is not None:
            im = np.asarray(images, dtype=np.float64).reshape(len(images), -1)
            if im.shape != (self.max_images, r.size) or not (np.array_equal(im[0], r) and np.array_equal(im[-1], p)):
                raise ValueError(f"images must be ({self.max_images}, {r.size}) with the reactant / product as first / last row")
            left, right = list(im[: self.max_images // 2]), list(im[self.max_images // 2:])
        self._x = torch.as_tensor(np.stack(left + right), dtype=torch.float64, device=self.device)
        self._nl = len(left)
        self._e: Optional[torch.Tensor] = None
        self._f: Optional[torch.Tensor] = None
        self.n_eval = 0
        self.lanczos_evals = 0
        self.fixed_climb_index: Optional[int] = None         # climb_fixed=True: the image that started to climb (set when climbing starts)
        self.lanczos_calls = 0                               # Lanczos recursions run (one per climbing cycle below climb_lanczos_rms)
        self.lanczos_warm_calls = 0                          # ... of which started from the previous cycle's mode (and were kept)
        self.lanczos_warm_rejected = 0                       # warm recursions whose result failed the guard (a cold one followed)
        self.lanczos_log: List[Tuple[float, float, int, bool]] = []      # per recursion kept: (lowest Ritz value, overlap with the tangent, gradients, warm)
        # (HEI index, images, unit vector) of the most recent trusted Lanczos mode: the next cycle's recursion starts from it (see run())
        self._lanczos_prev: Optional[Tuple[int, int, torch.Tensor]] = None
        self.tangent_kind = str(self.gs.get("tangent", "spline"))
        self._hist_s: Optional[torch.Tensor] = None          # (m, n) rows oldest -> newest
        self._hist_y: Optional[torch.Tensor] = None
        self._prev: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
        self.t_eval = 0.0
        self.redo_steps = 0                                  # how often the optimistic step had to be recomputed

    @classmethod
    def from_calculator(cls, atoms: Sequence[str], reactant: np.ndarray, product: np.ndarray, calc: Any, *, group=None, **kw) -> "GrowingStringDriver":
        """The driver for a ``pdb2reaction_amd.uma_pysis`` calculator, DEVICE RESIDENT when the calculator runs on the HIP engine: the string
        lives on the engine's GPU and is evaluated through ``parallel.EngineStringEvaluator`` (device-pointer entry, frozen rows zeroed,
        images sharded over the ranks of `group` when torch.distributed is initialised) -- per image exactly what ``calc.get_forces``
        returns.  Any other calculator (a stand-in, a core without an engine, the graph-parallel mode) gets the numpy evaluator."""
        core = calc._ensure_core(atoms) if hasattr(calc, "_ensure_core") else None
        engine = getattr(core, "engine", None)
        if engine is None or getattr(core, "_gp", None) is not None or not hasattr(engine, "energy_forces_dev"):
            return cls(atoms, reactant, product, calc, **kw)
        from .parallel import EngineStringEvaluator

        max_images = int({**GS_KW, **(kw.get("gs_kw") or {})}["max_nodes"]) + 2
        ev = EngineStringEvaluator(engine, len(list(atoms)), core.device, frozen=getattr(calc, "freeze_atoms", ()), group=group, max_images=max_images)
        drv = cls(atoms, reactant, product, calc=None, evaluate_device=ev, device=core.device, **kw)
        drv.calc = calc
        return drv

    # ---- helpers -------------------------------------------------------------------------------------
    @property
    def coords(self) -> np.ndarray:
        return self._x.detach().cpu().numpy().copy()

    @property
    def energies(self) -> Optional[np.ndarray]:
        return None if self._e is None else self._e.detach().cpu().numpy().copy()

    @property
    def forces(self) -> Optional[np.ndarray]:
        return None if self._f is None else self._f.detach().cpu().numpy().copy()

    @property
    def n_images(self) -> int:
        return int(self._x.shape[0])

    @property
    def fully_grown(self) -> bool:
        return self.n_images >= self.max_images

    def _raw_eval(self, xq: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        import time
        t0 = time.perf_counter()
        k = xq.shape[0]
        if self._evaluate_device is not None:
            e, f = self._evaluate_device(xq)
            e = e.to(device=self.device, dtype=torch.float64)
            f = f.to(device=self.device, dtype=torch.float64).reshape(k, -1)
        else:
            xn = xq.detach().cpu().numpy()
            if self._evaluate is not None:
                e, f = self._evaluate(xn)
            else:
                res = self.calc.get_forces_batch(self.atoms, xn)
                e, f = res["energy"], res["forces"]
            e = torch.as_tensor(np.asarray(e, dtype=np.float64), device=self.device)
            f = torch.as_tensor(np.asarray(f, dtype=np.float64).reshape(k, -1), device=self.device)
        self.n_eval += k
        self.t_eval += time.perf_counter() - t0
        return e, f

    def _eval(self, idx: Sequence[int]):
        """Batched E+F for the images `idx` (host-known list)."""
        k = self.n_images
        if self._e is None or self._e.shape[0] != k:
            self._e = torch.zeros(k, dtype=torch.float64, device=self.device)
            self._f = torch.zeros_like(self._x)
        if len(idx) == 0:
            return
        if len(idx) == k:
            self._e, self._f = self._raw_eval(self._x)
            return
        it = torch.as_tensor(list(idx), dtype=torch.long, device=self.device)
        e, f = self._raw_eval(self._x.index_select(0, it))
        self._e = self._e.index_copy(0, it, e)
        self._f = self._f.index_copy(0, it, f)

    def _single_forces(self, xq: torch.Tensor) -> torch.Tensor:
        """Forces of ONE geometry through the same (batched) evaluator -- the serial steps of the Lanczos recursion."""
        return self._raw_eval(xq.reshape(1, -1))[1].reshape(-1)

    def _targets(self, k: int, nl: int) -> torch.Tensor:
        if k >= self.max_images:
            tg = np.linspace(0.0, 1.0, k)
        else:                                                # growing: left nodes at i*step, right nodes mirrored from the end
            step = 1.0 / (self.max_images - 1)
            tg = np.concatenate([np.arange(nl) * step, 1.0 - np.arange(k - nl - 1, -1, -1) * step])
        return torch.as_tensor(tg, dtype=torch.float64, device=self.device)

    def _reparametrize(self, x: torch.Tensor, nl: int) -> torch.Tensor:
        out = place_t(x, self._targets(x.shape[0], nl))
        out[0], out[-1] = x[0], x[-1]
        return out

    def _grow(self, rms_img: np.ndarray) -> bool:
        """Add a node next to a frontier whose perpendicular force is below perp_thresh; True if the string changed."""
        grew = False
        nl0 = self._nl
        for side in ("left", "right"):
            if self.fully_grown:
                break
            x, nl = self._x, self._nl
            fl, fr = x[nl - 1], x[nl]
            gap_nodes = self.max_images - x.shape[0]
            frontier = nl0 - 1 if side == "left" else nl0
            if rms_img[frontier] >= self.gs["perp_thresh"]:
                continue
            d = (fr - fl) / (gap_nodes + 1)
            new = (fl + d) if side == "left" else (fr - d)
            self._x = torch.cat([x[:nl], new[None], x[nl:]], dim=0)
            if side == "left":
                self._nl += 1
            grew = True
        return grew

    def _reset_history(self):
        self._hist_s = self._hist_y = None
        self._prev = None
        self._lanczos_prev = None

    # ---- one optimistic step on the device -----------------------------------------------------------
    def _device_cycle(self, moving_idx: torch.Tensor, moving_mask: torch.Tensor, climbing: bool, full: bool,
                      lanczos_t: Optional[torch.Tensor], offer_pair: bool, hei_host: Optional[int]):
        """Projection, statistics and the step of one cycle, all on the device; nothing is committed here.  Returns
        (stats vector [rms_all, max_all, ci_max, ci_rms, s.y, direction.g, hei, max|step|, rms_img (K), E (K)], new coordinates of the
        moving images, the curvature pair that would be appended or None, the (xm, g) to remember, the tangents)."""
        x, f, e = self._x, self._f, self._e
        k, n_dof = x.shape
        zero = torch.zeros((), dtype=x.dtype, device=x.device)
        nan = torch.full((), float("nan"), dtype=x.dtype, device=x.device)
        t = tangents_t(x, self.tangent_kind)
        fpar = (f * t).sum(dim=1, keepdim=True) * t
        fperp = (f - fpar) * moving_mask[:, None]
        rms_img = torch.sqrt((fperp ** 2).sum(1) / n_dof)
        fm = fperp.index_select(0, moving_idx)
        rms_all, max_all = (torch.sqrt((fm ** 2).mean()), fm.abs().max()) if fm.numel() else (zero, zero)
        hei = hei_index_t(e) if hei_host is None else torch.full((), int(hei_host), dtype=torch.long, device=x.device)
        step_force = fperp
        ci_max = ci_rms = zero
        if climbing:
            # the climbing image's force: component along its tangent inverted.  Applied through a row selector so that `hei` need not
            # be known on the host; an end image (hei == 0 or k-1) never climbs.
            inner = ((hei > 0) & (hei < k - 1)).to(x.dtype)
            t_ci = t.index_select(0, hei[None])[0] if lanczos_t is None else lanczos_t
            f_h = f.index_select(0, hei[None])[0]
            ci = f_h - 2.0 * (f_h @ t_ci) * t_ci
            sel = torch.zeros(k, dtype=x.dtype, device=x.device).index_fill(0, hei[None], 1.0) * inner
            step_force = fperp * (1.0 - sel[:, None]) + sel[:, None] * ci[None, :]
            ci_max, ci_rms = ci.abs().max(), torch.sqrt((ci ** 2).mean())
        # ---- step: steepest descent while growing, L-BFGS once fully grown (history reset on any size change)
        g = -step_force.index_select(0, moving_idx).reshape(-1)
        xm = x.index_select(0, moving_idx).reshape(-1)
        sy = dg = nan
        pair = None
        direction = -g
        if full and g.numel():
            hs, hy = self._hist_s, self._hist_y
            if offer_pair and self._prev is not None and self._prev[0].numel() == xm.numel():
                s_, y_ = xm - self._prev[0], g - self._prev[1]
                sy = s_ @ y_
                pair = (s_, y_)
                hs = s_[None] if hs is None else torch.cat([hs, s_[None]])[-LBFGS_HISTORY:]
                hy = y_[None] if hy is None else torch.cat([hy, y_[None]])[-LBFGS_HISTORY:]
            if hs is not None:
                direction = lbfgs_direction_t(hs, hy, g)
                dg = direction @ g
        if g.numel():
            biggest = direction.abs().max()
            if self.opt.get("scale_step", "global") == "per_image":
                # every image's step scaled on its own: only the images whose largest component exceeds max_step are shortened
                rows = direction.reshape(-1, n_dof)
                big = rows.abs().amax(dim=1, keepdim=True)
                direction = (rows * torch.clamp(self.opt["max_step"] / big.clamp_min(1e-300), max=1.0)).reshape(-1)
            else:                                                                                              # scale_step="global"
                direction = direction * torch.clamp(self.opt["max_step"] / biggest.clamp_min(1e-300), max=1.0)
            smax = direction.abs().max()
        else:
            smax = zero
        xn_moving = (xm + direction).reshape(-1, n_dof)
        stats = torch.cat([torch.stack([rms_all, max_all, ci_max, ci_rms, sy, dg, hei.to(x.dtype), smax]), rms_img, e])
        return stats, xn_moving, pair, (xm, g), t

    # ---- main loop -----------------------------------------------------------------------------------
    @with_small_host_math
    def run(self) -> GSMResult:
        import time

        t_run = time.perf_counter()
        self.t_eval = 0.0
        gs, opt = self.gs, self.opt
        max_f, rms_f, _, _ = THRESH[opt["thresh"]] if isinstance(opt["thresh"], str) else opt["thresh"]
        history: List[Dict[str, float]] = []
        converged = False
        full_cycles = 0
        need: Optional[List[int]] = None
        climbing = False
        fixed_hei: Optional[int] = None                 # climb_fixed=True: the image that started to climb keeps climbing
        cycle = 0
        stale = False
        dev = self.device

        def movers(kk: int) -> List[int]:
            return [i for i in range(kk) if not ((i == 0 and gs["fix_first"]) or (i == kk - 1 and gs["fix_last"]))]

        for cycle in range(1, int(opt["max_cycles"]) + 1):
            k = self.n_images
            if need is None:
                need = list(range(k))
            self._eval(need)
            stale = False                                       # energies/forces belong to the current coordinates
            moving = movers(k)
            moving_idx = torch.as_tensor(moving, dtype=torch.long, device=dev)
            moving_mask = torch.zeros(k, dtype=torch.float64, device=dev).index_fill(0, moving_idx, 1.0)
            full = self.fully_grown
            # Optimistic device cycle with the climbing state as it stands, then ONE read of 2K + 8 doubles for the host's decisions.
            # A decision that invalidates the step (climbing switches on, Lanczos tangent, rejected curvature pair, non-descent
            # direction, an HEI tie broken differently on the host) recomputes it -- rare, counted in `redo_steps`.
            lanczos_t: Optional[torch.Tensor] = None
            offer_pair, hei_host = True, fixed_hei
            rms_all = max_all = 0.0
            hei, rms_img, energies = 0, None, None
            for attempt in range(5):
                stats_t, xn_moving, pair, prev_new, t_dev = self._device_cycle(moving_idx, moving_mask, climbing, full, lanczos_t, offer_pair, hei_host)
                stats = stats_t.cpu().numpy()
                ci_max, ci_rms, sy, dg = float(stats[2]), float(stats[3]), float(stats[4]), float(stats[5])
                again = False
                if attempt == 0:
                    rms_all, max_all = float(stats[0]), float(stats[1])
                    rms_img, energies = stats[8:8 + k].copy(), stats[8 + k:8 + 2 * k].copy()
                    hei = select_hei_index(energies) if fixed_hei is None else fixed_hei
                    if hei != int(stats[6]):
                        hei_host, again = hei, True
                    if full and gs["climb"] and not climbing and rms_all <= gs["climb_rms"] and 0 < hei < k - 1:
                        climbing = True
                        if gs.get("climb_fixed", False):
                            fixed_hei = self.fixed_climb_index = hei      # determined once, when climbing starts (reference GS_KW climb_fixed)
                        self._reset_history()
                        again = True
                    if climbing and 0 < hei < k - 1 and gs["climb_lanczos"] and rms_all <= gs["climb_lanczos_rms"]:
                        # lowest-curvature direction at the HEI instead of the string tangent (reference GS_KW climb_lanczos)
                        # Start vector.  Cold: the string tangent (what a first recursion has).  Warm (gs_kw["climb_lanczos_warm_start"], default
                        # True): last cycle's mode -- the HEI moves by at most max_step per cycle, so it is an almost converged start vector and
                        # the recursion (same dl, same max_cycles, same Ritz-value stop rule) needs its minimum of two gradients instead of
                        # 3-25: the SERIAL single-image depth of the reference's default climbing phase (path_opt.py:179-182; measured on the c3
                        # string: 11.6 -> 4.0 gradients per cycle).  Guarded: a recursion started from a near-eigenvector spans a tiny Krylov
                        # space, meets the stop rule at ANY eigenvector and would follow one that has stopped being the lowest.  So (i) a warm
                        # result is kept only while its curvature is negative (and, optionally, while it overlaps the tangent by at least
                        # `climb_lanczos_warm_overlap`, default off: the lowest mode need not lie along the path -- on the synthetic c3 string it
                        # is orthogonal to it for the cold recursion too); otherwise this cycle pays for a cold recursion as well; (ii) every
                        # `climb_lanczos_refresh`-th recursion (10) is a cold one, which sees every direction again.
                        t_hei = t_dev[hei] / t_dev[hei].norm().clamp_min(1e-30)
                        lp = self._lanczos_prev
                        grad_fn = lambda xq: -self._single_forces(xq)                      # noqa: E731
                        warm = bool(gs.get("climb_lanczos_warm_start", True)) and lp is not None and lp[0] == hei and lp[1] == k
                        guard = bool(gs.get("climb_lanczos_warm_guard", True))          # False: measurement only (bench.py) -- any warm result is kept
                        min_ov = float(gs.get("climb_lanczos_warm_overlap", -1.0))
                        refresh = max(int(gs.get("climb_lanczos_refresh", 10)), 1)
                        warm = warm and (self.lanczos_calls % refresh != refresh - 1)
                        trusted = lambda w_, v_: (not guard) or (w_ < 0.0 and float(v_ @ t_hei) >= min_ov)     # noqa: E731
                        done = False
                        if warm:
                            w_l, lanczos_t, n_l = lanczos_lowest_mode_t(grad_fn, self._x[hei], -self._f[hei], lp[2], orient=t_hei)
                            self.lanczos_evals += n_l
                            done = trusted(w_l, lanczos_t)
                            self.lanczos_warm_calls += int(done)
                            self.lanczos_warm_rejected += int(not done)
                        if not done:
                            w_l, lanczos_t, n_l = lanczos_lowest_mode_t(grad_fn, self._x[hei], -self._f[hei], t_hei)
                            self.lanczos_evals += n_l
                        # only a mode that can be trusted next cycle is remembered
                        self._lanczos_prev = (hei, k, lanczos_t) if trusted(w_l, lanczos_t) else None
                        self.lanczos_calls += 1
                        self.lanczos_log.append((float(w_l), float(lanczos_t @ t_hei), int(n_l), bool(done)))
                        again = True
                if not again and pair is not None and not (sy > 1e-12):
                    offer_pair, again = False, True                # curvature pair rejected: rebuild the direction without it
                if not again and np.isfinite(dg) and dg >= 0.0:
                    self._hist_s = self._hist_y = None            # not a descent direction: steepest descent, history dropped
                    offer_pair, again = False, True
                if not again:
                    break
                self.redo_steps += 1
            history.append({"cycle": cycle, "images": k, "rms_fperp": rms_all, "max_fperp": max_all, "e_hei": float(energies[hei]),
                            "climbing": float(climbing)})
            if cycle % max(int(opt["print_every"]), 1) == 0:
                self.log(f"cycle {cycle:4d} images {k:3d} rms(F_perp) {rms_all:.3e} max {max_all:.3e} E_HEI {energies[hei]:.8f}")
            if full:
                full_cycles += 1
                ci_ok = True
                if climbing and 0 < hei < k - 1:
                    ci_ok = ci_max <= max_f and ci_rms <= rms_f
                if max_all <= max_f and rms_all <= rms_f and ci_ok and (climbing or not gs["climb"] or not (0 < hei < k - 1)):
                    converged = True
                    break
                if full_cycles > int(opt["stop_in_when_full"]):
                    break
            # ---- commit the step
            if full:
                if pair is not None:
                    self._hist_s = pair[0][None] if self._hist_s is None else torch.cat([self._hist_s, pair[0][None]])[-LBFGS_HISTORY:]
                    self._hist_y = pair[1][None] if self._hist_y is None else torch.cat([self._hist_y, pair[1][None]])[-LBFGS_HISTORY:]
                self._prev = prev_new
            self._x = self._x.index_copy(0, moving_idx, xn_moving)
            stale = True
            # ---- reparametrise / grow
            every = gs["reparam_every_full"] if full else gs["reparam_every"]
            changed = False
            if not full:
                changed = self._grow(rms_img)
            if changed or (every and cycle % int(every) == 0):
                x_before = self._x
                xr = self._reparametrize(x_before, self._nl)
                if climbing and 0 < hei < xr.shape[0] - 1 and xr.shape[0] == k:
                    xr[hei] = x_before[hei]                       # the climbing image is not redistributed
                self._x = xr
            if changed:
                self._reset_history()
                self._e, self._f = None, None
                need = None
            else:
                need = movers(self.n_images)
        k = self.n_images
        if self._e is None or self._e.shape[0] != k:
            self._eval(list(range(k)))
        elif stale and need is not None:
            # max_cycles exhausted right after a step: the result must pair coordinates with THEIR energies,
            # so the stepped images get one more evaluation instead of returning pre-step energies
            self._eval(need)
        e_out = self.energies
        t_total = time.perf_counter() - t_run
        return GSMResult(coords=self.coords, energies=e_out, converged=converged, cycles=cycle, fully_grown=self.fully_grown,
                         hei_index=select_hei_index(e_out), force_evaluations=self.n_eval, history=history,
                         timing={"total_s": t_total, "evaluator_s": self.t_eval, "host_s": t_total - self.t_eval,
                                 "redo_steps": float(self.redo_steps), "lanczos_evals": float(self.lanczos_evals),
                                 "lanczos_calls": float(self.lanczos_calls), "lanczos_warm_calls": float(self.lanczos_warm_calls),
                                 "lanczos_warm_rejected": float(self.lanczos_warm_rejected)})
