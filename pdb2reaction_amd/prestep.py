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
from ._host import with_small_host_math
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


# ======================================================================================================================
# Rigid alignment with the reference's anchor special cases, and the staged "move anchors + relax" scan
# (reference align_freeze_atoms.py:245-386 and :390-517).  Array API: coordinates (N, 3) in Bohr, `anchors` = the union
# of the pair's freeze_atoms.  The scan relaxes through `lbfgs.BatchedLBFGS`, so several mobile images advance together.
# ======================================================================================================================
from ._calculator_base import BOHR2ANG  # noqa: E402


def _rodrigues(axis: np.ndarray, theta: float) -> np.ndarray:
    """Rotation matrix (column-vector convention) by `theta` about `axis` (identity for a null axis)."""
    u = np.asarray(axis, dtype=float)
    nrm = np.linalg.norm(u)
    if nrm < 1e-16:
        return np.eye(3)
    x, y, z = u / nrm
    k = np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]])
    return np.eye(3) + np.sin(theta) * k + (1.0 - np.cos(theta)) * (k @ k)


def _rotation_a_to_b(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Column-vector rotation taking direction a to direction b (any perpendicular axis for the antiparallel case)."""
    a = np.asarray(a, dtype=float); b = np.asarray(b, dtype=float)
    na, nb = np.linalg.norm(a), np.linalg.norm(b)
    if na < 1e-16 or nb < 1e-16:
        return np.eye(3)
    a, b = a / na, b / nb
    v = np.cross(a, b)
    s, c = np.linalg.norm(v), float(np.clip(a @ b, -1.0, 1.0))
    if s < 1e-12:
        if c > 0.0:
            return np.eye(3)
        helper = np.array([1.0, 0.0, 0.0]) if abs(a[0]) <= 0.9 else np.array([0.0, 1.0, 0.0])
        ax = np.cross(a, helper)
        return _rodrigues(ax / (np.linalg.norm(ax) + 1e-16), np.pi)
    return _rodrigues(v / s, np.arctan2(s, c))


def rmsd_ang(a_bohr: np.ndarray, b_bohr: np.ndarray) -> float:
    a = np.asarray(a_bohr, dtype=float).reshape(-1, 3) * BOHR2ANG
    b = np.asarray(b_bohr, dtype=float).reshape(-1, 3) * BOHR2ANG
    return float(np.sqrt(np.mean(np.sum((a - b) ** 2, axis=1)))) if len(a) else float("nan")


def freeze_union(freeze_ref, freeze_mob, n_atoms: Optional[int] = None) -> List[int]:
    """Sorted union of two freeze lists, out-of-range indices dropped (reference ``_freeze_union``, ``align_freeze_atoms.py:253-268``)."""
    cand = sorted({int(i) for i in list(freeze_ref if freeze_ref is not None else []) + list(freeze_mob if freeze_mob is not None else [])})
    return cand if n_atoms is None else [i for i in cand if 0 <= i < int(n_atoms)]


def align_second_to_first(ref_bohr: np.ndarray, mob_bohr: np.ndarray, anchors: Sequence[int] = ()):
    """Rigid fit of `mob` onto `ref`; returns (aligned coordinates, {before_A, after_A, n_used, mode}).

    1 anchor: translate it onto its partner, then the best rotation ABOUT that point (all-atom RMSD);
    2 anchors: match midpoints, align the anchor axis, then the best rotation about that axis (all-atom RMSD), falling
    back to Kabsch when the axis is degenerate; otherwise Kabsch on the anchors (all atoms when there are none), RMSD
    reported on the fitted selection (reference ``align_second_to_first_kabsch_inplace``, ``align_freeze_atoms.py:271-387``).
    """
    p = np.asarray(ref_bohr, dtype=float).reshape(-1, 3)
    q = np.asarray(mob_bohr, dtype=float).reshape(-1, 3)
    if p.shape != q.shape:
        raise ValueError(f"Different atom counts: {p.shape[0]} vs {q.shape[0]}")
    n = p.shape[0]
    idx = [int(i) for i in anchors if 0 <= int(i) < n]
    report_all = False
    if len(idx) == 1:
        i = idx[0]
        p0 = p[i].copy()
        q_rel, p_rel = q + (p0 - q[i]) - p0, p - p0
        u, _, vt = np.linalg.svd(p_rel.T @ q_rel)
        rot = vt.T @ u.T
        if np.linalg.det(rot) < 0.0:
            vt = vt.copy(); vt[-1] *= -1.0
            rot = vt.T @ u.T
        out = q_rel @ rot + p0
        return out, {"before_A": rmsd_ang(p, q), "after_A": rmsd_ang(p, out), "n_used": 1, "mode": "one_anchor"}
    if len(idx) == 2:
        i0, i1 = idx
        v_p, v_q = p[i1] - p[i0], q[i1] - q[i0]
        if np.linalg.norm(v_p) > 1e-16 and np.linalg.norm(v_q) > 1e-16:
            pm, qm = 0.5 * (p[i0] + p[i1]), 0.5 * (q[i0] + q[i1])
            q0 = ((q + (pm - qm)) - pm) @ _rotation_a_to_b(v_q, v_p).T + pm
            u = v_p / (np.linalg.norm(v_p) + 1e-16)
            perp = np.eye(3) - np.outer(u, u)
            a, b = (p - pm) @ perp.T, (q0 - pm) @ perp.T
            s1, s2 = float(np.sum(a * b)), float(np.sum(a * np.cross(u, b)))
            theta = np.arctan2(s2, s1) if (abs(s1) + abs(s2)) > 1e-16 else 0.0
            out = (q0 - pm) @ _rodrigues(u, theta).T + pm
            return out, {"before_A": rmsd_ang(p, q), "after_A": rmsd_ang(p, out), "n_used": 2, "mode": "two_anchor"}
        report_all = True
    use = np.zeros(n, dtype=bool)
    if idx:
        use[idx] = True
    else:
        use[:] = True
    rot, t = kabsch_R_t(p[use], q[use])
    out = q @ rot + t
    before = rmsd_ang(p, q) if report_all else rmsd_ang(p[use], q[use])
    after = rmsd_ang(p, out) if report_all else rmsd_ang(p[use], out[use])
    return out, {"before_A": before, "after_A": after, "n_used": int(use.sum()), "mode": "kabsch"}


@with_small_host_math
def scan_toward_target(calc, elem, ref_bohr: np.ndarray, mob_bohr: np.ndarray, anchors: Sequence[int], *, step_A: float = 0.1,
                       per_step_cycles: int = 50, final_cycles: int = 200, max_steps: int = 1000, thresh="gau", verbose: bool = False):
    """Staged scan of ONE OR MORE mobile images toward their references (reference
    ``scan_freeze_atoms_toward_target_inplace``, :390-517).

    ``ref_bohr``/``mob_bohr``: (N,3) or (K,N,3) Bohr.  Every step moves each anchor ``step_A`` Angstrom along its remaining
    displacement (anchors closer than 1e-12 Bohr stay), holds the anchors fixed and relaxes the rest for at most
    ``per_step_cycles`` L-BFGS cycles; once the largest remaining distance is within one step the anchors are set exactly
    onto the reference and a final relaxation of ``final_cycles`` runs.  All K images share each batched E+F call.
    Returns (coords like `mob_bohr`, [ {max_remaining_A, n_steps, converged} per image ]) -- a single dict for 2-D input.
    """
    from .lbfgs import BatchedLBFGS

    single = np.asarray(mob_bohr).ndim == 2
    q = np.array(mob_bohr, dtype=float).reshape((1, -1, 3) if single else (np.asarray(mob_bohr).shape[0], -1, 3))
    p = np.broadcast_to(np.asarray(ref_bohr, dtype=float).reshape((1, -1, 3) if np.asarray(ref_bohr).ndim == 2 else q.shape), q.shape)
    k, n = q.shape[0], q.shape[1]
    idx = np.array([int(i) for i in anchors if 0 <= int(i) < n], dtype=int)
    info = [{"max_remaining_A": 0.0, "n_steps": 0, "converged": True} for _ in range(k)]
    if idx.size == 0:
        return (q[0] if single else q), (info[0] if single else info)
    step_bohr = float(step_A) / BOHR2ANG
    done = np.zeros(k, dtype=bool)
    for inf in info:
        inf["converged"] = False
    for istep in range(1, int(max_steps) + 1):
        live = np.flatnonzero(~done)
        if live.size == 0:
            break
        d = p[live][:, idx, :] - q[live][:, idx, :]
        rem = np.linalg.norm(d, axis=2)                                   # (live, anchors)
        max_rem = rem.max(axis=1)
        final = max_rem <= step_bohr + 1e-12
        budgets = np.where(final, int(final_cycles), int(per_step_cycles))
        for row, img in enumerate(live):
            info[img]["max_remaining_A"] = float(max_rem[row] * BOHR2ANG)
            info[img]["n_steps"] = istep
            if final[row]:
                q[img, idx] = p[img, idx]
            else:
                sel = rem[row] > 1e-12
                q[img, idx[sel]] += d[row, sel] / rem[row, sel, None] * step_bohr
        if verbose:
            print(f"[scan] step {istep:03d}: max remaining = {max(info[i]['max_remaining_A'] for i in live):.6f} A ({live.size} image(s))")
        opt = BatchedLBFGS(calc, elem, q[live], freeze=list(idx), thresh=thresh, max_cycles=budgets)
        q[live] = opt.run()["coords"]
        for row, img in enumerate(live):
            if final[row]:
                done[img] = True
                info[img]["converged"] = True
    return (q[0] if single else q), (info[0] if single else info)


def align_and_refine_pair(calc, elem, ref_bohr, mob_bohr, freeze_ref=(), freeze_mob=(), **scan_kw):
    """Rigid alignment then the staged scan for one (reference, mobile) pair (reference ``align_and_refine_pair_inplace``)."""
    n = np.asarray(ref_bohr).reshape(-1, 3).shape[0]
    idx = freeze_union(freeze_ref, freeze_mob, n)
    aligned, a_info = align_second_to_first(ref_bohr, mob_bohr, idx)
    out, s_info = scan_toward_target(calc, elem, np.asarray(ref_bohr, dtype=float).reshape(-1, 3), aligned, idx, **scan_kw)
    return out, {"align": a_info, "scan": s_info}


@with_small_host_math
def align_and_refine_sequence(calc, elem, coords_list, freeze_list=None, *, batched: bool = True, **scan_kw):
    """Pairs (g0<-g1), (g1<-g2), ... along a list of geometries (reference ``align_and_refine_sequence_inplace``).

    With three or more common anchors the fit and the scan targets of every pair depend only on the anchor positions,
    which after each converged scan equal g0's anchors -- so with ``batched=True`` all mobile images are fitted onto g0's
    anchors and scanned TOGETHER (one batched E+F per L-BFGS cycle instead of one per image).  Other cases (1-2 anchors,
    differing freeze lists) run pair after pair exactly in the reference's order.
    Returns (list of coordinates, list of {"align", "scan"} per pair); element 0 is returned unchanged.
    """
    geoms = [np.asarray(c, dtype=float).reshape(-1, 3) for c in coords_list]
    m = len(geoms)
    fl = [list(f) for f in (freeze_list if freeze_list is not None else [[]] * m)]
    if m < 2:
        return geoms, []
    n = geoms[0].shape[0]
    unions = [freeze_union(fl[i], fl[i + 1], n) for i in range(m - 1)]
    if batched and all(u == unions[0] for u in unions) and len(unions[0]) >= 3:
        idx = unions[0]
        aligned, a_infos = [], []
        target = geoms[0].copy()
        for i in range(1, m):
            # the reference fits g_i onto g_{i-1}; its anchors coincide with g0's once the previous pair has converged
            ref = geoms[i - 1].copy(); ref[idx] = geoms[0][idx]
            q, ai = align_second_to_first(ref, geoms[i], idx)
            aligned.append(q); a_infos.append(ai)
        out, s_infos = scan_toward_target(calc, elem, np.broadcast_to(target, (m - 1, n, 3)), np.stack(aligned), idx, **scan_kw)
        return [geoms[0]] + [out[i] for i in range(m - 1)], [{"align": a, "scan": s} for a, s in zip(a_infos, s_infos)]
    results = []
    for i in range(m - 1):
        geoms[i + 1], res = align_and_refine_pair(calc, elem, geoms[i], geoms[i + 1], fl[i], fl[i + 1], **scan_kw)
        results.append(res)
    return geoms, results
