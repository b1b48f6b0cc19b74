"""CPU tests of the row-f3 helpers (Kabsch alignment, harmonic bias wrapper)."""
import importlib

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from pdb2reaction_amd.prestep import HarmonicBias, align_onto, kabsch_R_t

U = importlib.import_module("pdb2reaction_amd.uma_pysis")
P = importlib.import_module("pdb2reaction_amd.prestep")


def test_kabsch_recovers_rigid_motion_and_rejects_reflections():
    rng = np.random.default_rng(0)
    q = rng.standard_normal((12, 3))
    rot = Rotation.random(random_state=1).as_matrix()
    p = q @ rot + np.array([1.0, -2.0, 0.3])
    r, t = kabsch_R_t(p, q)
    assert np.allclose(q @ r + t, p, atol=1e-12) and np.linalg.det(r) == pytest.approx(1.0)
    mirrored = q * np.array([1.0, 1.0, -1.0])
    r2, _ = kabsch_R_t(p, mirrored)
    assert np.linalg.det(r2) == pytest.approx(1.0)                      # improper fit is folded back to a rotation
    assert np.allclose(align_onto(p, q, subset=range(6)), p, atol=1e-12)
    with pytest.raises(ValueError):
        kabsch_R_t(p, q[:5])


class Zero:
    extra = "forwarded"

    def get_forces(self, elem, coords):
        return {"energy": 1.0, "forces": np.zeros(np.asarray(coords).size)}

    def get_energy(self, elem, coords):
        return {"energy": 1.0}

    def get_forces_batch(self, elem, coords):
        c = np.asarray(coords)
        return {"energy": np.ones(len(c)), "forces": np.zeros((len(c), c[0].size))}


def test_harmonic_bias_units_and_gradient():
    b = HarmonicBias(Zero(), k=10.0, pairs=[(0, 1, 1.0), (1, 2, 2.0), (0, 9, 1.0)])      # last pair is out of range -> skipped
    x = np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 2.5], [3.0, 0.0, 2.5]])                   # Bohr
    r = b.get_forces(["H"] * 3, x.reshape(-1))
    k_au = 10.0 * U.H_EVAA_2_AU
    d01, d12 = 2.5 - 1.0 * U.ANG2BOHR, 3.0 - 2.0 * U.ANG2BOHR
    assert r["energy"] == pytest.approx(1.0 + 0.5 * k_au * (d01 ** 2 + d12 ** 2), rel=1e-12)
    h = 1e-5
    for i in range(9):
        xp, xm = x.reshape(-1).copy(), x.reshape(-1).copy()
        xp[i] += h
        xm[i] -= h
        fd = -(b.get_energy(["H"] * 3, xp)["energy"] - b.get_energy(["H"] * 3, xm)["energy"]) / (2 * h)
        assert r["forces"][i] == pytest.approx(fd, abs=1e-9)
    e, g = b.get_energy_and_gradient(["H"] * 3, x)
    assert np.allclose(g, -r["forces"]) and b.extra == "forwarded"
    coincident = np.zeros((3, 3))
    assert b.get_energy(["H"] * 3, coincident)["energy"] == 1.0                             # |r_ij| < 1e-14 pairs are skipped


def test_harmonic_bias_batch_matches_serial():
    b = HarmonicBias(Zero(), k=3.0, pairs=[(0, 2, 1.2), (1, 2, 0.9)])
    xb = np.random.default_rng(2).standard_normal((5, 9))
    rb = b.get_forces_batch(["C"] * 3, xb)
    for k in range(5):
        r = b.get_forces(["C"] * 3, xb[k])
        assert rb["energy"][k] == pytest.approx(r["energy"], rel=1e-14) and np.allclose(rb["forces"][k], r["forces"], atol=1e-15)


# ---- alignment special cases, batched L-BFGS, staged scan (row f3; reference align_freeze_atoms.py:245-517) ------------
class _SpringNet:
    """Toy calculator: harmonic bonds between consecutive atoms (rest length r0) -- translation/rotation invariant."""

    def __init__(self, r0=2.0, k=0.8):
        self.r0, self.k, self.calls, self.images = r0, k, 0, 0

    def get_forces_batch(self, elem, coords):
        c = np.asarray(coords, dtype=float).reshape(len(coords), -1, 3)
        self.calls += 1; self.images += len(c)
        d = c[:, 1:] - c[:, :-1]
        r = np.linalg.norm(d, axis=2)
        e = 0.5 * self.k * ((r - self.r0) ** 2).sum(axis=1)
        g = (self.k * (r - self.r0) / r)[..., None] * d
        f = np.zeros_like(c)
        f[:, 1:] -= g
        f[:, :-1] += g
        return {"energy": e, "forces": f.reshape(len(c), -1)}


def _rigid_copy(p, seed):
    rng = np.random.default_rng(seed)
    rot = P._rodrigues(rng.normal(size=3), 1.1)
    return p @ rot.T + rng.normal(size=3) * 3.0


def test_alignment_modes_recover_rigid_copy():
    rng = np.random.default_rng(3)
    p = rng.normal(size=(12, 3)) * 4.0
    q = _rigid_copy(p, 4)
    for anchors, mode in (([], "kabsch"), ([1, 5, 7], "kabsch"), ([2], "one_anchor"), ([2, 9], "two_anchor")):
        out, info = P.align_second_to_first(p, q, anchors)
        assert info["mode"] == mode and info["n_used"] == (len(anchors) or 12)
        np.testing.assert_allclose(out, p, atol=1e-9)
        assert info["after_A"] < 1e-9 < info["before_A"]
    # a NON-rigid mobile: the one-anchor mode pins the anchor and only rotates about it
    q2 = q + rng.normal(size=q.shape) * 0.3
    out, info = P.align_second_to_first(p, q2, [2])
    np.testing.assert_allclose(out[2], p[2], atol=1e-12)
    np.testing.assert_allclose(np.linalg.norm(out - out[2], axis=1), np.linalg.norm(q2 - q2[2], axis=1), atol=1e-10)
    assert info["after_A"] < info["before_A"]
    # two anchors: midpoints coincide and the anchor axes are parallel
    out, info = P.align_second_to_first(p, q2, [2, 9])
    np.testing.assert_allclose(0.5 * (out[2] + out[9]), 0.5 * (p[2] + p[9]), atol=1e-10)
    assert np.linalg.norm(np.cross(out[9] - out[2], p[9] - p[2])) < 1e-9
    # degenerate axis (coincident anchors in the reference) falls back to Kabsch on the two anchors, all-atom RMSD
    pd = p.copy(); pd[9] = pd[2]
    _, info = P.align_second_to_first(pd, q2, [2, 9])
    assert info["mode"] == "kabsch" and info["n_used"] == 2
    assert P.freeze_union([3, 1, 99], [1, 2, -1], 10) == [1, 2, 3]
    with pytest.raises(ValueError):
        P.align_second_to_first(p, q[:5], [])


def test_batched_lbfgs_matches_per_image_runs():
    from pdb2reaction_amd.lbfgs import BatchedLBFGS
    rng = np.random.default_rng(7)
    base = np.cumsum(np.tile([[2.0, 0.3, -0.2]], (8, 1)), axis=0)
    x0 = base[None] + rng.normal(size=(3, 8, 3)) * 0.4
    calc = _SpringNet()
    batch = BatchedLBFGS(calc, ["C"] * 8, x0, freeze=[0, 7], thresh="gau_tight", max_cycles=200).run()
    assert batch["converged"].all() and calc.calls == batch["n_calls"] <= batch["cycles"].max() + 1
    for i in range(3):
        one = BatchedLBFGS(_SpringNet(), ["C"] * 8, x0[i], freeze=[0, 7], thresh="gau_tight", max_cycles=200).run()
        np.testing.assert_allclose(one["coords"][0], batch["coords"][i], atol=1e-12)
        assert one["cycles"][0] == batch["cycles"][i]
    np.testing.assert_allclose(batch["coords"][:, [0, 7]], x0[:, [0, 7]], atol=0)          # frozen atoms never move
    r = np.linalg.norm(np.diff(batch["coords"], axis=1), axis=2)
    assert np.abs(batch["forces"]).max() < 1.5e-5 and batch["energies"].max() < 2.0          # end-pinned chain: relaxed, not rest length
    assert r.std(axis=1).max() < 1e-3                                                        # ... with equal bond lengths
    # per-image budgets + calculators without a batch entry point
    class OnlySingle:
        def __init__(self): self.c = _SpringNet()
        def get_forces(self, elem, coords):
            r_ = self.c.get_forces_batch(elem, np.asarray(coords)[None])
            return {"energy": float(r_["energy"][0]), "forces": r_["forces"][0]}
    lim = BatchedLBFGS(OnlySingle(), ["C"] * 8, x0, thresh="gau_tight", max_cycles=[1, 3, 200]).run()
    assert list(lim["cycles"][:2]) == [1, 3] and not lim["converged"][0] and lim["converged"][2]


def test_staged_scan_moves_anchors_exactly_and_batches():
    rng = np.random.default_rng(11)
    ref = np.cumsum(np.tile([[2.0, 0.0, 0.0]], (10, 1)), axis=0) / P.BOHR2ANG * 0 + np.cumsum(np.tile([[2.0, 0.0, 0.0]], (10, 1)), axis=0)
    anchors = [0, 4, 9]
    mobs = np.stack([ref + rng.normal(size=ref.shape) * 0.25 for _ in range(3)])
    mobs[:, 4] += [0.0, 0.9, 0.0]                                   # anchor 4 starts ~0.9 Bohr (~0.48 A) away: several scan steps
    calc = _SpringNet()
    out, infos = P.scan_toward_target(calc, ["C"] * 10, ref, mobs, anchors, step_A=0.1, per_step_cycles=30, final_cycles=100, thresh="gau")
    assert all(i["converged"] for i in infos) and all(4 <= i["n_steps"] <= 8 for i in infos)
    np.testing.assert_array_equal(out[:, anchors], np.broadcast_to(ref[anchors], (3, 3, 3)))     # exact coincidence
    assert all(i["max_remaining_A"] <= 0.1 + 1e-9 for i in infos)
    batched_images = calc.images / calc.calls
    assert batched_images > 1.5                                                                   # images shared the E+F calls
    # one image alone takes the same path as inside the batch
    one, info1 = P.scan_toward_target(_SpringNet(), ["C"] * 10, ref, mobs[1], anchors, step_A=0.1, per_step_cycles=30, final_cycles=100)
    np.testing.assert_allclose(one, out[1], atol=1e-12)
    assert info1["n_steps"] == infos[1]["n_steps"] and isinstance(info1, dict)
    # no anchors: nothing to do (reference :432-436)
    same, inf0 = P.scan_toward_target(calc, ["C"] * 10, ref, mobs[0], [])
    np.testing.assert_array_equal(same, mobs[0]); assert inf0 == {"max_remaining_A": 0.0, "n_steps": 0, "converged": True}
    # max_steps exhausted: reported, not raised
    _, inf_short = P.scan_toward_target(_SpringNet(), ["C"] * 10, ref, mobs[0], anchors, step_A=0.05, max_steps=2, per_step_cycles=5)
    assert inf_short["converged"] is False and inf_short["n_steps"] == 2


def test_sequence_batched_equals_pair_by_pair():
    rng = np.random.default_rng(5)
    g0 = np.cumsum(np.tile([[2.0, 0.1, 0.0]], (9, 1)), axis=0)
    geoms = [g0] + [_rigid_copy(g0 + rng.normal(size=g0.shape) * 0.15, 20 + i) for i in range(3)]
    fz = [[0, 3, 8]] * 4
    kw = dict(step_A=0.1, per_step_cycles=40, final_cycles=150, thresh="gau_tight")
    c_b, c_s = _SpringNet(), _SpringNet()
    out_b, res_b = P.align_and_refine_sequence(c_b, ["C"] * 9, geoms, fz, batched=True, **kw)
    out_s, res_s = P.align_and_refine_sequence(c_s, ["C"] * 9, geoms, fz, batched=False, **kw)
    assert len(res_b) == len(res_s) == 3 and all(r["scan"]["converged"] for r in res_b + res_s)
    for a, b in zip(out_b, out_s):
        np.testing.assert_allclose(a, b, atol=1e-9)
        np.testing.assert_array_equal(a[[0, 3, 8]], g0[[0, 3, 8]])
    assert c_b.calls < c_s.calls                                     # fewer (batched) engine calls for the same result
    assert [r["align"]["mode"] for r in res_b] == ["kabsch"] * 3
    # two anchors: sequential path is taken even when batched=True
    out2, res2 = P.align_and_refine_sequence(_SpringNet(), ["C"] * 9, geoms[:2], [[0, 8]] * 2, batched=True, **kw)
    assert res2[0]["align"]["mode"] == "two_anchor" and res2[0]["scan"]["converged"]
