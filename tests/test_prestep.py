"""CPU tests of the row-f3 helpers (Kabsch alignment, harmonic bias wrapper)."""
import importlib

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from pdb2reaction_amd.prestep import HarmonicBias, align_onto, kabsch_R_t

U = importlib.import_module("pdb2reaction_amd.uma_pysis")


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
