"""CPU tests of the batched Growing-String driver on the Mueller-Brown surface."""
import numpy as np
import pytest

from pdb2reaction_amd.gsm import GS_KW, STOPT_KW, GrowingStringDriver

A = np.array([-200.0, -100.0, -170.0, 15.0])
a = np.array([-1.0, -1.0, -6.5, 0.7])
b = np.array([0.0, 0.0, 11.0, 0.6])
c = np.array([-10.0, -10.0, -6.5, 0.7])
X0 = np.array([1.0, 0.0, -0.5, -1.0])
Y0 = np.array([0.0, 0.5, 1.5, 1.0])
SCALE = 1e-3          # bring the surface to a Hartree-like scale


class MuellerBrown:
    """One 'atom' at (x, y, z): E = SCALE * MB(x, y) + z^2/2; batched interface of uma_pysis.get_forces_batch."""

    def __init__(self):
        self.batches = []

    def get_forces_batch(self, atoms, coords):
        q = np.asarray(coords, dtype=float).reshape(len(coords), 3)
        self.batches.append(len(q))
        x, y, z = q[:, 0:1], q[:, 1:2], q[:, 2]
        dx, dy = x - X0, y - Y0
        t = A * np.exp(a * dx ** 2 + b * dx * dy + c * dy ** 2)
        e = SCALE * t.sum(1) + 0.5 * z ** 2
        fx = -SCALE * (t * (2 * a * dx + b * dy)).sum(1)
        fy = -SCALE * (t * (b * dx + 2 * c * dy)).sum(1)
        return {"energy": e, "forces": np.stack([fx, fy, -z], axis=1)}


MIN_A = np.array([-0.558224, 1.441726, 0.0])
MIN_B = np.array([0.623499, 0.028038, 0.0])
SADDLE_1 = (-0.822002, 0.624313, -40.664843)        # highest saddle on the A -> B path


def test_defaults_mirror_reference():
    assert GS_KW["max_nodes"] == 10 and GS_KW["perp_thresh"] == 5e-3 and GS_KW["climb_rms"] == 5e-4 and GS_KW["fix_first"] is True
    assert STOPT_KW["stop_in_when_full"] == 300 and STOPT_KW["max_cycles"] == 300 and STOPT_KW["scale_step"] == "global"


def test_gsm_finds_the_mueller_brown_saddle():
    calc = MuellerBrown()
    drv = GrowingStringDriver(["X"], MIN_A, MIN_B, calc, gs_kw={"max_nodes": 13, "perp_thresh": 2e-2, "climb_rms": 5e-3},
                              stopt_kw={"thresh": "gau", "max_step": 0.05, "max_cycles": 600})
    res = drv.run()
    assert res.fully_grown and len(res.coords) == 15
    assert res.converged, res.history[-1]
    hei = res.coords[res.hei_index]
    assert abs(res.energies[res.hei_index] - SCALE * SADDLE_1[2]) < 2e-5          # climbing image sits on the saddle
    assert np.hypot(hei[0] - SADDLE_1[0], hei[1] - SADDLE_1[1]) < 2e-2
    assert np.abs(res.coords[:, 2]).max() < 1e-6
    assert np.array_equal(res.coords[0], MIN_A) and np.array_equal(res.coords[-1], MIN_B)   # fixed endpoints untouched
    # one batched call per cycle; fixed endpoints are evaluated only when the string changes size
    assert len(calc.batches) <= res.cycles + 1 and max(calc.batches) <= 15
    assert res.force_evaluations == sum(calc.batches)
    assert min(calc.batches[-5:]) == 13                                               # K-2 evaluations per cycle once grown
    # nodes (except the climbing image's neighbourhood) are spread along the path
    seg = np.linalg.norm(np.diff(res.coords, axis=0), axis=1)
    assert seg.max() / seg.min() < 4.0


def test_growth_sequence_and_small_strings():
    calc = MuellerBrown()
    drv = GrowingStringDriver(["X"], MIN_A, MIN_B, calc, gs_kw={"max_nodes": 4, "perp_thresh": 1e9, "climb": False},
                              stopt_kw={"max_cycles": 3})
    assert len(drv.coords) == 4 and not drv.fully_grown
    res = drv.run()
    assert res.fully_grown and len(res.coords) == 6                                  # grows one node per side per cycle
    sizes = [int(h["images"]) for h in res.history]
    assert sizes == [4, 6, 6]
    drv1 = GrowingStringDriver(["X"], MIN_A, MIN_B, calc, gs_kw={"max_nodes": 1}, stopt_kw={"max_cycles": 2})
    assert len(drv1.coords) == 3 and drv1.fully_grown
    with pytest.raises(ValueError):
        GrowingStringDriver(["X", "Y"], MIN_A, MIN_B, calc)
    with pytest.raises(NotImplementedError):
        GrowingStringDriver(["X"], MIN_A, MIN_B, calc, gs_kw={"param": "energy"})


def test_custom_evaluator_is_used():
    calc = MuellerBrown()
    seen = []

    def evaluate(x):
        seen.append(len(x))
        r = calc.get_forces_batch(["X"], x)
        return r["energy"], r["forces"]

    drv = GrowingStringDriver(["X"], MIN_A, MIN_B, calc=None, evaluate=evaluate, gs_kw={"max_nodes": 3}, stopt_kw={"max_cycles": 4})
    drv.run()
    assert seen and seen[0] == 4


def test_unconverged_exit_returns_energies_of_the_returned_coordinates():
    """max_cycles exhausted right after a step: energies must belong to the returned geometries."""
    calc = MuellerBrown()
    drv = GrowingStringDriver(["X"], MIN_A, MIN_B, calc, gs_kw={"max_nodes": 3, "climb": False}, stopt_kw={"max_cycles": 5})
    res = drv.run()
    assert not res.converged and res.cycles == 5
    e_now = calc.get_forces_batch(["X"], res.coords)["energy"]
    np.testing.assert_allclose(res.energies, e_now, rtol=0, atol=1e-12)
