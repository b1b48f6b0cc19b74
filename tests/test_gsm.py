"""CPU tests of the batched Growing-String driver on the Mueller-Brown surface."""
import numpy as np
import pytest

from pdb2reaction_amd.gsm import GS_KW, STOPT_KW, GrowingStringDriver, _tangents, lanczos_lowest_mode

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
    # (the Lanczos tangent of the climbing image costs extra single-image evaluations: counted separately)
    assert drv.lanczos_evals > 0 and calc.batches.count(1) >= drv.lanczos_evals
    assert len(calc.batches) - drv.lanczos_evals <= res.cycles + 1 and max(calc.batches) <= 15
    assert res.force_evaluations == sum(calc.batches)
    assert sorted(set(calc.batches[-12:])) == [1, 13]                                 # K-2 evaluations per cycle once grown (+ Lanczos singles)
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


def test_lanczos_finds_the_lowest_mode_of_a_known_hessian():
    """climb_lanczos (reference GS_KW, path_opt.py:181-182): lowest-curvature direction from gradient differences only."""
    rng = np.random.default_rng(3)
    n = 30
    qmat, _ = np.linalg.qr(rng.standard_normal((n, n)))
    evals = np.concatenate([[-0.8], np.linspace(0.3, 4.0, n - 1)])
    hess = (qmat * evals) @ qmat.T
    x0 = rng.standard_normal(n)
    grad = lambda x: hess @ (x - 0.1) + 1e-3 * np.sin(x)          # slightly anharmonic
    guess = qmat[:, 0] + 0.6 * rng.standard_normal(n) / np.sqrt(n)
    w, v, steps = lanczos_lowest_mode(grad, x0, grad(x0), guess, dx=1e-4, dl=1e-5, max_cycles=30)
    assert abs(w - (-0.8)) < 5e-3 and steps <= 30
    assert abs(v @ qmat[:, 0]) > 0.995 and v @ guess > 0 and abs(np.linalg.norm(v) - 1) < 1e-12
    with pytest.raises(ValueError):
        lanczos_lowest_mode(grad, x0, grad(x0), np.zeros(n))


def test_spline_tangents_follow_the_curve():
    th = np.linspace(0.2, 2.6, 9)
    x = np.stack([2 * np.cos(th), 2 * np.sin(th), 0.3 * th], axis=1)
    exact = np.stack([-2 * np.sin(th), 2 * np.cos(th), 0.3 * np.ones_like(th)], axis=1)
    exact /= np.linalg.norm(exact, axis=1, keepdims=True)
    ts, tc = _tangents(x, "spline"), _tangents(x, "central")
    err_s = np.linalg.norm(ts - exact, axis=1).max()
    err_c = np.linalg.norm(tc - exact, axis=1).max()
    assert err_s < 5e-3 and err_s < 0.1 * err_c                    # one-sided end differences are the weak spot of "central"
    assert np.allclose(np.linalg.norm(ts, axis=1), 1.0)
    assert np.array_equal(_tangents(x[:3], "spline"), _tangents(x[:3], "central"))     # too few images: fallback


def test_driver_announces_the_final_string_size_to_the_calculator():
    """The string grows to max_nodes + 2 images: the driver says so up front (`reserve_images`), so that the engine allocates its workspace
    once instead of once per growth (seconds each on the GPU); calculators without the method are left alone."""
    class Announced(MuellerBrown):
        def __init__(self):
            super().__init__()
            self.reserved = []

        def reserve_images(self, n):
            self.reserved.append(n)

    calc = Announced()
    GrowingStringDriver(["X"], MIN_A, MIN_B, calc, gs_kw={"max_nodes": 7})
    assert calc.reserved == [9]
    GrowingStringDriver(["X"], MIN_A, MIN_B, MuellerBrown(), gs_kw={"max_nodes": 7})      # no such method: nothing happens

    import importlib
    U = importlib.import_module("pdb2reaction_amd.uma_pysis")
    c = U.uma_pysis()
    c.reserve_images(12)                                       # before the engine exists: kept and applied when the core is built
    assert c._reserve_images == 12 and c._core is None
