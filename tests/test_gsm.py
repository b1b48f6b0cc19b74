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


# ---- round 4: the driver's arithmetic lives on torch tensors (the engine's device on a GPU box) ----------------------------------
def test_spline_derivative_equals_scipy_not_a_knot():
    import torch
    from scipy.interpolate import CubicSpline, make_interp_spline
    from pdb2reaction_amd.gsm import spline_derivative_t

    rng = np.random.default_rng(0)
    for k in (4, 5, 9, 16):
        x = np.cumsum(rng.standard_normal((k, 7)), axis=0)
        u = np.concatenate([[0.0], np.cumsum(rng.uniform(0.3, 2.0, k - 1))])
        d = spline_derivative_t(torch.as_tensor(u), torch.as_tensor(x)).numpy()
        ref = make_interp_spline(u, x, k=3)(u, 1)
        np.testing.assert_allclose(d, ref, rtol=0, atol=5e-12 * max(1.0, np.abs(ref).max()))
        np.testing.assert_allclose(d, CubicSpline(u, x, bc_type="not-a-knot")(u, 1), rtol=0, atol=5e-12 * max(1.0, np.abs(ref).max()))


def test_compact_lbfgs_equals_the_two_loop_recursion():
    import torch
    from pdb2reaction_amd.gsm import lbfgs_direction_t, lbfgs_two_loop_t

    g = torch.Generator().manual_seed(5)
    n = 60
    m0 = torch.randn(n, n, dtype=torch.float64, generator=g)
    hess = m0 @ m0.T / n + torch.eye(n, dtype=torch.float64)            # s.y > 0 for every pair
    for m in (1, 2, 5, 10):
        s_hist = torch.randn(m, n, dtype=torch.float64, generator=g)
        y_hist = s_hist @ hess
        grad = torch.randn(n, dtype=torch.float64, generator=g)
        a, b = lbfgs_direction_t(s_hist, y_hist, grad), lbfgs_two_loop_t(s_hist, y_hist, grad)
        assert float((a - b).abs().max()) <= 1e-11 * float(b.abs().max()) and float(a @ grad) < 0
    assert torch.equal(lbfgs_direction_t(s_hist[:0], y_hist[:0], grad), -grad)


def test_device_side_helpers_match_their_host_definitions():
    import torch
    from pdb2reaction_amd.gsm import hei_index_t, place_t
    from pdb2reaction_amd.string import select_hei_index

    rng = np.random.default_rng(1)
    for n in (2, 3, 4, 9, 16):
        for _ in range(40):
            e = rng.standard_normal(n)
            if rng.random() < 0.3:
                e = np.sort(e)                                               # monotone: no internal maximum
            assert int(hei_index_t(torch.as_tensor(e))) == select_hei_index(e)
    x = np.cumsum(rng.standard_normal((7, 5)), axis=0)
    seg = np.linalg.norm(np.diff(x, axis=0), axis=1)
    s = np.concatenate([[0.0], np.cumsum(seg)])
    tg = np.array([0.0, 0.1, 0.37, 0.5, 0.93, 1.0])
    out = place_t(torch.as_tensor(x), torch.as_tensor(tg)).numpy()
    for o, t in zip(out, tg):
        uu = t * s[-1]
        i = int(np.clip(np.searchsorted(s, uu, side="right"), 1, len(s) - 1))
        w = (uu - s[i - 1]) / (s[i] - s[i - 1])
        np.testing.assert_allclose(o, x[i - 1] * (1 - w) + x[i] * w, atol=1e-13)


def test_device_evaluator_contract_and_timing_split():
    """evaluate_device: torch in, torch out, same results as the numpy evaluator; the result says where the wall time went."""
    import torch

    calc = MuellerBrown()
    kw = dict(gs_kw={"max_nodes": 9, "perp_thresh": 2e-2, "climb_rms": 5e-3}, stopt_kw={"thresh": "gau", "max_step": 0.05, "max_cycles": 400})
    ref = GrowingStringDriver(["X"], MIN_A, MIN_B, calc, **kw).run()
    seen = []

    def evaluate_device(x):
        assert torch.is_tensor(x) and x.dtype == torch.float64
        seen.append(int(x.shape[0]))
        r = calc.get_forces_batch(["X"], x.numpy())
        return torch.as_tensor(r["energy"]), torch.as_tensor(r["forces"])

    res = GrowingStringDriver(["X"], MIN_A, MIN_B, calc=None, evaluate_device=evaluate_device, device=torch.device("cpu"), **kw).run()
    assert res.converged and res.cycles == ref.cycles and np.array_equal(res.coords, ref.coords) and np.array_equal(res.energies, ref.energies)
    assert seen and res.force_evaluations == sum(seen)
    t = res.timing
    assert t["total_s"] > 0 and 0 <= t["evaluator_s"] <= t["total_s"] and abs(t["host_s"] + t["evaluator_s"] - t["total_s"]) < 1e-9
    with pytest.raises(ValueError, match="evaluate_device"):
        GrowingStringDriver(["X"], MIN_A, MIN_B, calc, device=torch.device("cuda", 0))


def test_unimplemented_keywords_are_refused_or_warned_about():
    """a reference keyword this driver ignores must not be accepted in silence when it carries a non-default value."""
    calc = MuellerBrown()
    for kw in ({"gs_kw": {"scheduler": object()}}, {"stopt_kw": {"align": True}}, {"stopt_kw": {"scale_step": "per_atom"}},
               {"geom_kw": {"coord_type": "dlc"}}, {"gs_kw": {"param": "energy"}}):
        with pytest.raises(NotImplementedError):
            GrowingStringDriver(["X"], MIN_A, MIN_B, calc, **kw)
    for kw, word in (({"gs_kw": {"reparam_check": "norm"}}, "reparam_check"), ({"gs_kw": {"max_micro_cycles": 25}}, "max_micro_cycles"),
                     ({"gs_kw": {"reset_dlc": False}}, "reset_dlc"), ({"stopt_kw": {"coord_diff_thresh": 1e-3}}, "coord_diff_thresh"),
                     ({"stopt_kw": {"reparam_thresh": 1e-3}}, "reparam_thresh"), ({"stopt_kw": {"dump": True}}, "dump")):
        with pytest.warns(RuntimeWarning, match=word):
            GrowingStringDriver(["X"], MIN_A, MIN_B, calc, **kw)
    import warnings as _w
    with _w.catch_warnings():
        _w.simplefilter("error")                                           # the reference's own defaults pass without a word
        GrowingStringDriver(["X"], MIN_A, MIN_B, calc, gs_kw=dict(GS_KW), stopt_kw=dict(STOPT_KW), geom_kw={"coord_type": "cart", "freeze_atoms": []})


def test_climb_fixed_keeps_the_image_that_started_to_climb():
    """Reference GS_KW `climb_fixed` (path_opt.py:183; default False): the climbing image is determined ONCE, when climbing starts, instead of being the
    highest image of every cycle.  On the Mueller-Brown string both settings end on the same saddle; the fixed index is recorded."""
    res = {}
    for fixed in (False, True):
        drv = GrowingStringDriver(["X"], MIN_A, MIN_B, MuellerBrown(), gs_kw={"max_nodes": 13, "perp_thresh": 2e-2, "climb_rms": 5e-3, "climb_fixed": fixed},
                                  stopt_kw={"thresh": "gau", "max_step": 0.05, "max_cycles": 600})
        res[fixed] = (drv.run(), drv.fixed_climb_index)
    (r0, i0), (r1, i1) = res[False], res[True]
    assert i0 is None and i1 is not None and 0 < i1 < len(r1.coords) - 1
    assert r1.converged and abs(r1.energies[i1] - SCALE * SADDLE_1[2]) < 2e-5            # the FIXED image sits on the saddle
    assert r0.converged and abs(r0.energies[r0.hei_index] - SCALE * SADDLE_1[2]) < 2e-5


def test_scale_step_per_image_scales_every_image_on_its_own():
    """Reference STOPT_KW `scale_step` (path_opt.py:193): "global" shortens the whole step by one factor so that its largest component is max_step;
    "per_image" shortens every image whose own largest component exceeds max_step, and only those.  One steepest-descent cycle, no growth, no
    re-parametrisation: the displacement of each moving image is read off the coordinates."""
    moved = {}
    for ss in ("global", "per_image"):
        drv = GrowingStringDriver(["X"], MIN_A, MIN_B, MuellerBrown(), gs_kw={"max_nodes": 3, "perp_thresh": 1e-14, "reparam_every": 0, "reparam_every_full": 0},
                                  stopt_kw={"max_step": 1e-4, "max_cycles": 1, "scale_step": ss})
        x0 = drv.coords.copy()
        drv.run()
        assert drv.coords.shape == x0.shape
        moved[ss] = np.abs(drv.coords - x0).max(axis=1)
    inner = moved["global"] > 0
    assert inner.sum() >= 2 and not inner[0] and not inner[-1]
    assert np.isclose(moved["global"].max(), 1e-4, rtol=1e-9) and moved["global"][inner].min() < 0.9e-4        # one factor: only the largest image reaches max_step
    assert np.allclose(moved["per_image"][inner], 1e-4, rtol=1e-9)                                            # every image reaches it on its own
    with pytest.raises(NotImplementedError):
        GrowingStringDriver(["X"], MIN_A, MIN_B, MuellerBrown(), stopt_kw={"scale_step": "per_atom"})


class MuellerBrown24:
    """8 'atoms' x 3 = 24 coordinates: the Mueller-Brown surface in a rotated plane + a harmonic bath (k = 0.3 ... 1.5) weakly coupled to it --
    a Hessian with a spread spectrum, so that the Krylov space of a Lanczos recursion is not exhausted after two steps."""

    def __init__(self):
        rng = np.random.default_rng(11)
        self.q, _ = np.linalg.qr(rng.standard_normal((24, 24)))
        self.k = np.linspace(0.3, 1.5, 22)
        self.cpl = 0.02
        self.batches = []

    def get_forces_batch(self, atoms, coords):
        xx = np.asarray(coords, dtype=float).reshape(len(coords), 24)
        self.batches.append(len(xx))
        u = xx @ self.q
        x, y, bth = u[:, 0:1], u[:, 1:2], u[:, 2:]
        dx, dy = x - X0, y - Y0
        t = A * np.exp(a * dx ** 2 + b * dx * dy + c * dy ** 2)
        e = SCALE * t.sum(1) + 0.5 * (self.k * bth ** 2).sum(1) + self.cpl * (y * bth).sum(1)
        gx = SCALE * (t * (2 * a * dx + b * dy)).sum(1)
        gy = SCALE * (t * (b * dx + 2 * c * dy)).sum(1) + self.cpl * bth.sum(1)
        gb = self.k * bth + self.cpl * y
        g = np.concatenate([gx[:, None], gy[:, None], gb], axis=1) @ self.q.T
        return {"energy": e, "forces": -g}


def _run_mb24(warm):
    calc = MuellerBrown24()
    r = np.zeros(24); r[:2] = MIN_A[:2]
    p = np.zeros(24); p[:2] = MIN_B[:2]
    drv = GrowingStringDriver(["X"] * 8, r @ calc.q.T, p @ calc.q.T, calc,
                              gs_kw={"max_nodes": 9, "perp_thresh": 2e-2, "climb_rms": 5e-3, "climb_lanczos_rms": 5e-3, "climb_lanczos_warm_start": warm},
                              stopt_kw={"thresh": "gau_loose", "max_step": 0.05, "max_cycles": 900, "stop_in_when_full": 800})
    return drv, drv.run(), calc


def test_lanczos_warm_start_cuts_the_serial_depth_not_the_answer():
    """once the string climbs, every cycle runs a Lanczos recursion of SERIAL single-image gradients (reference defaults
    climb_lanczos=True, path_opt.py:179-182).  Started from last cycle's mode instead of the string tangent, the recursion keeps its stop rule
    (dl, max_cycles) and needs its minimum of two gradients per cycle; the guarded result (negative curvature, overlap with the tangent) leaves
    the optimisation where the cold recursion takes it."""
    cold_drv, cold, cold_calc = _run_mb24(False)
    warm_drv, warm, warm_calc = _run_mb24(True)
    assert cold.converged and warm.converged and cold.fully_grown and warm.fully_grown
    for drv, res, calc in ((cold_drv, cold, cold_calc), (warm_drv, warm, warm_calc)):
        assert drv.lanczos_calls >= 10 and res.timing["lanczos_evals"] == drv.lanczos_evals == calc.batches.count(1)
    per_cold, per_warm = cold_drv.lanczos_evals / cold_drv.lanczos_calls, warm_drv.lanczos_evals / warm_drv.lanczos_calls
    print(f"Lanczos gradients per climbing cycle: cold {per_cold:.2f} ({cold_drv.lanczos_calls} recursions), warm {per_warm:.2f} "
          f"({warm_drv.lanczos_warm_calls} kept, {warm_drv.lanczos_warm_rejected} rejected)")
    assert cold_drv.lanczos_warm_calls == 0 and per_cold >= 2.8
    assert warm_drv.lanczos_warm_calls >= warm_drv.lanczos_calls - 4 and per_warm <= 2.4      # (the first and every 10th recursion are cold)
    assert abs(warm.energies[warm.hei_index] - cold.energies[cold.hei_index]) < 1e-6 and warm.hei_index == cold.hei_index
    assert np.abs(warm.coords - cold.coords).max() < 2e-3
    assert abs(warm.cycles - cold.cycles) <= 5


def test_lanczos_warm_start_is_dropped_when_the_mode_leaves_the_path():
    """The guard: a warm-started recursion spans a tiny Krylov space and follows its eigenvector wherever it goes; a result with positive curvature
    (or, when asked for, little overlap with the string tangent) is rejected, the cold recursion runs, and nothing is remembered for the next
    cycle; every 10th recursion is cold anyway."""
    calc = MuellerBrown24()
    r = np.zeros(24); r[:2] = MIN_A[:2]
    p = np.zeros(24); p[:2] = MIN_B[:2]
    # climbing forced from the first fully grown cycle on: the HEI is far from the saddle, where the lowest mode is a soft bath direction
    drv = GrowingStringDriver(["X"] * 8, r @ calc.q.T, p @ calc.q.T, calc,
                              gs_kw={"max_nodes": 9, "perp_thresh": 1e9, "climb_rms": 1e9, "climb_lanczos_rms": 1e9, "climb_lanczos_warm_overlap": 0.5},
                              stopt_kw={"thresh": "gau_loose", "max_step": 0.05, "max_cycles": 40})
    res = drv.run()
    assert drv.lanczos_calls > 10
    assert drv.lanczos_warm_calls + drv.lanczos_warm_rejected <= drv.lanczos_calls
    kept = [lg for lg in drv.lanczos_log if lg[3]]
    assert all(w < 0.0 and ov >= 0.5 for w, ov, _, _ in kept)                            # what was trusted met both conditions
    cold_every_10th = [lg[3] for i, lg in enumerate(drv.lanczos_log) if i % 10 == 9]
    assert cold_every_10th and not any(cold_every_10th)
    assert np.isfinite(res.energies).all() and np.abs(res.coords).max() < 10.0          # the string did not run away along a bath mode
