"""CPU tests of the RFO half of ``_optimize_single`` (reference path_opt.py:464-518, RFO_KW opt.py:231-277)."""
import warnings

import numpy as np
import pytest

from pdb2reaction_amd.rfo import RFO_KW, RFOptimizer, bfgs_update, bofill_update, optimize_single, rs_rfo_step

QUIET = dict(gdiis=False, line_search=False, adapt_step_func=False)      # the pysisyphus features that have no counterpart here


class Morse:
    """A chain of N atoms with Morse bonds between neighbours + a weak harmonic tether of every atom to its place on the straight chain
    (a well-defined minimum, no free translations / rotations): analytic E and F, finite-difference H."""

    def __init__(self, n, seed=0):
        self.n = n
        rng = np.random.default_rng(seed)
        self.r0 = 2.0 + 0.2 * rng.random(n - 1)
        self.d, self.a, self.kt = 0.15, 1.1, 0.3
        self.ref = np.zeros((n, 3)); self.ref[:, 0] = 2.1 * np.arange(n)
        self.calls = {"f": 0, "h": 0}

    def _e(self, x):
        q = x.reshape(-1, 3)
        r = np.linalg.norm(q[1:] - q[:-1], axis=1)
        return float(np.sum(self.d * (1 - np.exp(-self.a * (r - self.r0))) ** 2) + 0.5 * self.kt * np.sum((q - self.ref) ** 2))

    def _g(self, x):
        q = x.reshape(-1, 3)
        d = q[1:] - q[:-1]
        r = np.linalg.norm(d, axis=1)
        ex = np.exp(-self.a * (r - self.r0))
        de = 2 * self.d * (1 - ex) * self.a * ex
        g = np.zeros_like(q)
        g[1:] += (de / r)[:, None] * d
        g[:-1] -= (de / r)[:, None] * d
        g += self.kt * (q - self.ref)
        return g.reshape(-1)

    def get_forces(self, elem, coords):
        self.calls["f"] += 1
        x = np.asarray(coords, dtype=float).reshape(-1)
        return {"energy": self._e(x), "forces": -self._g(x)}

    def get_hessian(self, elem, coords):
        self.calls["h"] += 1
        x = np.asarray(coords, dtype=float).reshape(-1)
        h = np.zeros((x.size, x.size))
        for k in range(x.size):
            e = np.zeros_like(x); e[k] = 1e-5
            h[:, k] = (self._g(x + e) - self._g(x - e)) / 2e-5
        return {"energy": self._e(x), "forces": -self._g(x), "hessian": 0.5 * (h + h.T)}


def start(n, seed=1):
    rng = np.random.default_rng(seed)
    q = np.zeros((n, 3))
    q[:, 0] = 2.1 * np.arange(n)
    return q + 0.15 * rng.standard_normal((n, 3))


def test_defaults_mirror_reference():
    assert RFO_KW["trust_radius"] == 0.10 and RFO_KW["trust_max"] == 0.10 and RFO_KW["hessian_update"] == "bfgs" and RFO_KW["hessian_init"] == "calc"
    assert RFO_KW["hessian_recalc"] == 200 and RFO_KW["max_micro_cycles"] == 50 and RFO_KW["thresh"] == "gau" and RFO_KW["max_cycles"] == 10000
    assert RFO_KW["gdiis"] is True and RFO_KW["gediis"] is False and RFO_KW["small_eigval_thresh"] == 1e-8


def test_rs_rfo_step_properties():
    rng = np.random.default_rng(2)
    m = rng.standard_normal((12, 12))
    hess = m @ m.T + 0.5 * np.eye(12)
    g = rng.standard_normal(12)
    h, v = np.linalg.eigh(hess)
    s, alpha, micro = rs_rfo_step(h, v, g, trust=10.0)
    # unrestricted: the lowest eigenvector of the augmented Hessian (alpha = 1)
    ah = np.block([[hess, g[:, None]], [g[None, :], np.zeros((1, 1))]])
    w, vec = np.linalg.eigh(ah)
    ref = vec[:-1, 0] / vec[-1, 0]
    assert alpha == 1.0 and np.allclose(s, ref, atol=1e-9) and float(g @ s) < 0
    # restricted: on the trust sphere, still a descent direction, tends to the (scaled) Newton direction for a tiny radius
    s2, alpha2, micro2 = rs_rfo_step(h, v, g, trust=0.05)
    assert alpha2 > 1.0 and micro2 > 1 and abs(np.linalg.norm(s2) - 0.05) < 1e-3 and float(g @ s2) < 0
    # a Hessian with a negative mode: RFO still goes downhill
    hneg = hess - 3.0 * np.outer(v[:, 0], v[:, 0]) * (h[0] + 1.0)
    h3, v3 = np.linalg.eigh(hneg)
    assert h3[0] < 0
    s3, _, _ = rs_rfo_step(h3, v3, g, trust=0.3)
    assert float(g @ s3) < 0 and np.linalg.norm(s3) <= 0.3 * (1 + 1e-6)
    # zero modes (|h| < small) are projected out
    hz = hess - np.outer(v[:, 0], v[:, 0]) * h[0]
    h4, v4 = np.linalg.eigh(hz)
    s4, _, _ = rs_rfo_step(h4, v4, g, trust=10.0)
    assert abs(float(s4 @ v4[:, np.argmin(np.abs(h4))])) < 1e-10
    assert np.array_equal(rs_rfo_step(h, v, np.zeros(12), 0.1)[0], np.zeros(12))


def test_hessian_updates_satisfy_the_secant_equation():
    rng = np.random.default_rng(3)
    m = rng.standard_normal((8, 8))
    hess = m @ m.T + np.eye(8)
    true = hess + 0.3 * np.diag(rng.random(8))
    s = rng.standard_normal(8) * 0.1
    y = true @ s
    for upd in (bfgs_update, bofill_update):
        h2 = upd(hess, s, y)
        assert np.allclose(h2 @ s, y, atol=1e-12) and np.allclose(h2, h2.T, atol=1e-12)
    assert bfgs_update(hess, s, -y) is hess                       # negative curvature pair: the BFGS model is kept


@pytest.mark.parametrize("init,update", [("calc", "bfgs"), ("unit", "bfgs"), ("calc", "bofill")])
def test_rfo_relaxes_a_morse_chain(init, update):
    calc = Morse(7)
    x0 = start(7)
    opt = RFOptimizer(calc, ["C"] * 7, x0, hessian_init=init, hessian_update=update, thresh="gau_tight", max_cycles=400, trust_max=0.3, trust_radius=0.3, **QUIET)
    res = opt.run()
    assert res["converged"], res["history"][-1]
    assert np.abs(res["forces"]).max() <= 1.5e-5 and res["energy"] < calc._e(x0.reshape(-1)) - 0.01
    assert res["n_hessian_calls"] == (1 if init == "calc" else 0) and res["n_force_calls"] == res["cycles"] + 1
    es = [h["energy"] for h in res["history"]]
    assert es[-1] <= min(es) + 1e-12                              # ends at the lowest energy seen
    if init == "calc":                                            # an exact initial Hessian beats the unit matrix by a wide margin
        unit = RFOptimizer(Morse(7), ["C"] * 7, x0, hessian_init="unit", thresh="gau_tight", max_cycles=400, trust_max=0.3, trust_radius=0.3, **QUIET).run()
        assert res["cycles"] < unit["cycles"]


def test_frozen_atoms_trust_radius_and_recalc():
    calc = Morse(6)
    x0 = start(6, seed=4)
    opt = RFOptimizer(calc, ["C"] * 6, x0, freeze=[0, 5], thresh="gau", max_cycles=300, hessian_recalc=5, **QUIET)
    res = opt.run()
    assert res["converged"] and np.array_equal(res["coords"][[0, 5]], x0[[0, 5]])                # frozen atoms never move
    assert max(h["step_norm"] for h in res["history"]) <= 0.10 * (1 + 1e-6)                        # reference default trust_max = 0.10
    assert res["n_hessian_calls"] == 1 + (res["cycles"] - 1) // 5 or res["n_hessian_calls"] == 1 + max(res["cycles"] - 1, 0) // 5
    act = np.ones(18, bool); act[:3] = False; act[15:] = False
    assert np.abs(res["forces"][act]).max() <= 4.5e-4
    # a calculator that returns the ACTIVE block of the Hessian (uma_pysis(return_partial_hessian=True)) works the same
    class Partial(Morse):
        def get_hessian(self, elem, coords):
            r = super().get_hessian(elem, coords)
            idx = np.flatnonzero(act)
            r["hessian"] = r["hessian"][np.ix_(idx, idx)]
            return r
    res2 = RFOptimizer(Partial(6), ["C"] * 6, x0, freeze=[0, 5], thresh="gau", max_cycles=300, hessian_recalc=5, **QUIET).run()
    assert res2["cycles"] == res["cycles"] and np.allclose(res2["coords"], res["coords"], atol=1e-10)


def test_unimplemented_keywords_and_dispatch():
    calc = Morse(4)
    x0 = start(4)
    with pytest.warns(RuntimeWarning, match="GDIIS"):
        RFOptimizer(calc, ["C"] * 4, x0)                           # the reference defaults switch GDIIS / line search on: said so, once
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        RFOptimizer(calc, ["C"] * 4, x0, **QUIET)
    for bad in ({"hessian_init": "lindh"}, {"hessian_update": "sr1"}, {"force_only": True}, {"thresh": "baker"}):
        with pytest.raises(NotImplementedError):
            RFOptimizer(calc, ["C"] * 4, x0, **QUIET, **bad)
    with pytest.raises(TypeError, match="unknown keyword"):
        RFOptimizer(calc, ["C"] * 4, x0, trust_radius_max=1.0)
    # _optimize_single dispatch: "lbfgs" -> L-BFGS, anything else -> RFO (path_opt.py:483-489)
    r_l = optimize_single(Morse(5), ["C"] * 5, start(5), "lbfgs", {"thresh": "gau", "max_cycles": 400, "out_dir": "x"})
    r_r = optimize_single(Morse(5), ["C"] * 5, start(5), "rfo", {"thresh": "gau", "max_cycles": 400, "out_dir": "x", **QUIET})
    assert r_l["converged"] and r_r["converged"] and r_l["n_hessian_calls"] == 0 and r_r["n_hessian_calls"] == 1
    assert abs(r_l["energy"] - r_r["energy"]) < 1e-4 and r_r["cycles"] <= r_l["cycles"]
