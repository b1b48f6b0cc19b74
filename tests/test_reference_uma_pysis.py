"""The calculator boundary pinned to the REFERENCE's own methods (rows a1 / a2 / a6 / a10 of SURVEY.md 8a).

``tests/golden/ref_uma_pysis_methods.json`` holds what the bodies of ``uma_pysis._au_energy / _au_forces / _au_hessian /
_active_and_frozen_dof_idx / _zero_frozen_forces_ev / _apply_analytical_active_trim / _build_fd_hessian_gpu / get_energy /
get_forces / get_hessian`` (reference ``pdb2reaction/uma_pysis.py:502-780``, ``ast``-compiled by
``tools/make_reference_fixtures.py`` in the build container) return when the core behind them is ``tests/toy_core.ToyPairCore``.
Here the SAME core sits behind ``pdb2reaction_amd.uma_pysis.uma_pysis`` and every output must agree: exactly for containers,
dtypes, shapes, zeros and index lists, and BITWISE for the numbers too -- the operation order of the unit conversions, of the
central difference ``-(F+ - F-) / (2 h)`` in the Hessian dtype and of the symmetrisation is the reference's, and the toy core
quantises positions to float32 like ``AtomicData.pos``, so the reference's finite-difference noise floor is in the fixture.
The documented tolerance for the float64 numbers is 1e-13 relative; the test asserts equality first and
reports the worst deviation if that ever fails.
"""
import importlib
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from toy_core import ToyPairCore, toy_geometry
from pdb2reaction_amd import hessian as H

U = importlib.import_module("pdb2reaction_amd.uma_pysis")


@pytest.fixture(scope="module")
def fx():
    with open(os.path.join(GOLDEN, "ref_uma_pysis_methods.json")) as f:
        return json.load(f)


def close(got, want, what=""):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, what
    if not np.array_equal(got, want):
        np.testing.assert_allclose(got, want, rtol=1e-13, atol=0, err_msg=what)


def check_hessian(h, rec, what=""):
    if rec["container"] == "torch":
        assert isinstance(h, torch.Tensor) and str(h.dtype).replace("torch.", "") == rec["dtype"], what
        assert not h.requires_grad
        v = h.detach().cpu().to(torch.float64).numpy()
    else:
        assert isinstance(h, np.ndarray) and str(h.dtype) == rec["dtype"], what
        v = h
    assert list(v.shape) == rec["shape"], what
    want = np.asarray(rec["values"], dtype=np.float64)
    assert np.array_equal(v == 0.0, want == 0.0), f"{what}: zero pattern (frozen columns) differs"
    close(v, want, what)


def test_constants(fx):
    c = fx["constants"]
    assert (c["EV2AU"], c["F_EVAA_2_AU"], c["H_EVAA_2_AU"]) == (U.EV2AU, U.F_EVAA_2_AU, U.H_EVAA_2_AU)
    assert (c["ANG2BOHR"], c["BOHR2ANG"]) == (U.ANG2BOHR, U.BOHR2ANG)


def test_unit_helpers_match_reference(fx):
    """_au_energy / _au_forces (:506-513): E * EV2AU; F -> float64, * F_EVAA_2_AU, flattened."""
    for c in fx["au_energy"]:
        assert c["e_ev"] * U.EV2AU == c["e_au"]
    for c in fx["au_forces"]:
        f = np.asarray(c["f_ev"], dtype=c["dtype"])
        out = (np.asarray(f, dtype=np.float64) * U.F_EVAA_2_AU).reshape(-1)           # the expression of get_forces / get_forces_batch
        assert str(out.dtype) == c["out_dtype"] and list(out.shape) == c["out_shape"]
        assert np.array_equal(out, np.asarray(c["f_au"]))


def test_freeze_helpers_match_reference(fx):
    """_active_and_frozen_dof_idx (:554-559), freeze_atoms normalisation (:497), _zero_frozen_forces_ev (:561-567)."""
    for c in fx["dof_idx"]:
        calc = U.uma_pysis(freeze_atoms=c["freeze_atoms"])
        assert calc.freeze_atoms == c["normalised"]
        active, dead = H.dof_partition(c["n_atoms"], calc.freeze_atoms)
        assert active == c["active_dof"] and dead == c["frozen_dof"]
        assert sorted(set(k // 3 for k in active)) == c["active_atoms"]
    for c in fx["zero_frozen"]:
        f = np.asarray(c["f"], dtype=np.float32)
        out = H.mask_frozen(f, sorted(set(c["freeze_atoms"])))
        assert (out is f) == c["same_object"] and out.dtype == np.float32
        assert np.array_equal(out.astype(np.float64), np.asarray(c["out"]))
        assert (H.mask_frozen(None, c["freeze_atoms"]) is None) == c["none_passthrough"]


def test_au_hessian_matches_reference(fx):
    """_au_hessian (:515-551): view (3n,3n), 0.5 (H + H^T), * H_EVAA_2_AU, -> float64 if hessian_double, torch or numpy."""
    for c in fx["au_hessian"]:
        h = torch.as_tensor(np.asarray(c["h"]), dtype=getattr(torch, c["in_dtype"]))
        out = H.hessian_to_au(h, double=c["hessian_double"], as_torch=c["out_hess_torch"])
        check_hessian(out, c["out"], str({k: c[k] for k in ("in_dtype", "hessian_double", "out_hess_torch")}))


def test_analytical_trim_matches_reference(fx):
    """_apply_analytical_active_trim (:569-592): active block, or full size with the frozen COLUMNS zeroed."""
    for c in fx["analytical_trim"]:
        h = torch.as_tensor(np.asarray(c["h"]), dtype=torch.float32)
        keep = h.clone()
        out = H.active_trim(h, c["freeze_atoms"], partial=c["return_partial_hessian"])
        check_hessian(out, c["out"], str((c["freeze_atoms"], c["return_partial_hessian"])))
        assert torch.equal(h, keep)                               # the caller's tensor is not modified


def test_get_energy_forces_hessian_match_reference(fx, monkeypatch):
    """get_energy / get_forces / get_hessian (:689-780) incl. _build_fd_hessian_gpu (:595-686) for every combination of
    freeze_atoms x return_partial_hessian x hessian_double x out_hess_torch, the mode dispatch (None / '' / unknown /
    padded spelling -> FD; Analytical only with an exposed model and workers == 1) and flat or (N,3) Bohr input."""
    assert len(fx["api"]) >= 32
    seen_modes = set()
    for vi, c in enumerate(fx["api"]):
        core = ToyPairCore(c["n_atoms"], seed=c["core_seed"], parallel_predict=c["workers"] > 1, has_torch_model=c["has_torch_model"])
        calc = U.uma_pysis(model="synthetic", freeze_atoms=c["freeze_atoms"], return_partial_hessian=c["return_partial_hessian"],
                           hessian_double=c["hessian_double"], out_hess_torch=c["out_hess_torch"], hessian_calc_mode=c["hessian_calc_mode"],
                           workers=c["workers"])
        calc._core = core
        coords = np.asarray(c["coords_bohr"], dtype=np.float64)
        assert np.array_equal(coords.reshape(-1, 3) * U.BOHR2ANG, (toy_geometry(c["n_atoms"], c["geometry_seed"]) * U.ANG2BOHR) * U.BOHR2ANG)
        what = f"variant {vi}: " + str({k: c[k] for k in ("freeze_atoms", "return_partial_hessian", "hessian_double", "out_hess_torch",
                                                         "hessian_calc_mode", "workers", "has_torch_model")})
        r = calc.get_energy(c["elem"], coords.tolist())
        assert sorted(r) == c["get_energy"]["keys"] and isinstance(r["energy"], float)
        assert r["energy"] == c["get_energy"]["energy"], what
        r = calc.get_forces(c["elem"], coords)
        assert sorted(r) == c["get_forces"]["keys"] and r["energy"] == c["get_forces"]["energy"]
        assert str(r["forces"].dtype) == c["get_forces"]["dtype"] and list(r["forces"].shape) == c["get_forces"]["shape"]
        close(r["forces"], c["get_forces"]["forces"], what)
        for a in sorted(set(c["freeze_atoms"])):
            assert not r["forces"].reshape(-1, 3)[a].any()
        # the batched entry points return per image exactly what the reference's single-image methods return
        rb = calc.get_forces_batch(c["elem"], np.stack([coords.reshape(-1), coords.reshape(-1)]))
        close(rb["forces"][1], c["get_forces"]["forces"], what)
        assert rb["energy"][0] == c["get_forces"]["energy"]
        calls0 = core.calls
        monkeypatch.setattr(U, "FD_BATCH", 1 + 2 * (vi % 5))      # the batch size of the displaced geometries must not matter
        r = calc.get_hessian(c["elem"], coords)
        assert sorted(r) == c["get_hessian"]["keys"] and r["energy"] == c["get_hessian"]["energy"]
        close(r["forces"], c["get_hessian"]["forces"], what)
        check_hessian(r["hessian"], c["get_hessian"]["hessian"], what)
        # same number of model evaluations as the reference: 1 + 2 per active DOF (FD) or exactly 1 (analytical)
        assert core.calls - calls0 == c["get_hessian"]["reference_core_calls"], what
        seen_modes.add((str(c["hessian_calc_mode"]).strip().lower(), c["get_hessian"]["reference_core_calls"] == 1))
    assert ("analytical", True) in seen_modes and ("analytical", False) in seen_modes and ("none", False) in seen_modes


def test_fd_displacements_are_the_references(fx):
    """The reference displaces the float64 Angstrom coordinate of ONE DOF by +-1e-3 and hands the float64 array to the core
    (:652-664), which then quantises it; the batched route must hand the core the very same arrays."""
    c = fx["api"][0]
    core = ToyPairCore(c["n_atoms"], seed=c["core_seed"], has_torch_model=False)
    calc = U.uma_pysis(model="synthetic", freeze_atoms=[1])
    calc._core = core
    x = np.asarray(c["coords_bohr"], dtype=np.float64).reshape(-1, 3) * U.BOHR2ANG
    calc.get_hessian(c["elem"], c["coords_bohr"])
    assert np.array_equal(core.seen[0], x)
    k = 0
    for a in range(c["n_atoms"]):
        if a == 1:
            continue
        for d in range(3):
            for sgn in (+1.0, -1.0):
                k += 1
                want = x.copy()
                want[a, d] = x[a, d] + sgn * 1.0e-3
                assert np.array_equal(core.seen[k], want), (a, d, sgn)
    assert len(core.seen) == k + 1


# ---- rows f2 / f3: the .trj writer of the DMF path and compare_structures, pinned the same way -------------------------------
def test_ase_trj_writer_matches_reference(fx, tmp_path):
    """_write_ase_trj_with_energy (path_opt.py:276-290): exact text."""
    from pdb2reaction_amd import formats as F

    assert len(fx["write_ase_trj"]) >= 3
    for c in fx["write_ase_trj"]:
        p = tmp_path / "dmf.trj"
        F.write_trj_with_energy(c["symbols"], [np.asarray(x) for x in c["images_ang"]], c["energies_hartree"], p)
        assert p.read_text() == c["text"]
        assert F.read_energies_xyz(p) == [float(f"{e:.12f}") for e in c["energies_hartree"]]


def _check_bond_case(c, d1, d2, formed, broken):
    n = len(c["atoms"])
    off = ~np.eye(n, dtype=bool)
    want1, want2 = np.asarray(c["d1"]).reshape(n, n), np.asarray(c["d2"]).reshape(n, n)
    # torch.cdist goes through |x|^2 + |y|^2 - 2 x.y for more than 25 points: absolute 1e-13-level noise off the diagonal and
    # sqrt(rounding) ~ 1e-7 ON it, where the direct difference is exactly 0
    np.testing.assert_allclose(np.asarray(d1)[off], want1[off], rtol=0, atol=1e-11)
    np.testing.assert_allclose(np.asarray(d2)[off], want2[off], rtol=0, atol=1e-11)
    assert np.abs(np.diag(want1)).max() < 1e-6 and not np.diag(np.asarray(d1)).any()
    assert sorted(map(list, formed)) == c["formed"] and sorted(map(list, broken)) == c["broken"]


def test_bond_change_oracle_matches_reference_compare_structures(fx):
    """oracle/bond_changes_oracle.py (the checker of the HIP kernel) against the reference's own compare_structures."""
    from oracle import bond_changes_oracle as O
    from pdb2reaction_amd import bond_changes as BC

    assert sum(len(c["formed"]) + len(c["broken"]) for c in fx["compare_structures"]) >= 6      # the cases do contain events
    for c in fx["compare_structures"]:
        _, cov = BC.element_radii(c["atoms"], 1.0, c["radii"])
        d1, d2, code = O.compare(np.asarray(c["r1"]), np.asarray(c["r2"]), cov, **c["kwargs"])
        _check_bond_case(c, d1, d2, np.argwhere(code == 1).tolist(), np.argwhere(code == 2).tolist())
    assert fx["compare_structures_mismatch"] == {"raises": "AssertionError", "message": "Atom types and ordering must be identical."}


@pytest.mark.gpu
def test_hip_bond_changes_match_reference_compare_structures(fx):
    """pdb2reaction_amd.bond_changes.compare_structures (HIP kernel k_bond_changes behind umx_bond_changes) against the
    recorded outputs of the reference's compare_structures (bond_changes.py:142-187) on the same inputs and radii."""
    from types import SimpleNamespace

    from pdb2reaction_amd import bond_changes as BC

    for c in fx["compare_structures"]:
        g1 = SimpleNamespace(atoms=c["atoms"], coords3d=np.asarray(c["r1"]))
        g2 = SimpleNamespace(atoms=list(c["atoms"]), coords3d=np.asarray(c["r2"]))
        res = BC.compare_structures(g1, g2, radii=c["radii"], unit_scale=1.0, **c["kwargs"])
        _check_bond_case(c, res.distances_1, res.distances_2, res.formed_covalent, res.broken_covalent)
    with pytest.raises(AssertionError, match="Atom types and ordering must be identical."):
        BC.compare_structures(SimpleNamespace(atoms=["H", "C"], coords3d=np.zeros((2, 3))),
                              SimpleNamespace(atoms=["C", "H"], coords3d=np.zeros((2, 3))))


def test_harmonic_bias_wrapper_matches_reference_class(fx):
    """opt.py:286-343, the whole ``HarmonicBiasCalculator`` as written, over the reference's own get_forces / get_energy on the toy
    core: restraint energy and forces added to the base calculator's, the two tuple-returning conveniences, ``set_pairs`` element
    types, attribute forwarding, and the number of base evaluations -- plus the batched entry this build adds."""
    from pdb2reaction_amd import prestep as PS

    assert len(fx["harmonic_bias_wrapper"]) >= 2
    for c in fx["harmonic_bias_wrapper"]:
        core = ToyPairCore(c["n_atoms"], seed=c["core_seed"])
        base = U.uma_pysis(freeze_atoms=c["freeze_atoms"])
        base._core = core
        wb = PS.HarmonicBias(base, k=c["k_ev_ang2"])
        wb.set_pairs([(np.int64(i), j, np.float32(t)) for i, j, t in c["pairs_in"]])
        assert [list(p) for p in wb._pairs] == c["pairs_stored"] and [type(v).__name__ for v in wb._pairs[0]] == c["pair_types"]
        assert wb.k_au_bohr2 == c["k_au_bohr2"]
        x = np.asarray(c["coords_bohr"])
        el = ["C"] * c["n_atoms"]
        rf = wb.get_forces(el, x.reshape(-1))
        close(rf["energy"], c["get_forces"]["energy"]); close(rf["forces"], c["get_forces"]["forces"])
        assert isinstance(rf["energy"], float) and rf["forces"].shape == (3 * c["n_atoms"],)
        close(wb.get_energy(el, x)["energy"], c["get_energy"])
        e2, f2 = wb.get_energy_and_forces(el, x)
        close(e2, c["energy_and_forces"][0]); close(f2, c["energy_and_forces"][1])
        e3, g3 = wb.get_energy_and_gradient(el, x.reshape(-1))
        close(e3, c["energy_and_gradient"][0]); close(g3, c["energy_and_gradient"][1])
        assert list(wb.freeze_atoms) == c["forwarded_freeze_atoms"] and core.calls == c["base_calls"]
        rb = wb.get_forces_batch(el, np.stack([x, x + 0.01]))                       # this build's batched entry: image 0 is the same number
        close(rb["energy"][0], c["get_forces"]["energy"]); close(rb["forces"][0], c["get_forces"]["forces"])
