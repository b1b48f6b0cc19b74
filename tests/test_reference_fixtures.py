"""Host-side restatements pinned to the REFERENCE's own functions.

tests/golden/ref_host_functions.json holds input -> output vectors produced by executing the reference's pure helper
functions in the build container (tools/make_reference_fixtures.py; the fixture is data, no reference source).  Each test
feeds the recorded inputs to this repo's counterpart and compares with the recorded outputs: exact for integer / text /
structural results, 1e-13 relative for float64 arithmetic whose operation order differs (vectorised bias, SVD sign
conventions are identical because both use numpy.linalg.svd on the same matrix)."""
import copy
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import GOLDEN
from pdb2reaction_amd import bond_changes as BC, formats as F, prestep as PS
from pdb2reaction_amd._calculator_base import ANG2BOHR, BOHR2ANG
from pdb2reaction_amd.hessian import EV_PER_ANG2_TO_AU
from pdb2reaction_amd.string import select_hei_index


@pytest.fixture(scope="module")
def fx():
    with open(os.path.join(GOLDEN, "ref_host_functions.json")) as f:
        return json.load(f)


def test_constants_are_the_ones_the_fixture_was_made_with(fx):
    c = fx["constants"]
    assert c["ANG2BOHR"] == ANG2BOHR and c["BOHR2ANG"] == BOHR2ANG and c["H_EVAA_2_AU"] == EV_PER_ANG2_TO_AU


def test_select_hei_index_matches_reference(fx):
    assert len(fx["select_hei_index"]) >= 40
    for case in fx["select_hei_index"]:
        assert select_hei_index(case["energies"]) == case["index"], case["energies"]


def test_kabsch_matches_reference(fx):
    for case in fx["kabsch_R_t"]:
        R, t = PS.kabsch_R_t(np.array(case["P"]), np.array(case["Q"]))
        np.testing.assert_allclose(R, case["R"], rtol=0, atol=1e-14)
        np.testing.assert_allclose(t, case["t"], rtol=0, atol=1e-13)
        assert np.linalg.det(R) > 0.999                                     # proper rotation, reflection case included


def test_rotation_helpers_match_reference(fx):
    for case in fx["rodrigues"]:
        np.testing.assert_allclose(PS._rodrigues(np.array(case["axis"]), case["theta"]), case["R"], rtol=0, atol=1e-15)
    for case in fx["rotation_align_vectors"]:
        np.testing.assert_allclose(PS._rotation_a_to_b(np.array(case["a"]), np.array(case["b"])), case["R"], rtol=0, atol=1e-15)
    for case in fx["rmsd"]:
        assert PS.rmsd_ang(np.array(case["A"]), np.array(case["B"])) == pytest.approx(case["rmsd_ang"], rel=1e-15)


def test_harmonic_bias_matches_reference(fx):
    for case in fx["harmonic_bias"]:
        hb = PS.HarmonicBias(base_calc=None, k=case["k_ev_ang2"], pairs=[tuple(p) for p in case["pairs"]])
        x = np.array(case["coords_bohr"])
        e, f = hb._bias(x[None])
        assert float(e[0]) == pytest.approx(case["energy"], rel=1e-13)
        np.testing.assert_allclose(f[0].reshape(-1), case["forces"], rtol=1e-13, atol=1e-16)
        # and through the calculator protocol on top of a base calculator that returns zeros
        base = SimpleNamespace(get_forces=lambda elem, c: {"energy": 0.0, "forces": np.zeros(x.size)},
                               get_energy=lambda elem, c: {"energy": 0.0})
        hb2 = PS.HarmonicBias(base, k=case["k_ev_ang2"], pairs=[tuple(p) for p in case["pairs"]])
        r = hb2.get_forces(["X"] * len(x), x.reshape(-1))
        assert r["energy"] == pytest.approx(case["energy"], rel=1e-13)
        np.testing.assert_allclose(r["forces"], case["forces"], rtol=1e-13, atol=1e-16)


def test_bond_change_report_text_matches_reference(fx):
    for case in fx["summarize_changes"]:
        res = BC.BondChangeResult(formed_covalent={tuple(p) for p in case["formed"]}, broken_covalent={tuple(p) for p in case["broken"]},
                                  distances_1=None if case["d1"] is None else np.array(case["d1"]),
                                  distances_2=None if case["d2"] is None else np.array(case["d2"]))
        assert BC.summarize_changes(SimpleNamespace(atoms=case["atoms"]), res, case["one_based"]) == case["text"]


def test_energy_reader_matches_reference(fx, tmp_path):
    p = tmp_path / "a.trj"
    for case in fx["read_energies_xyz"]:
        p.write_text(case["text"])
        if "raises" in case:
            with pytest.raises(RuntimeError) as ei:
                F.read_energies_xyz(p)
            assert case["raises"] == "RuntimeError"
            assert str(ei.value).replace(str(p), "<path>") == case["message"]
        else:
            assert F.read_energies_xyz(p) == case["energies"]


def test_yaml_helpers_match_reference(fx, tmp_path):
    for case in fx["deep_update"]:
        dst = copy.deepcopy(case["dst"])
        out = F.deep_update(dst, copy.deepcopy(case["src"]))
        assert out is dst and dst == case["result"]
    for case in fx["apply_yaml_overrides"]:
        targets = [(copy.deepcopy(t["before"]), tuple(tuple(p) for p in t["paths"])) for t in case["targets"]]
        F.apply_yaml_overrides(case["yaml"], targets)
        for (got, _), t in zip(targets, case["targets"]):
            assert got == t["after"], t
    p = tmp_path / "c.yaml"
    for case in fx["load_yaml_dict"]:
        if case["text"] is None:
            assert F.load_yaml_dict(None) == case["data"]
            continue
        p.write_text(case["text"])
        if "raises" in case:
            with pytest.raises(ValueError) as ei:
                F.load_yaml_dict(p)
            assert case["raises"] == "ValueError" and str(ei.value) == case["message"]
        else:
            assert F.load_yaml_dict(p) == case["data"]


def test_summary_yaml_matches_reference(fx, tmp_path):
    """summary.yaml (path_search.py:2762-2786): the bond-change block builder and the exact YAML text."""
    for case in fx["bond_changes_block"]:
        got = F.bond_changes_block(case["text"])
        assert got == case["result"], case["text"]
    for case in fx["summary_yaml"]:
        d = F.summary_dict(case["out_dir"], case["n_images"], case["segments"], case["energy_diagram"])
        text = F.write_summary_yaml(tmp_path / "summary.yaml", d)
        assert text == case["text"]
        assert (tmp_path / "summary.yaml").read_text(encoding="utf-8") == case["text"]
    b, dlt = F.barrier_and_delta_kcal([-1.0, -0.98, -1.01])
    assert b == pytest.approx(0.02 * 627.509474, rel=1e-7) and dlt == pytest.approx(-0.01 * 627.509474, rel=1e-7)


def test_default_settings_match_reference(fx):
    """uma_pysis.py:132-165 (CALC_KW / GEOM_KW_DEFAULT), the constructor's keyword-only signature (:432-452), path_opt.py:168-200
    (GS_KW / STOPT_KW) and the L-BFGS memory / damping defaults (opt.py:222-246): recorded VALUES of the reference's dicts."""
    import importlib
    import inspect

    from pdb2reaction_amd import gsm, lbfgs
    import pdb2reaction_amd as pkg

    U = importlib.import_module("pdb2reaction_amd.uma_pysis")
    d = fx["defaults"]
    assert U.CALC_KW == d["CALC_KW"] and list(U.CALC_KW) == list(d["CALC_KW"])          # same keys in the same order, same values
    assert U.GEOM_KW_DEFAULT == d["GEOM_KW_DEFAULT"] and pkg.CALC_KW is U.CALC_KW
    sig = inspect.signature(U.uma_pysis.__init__)
    kw_only = {n: p.default for n, p in sig.parameters.items() if p.kind is inspect.Parameter.KEYWORD_ONLY}
    ref_kw = d["uma_pysis.__init__"]["keyword_only"]
    assert {k: kw_only[k] for k in ref_kw} == ref_kw                                     # every reference keyword, same default
    assert [n for n, p in sig.parameters.items() if p.kind is inspect.Parameter.POSITIONAL_OR_KEYWORD] == d["uma_pysis.__init__"]["positional"]
    assert [n for n, p in sig.parameters.items() if p.kind is inspect.Parameter.VAR_KEYWORD] == [d["uma_pysis.__init__"]["var_keyword"]]
    extra = set(kw_only) - set(ref_kw)
    assert extra <= {"precision"}, extra                                                 # the one keyword this build adds
    assert gsm.GS_KW == d["GS_KW"]
    assert {k: gsm.STOPT_KW[k] for k in d["STOPT_KW"]} == d["STOPT_KW"]                  # + max_step / thresh, which the reference merges in from opt:
    assert set(gsm.STOPT_KW) - set(d["STOPT_KW"]) == {"max_step", "thresh"}
    lb = inspect.signature(lbfgs.BatchedLBFGS.__init__).parameters
    assert lb["keep_last"].default == d["LBFGS_KW"]["keep_last"] and lb["beta"].default == d["LBFGS_KW"]["beta"]
    assert lb["thresh"].default == d["OPT_BASE_KW"]["thresh"]
    # RFO_KW (opt.py:231-277, on top of OPT_BASE_KW): every key, every value, in the reference's order
    from pdb2reaction_amd import rfo
    assert rfo.RFO_KW == d["RFO_KW"] and list(rfo.RFO_KW) == list(d["RFO_KW"])


def test_pair_alignment_matches_reference(fx):
    """align_second_to_first_kabsch_inplace + _freeze_union (align_freeze_atoms.py:253-387): no anchors, one, two, two on ONE point
    (degenerate axis -> Kabsch reported on all atoms), many, out-of-range indices; the union comes from both structures."""
    from pdb2reaction_amd import prestep as P

    assert len(fx["align_pair"]) >= 10
    modes = set()
    for c in fx["align_pair"]:
        n = len(c["ref_bohr"])
        assert P.freeze_union(c["freeze_ref"], c["freeze_mob"], n) == c["union"]
        assert P.freeze_union(c["freeze_ref"], c["freeze_mob"]) == c["union_unbounded"]
        out, rep = P.align_second_to_first(np.asarray(c["ref_bohr"]), np.asarray(c["mob_bohr"]), c["union"])
        want = c["report"]
        assert rep["mode"] == want["mode"] and rep["n_used"] == want["n_used"] and type(rep["n_used"]) is int
        np.testing.assert_allclose(out, np.asarray(c["aligned_bohr"]), rtol=0, atol=1e-12)
        np.testing.assert_allclose([rep["before_A"], rep["after_A"]], [want["before_A"], want["after_A"]], rtol=1e-12, atol=1e-13)
        assert c["freeze_during_write"] == [] and c["freeze_after"] == sorted(c["freeze_mob"], key=c["freeze_mob"].index)   # (the fixture's own sanity)
        modes.add((want["mode"], want["n_used"] <= 2))
    assert {("kabsch", False), ("one_anchor", True), ("two_anchor", True), ("kabsch", True)} <= modes
    with pytest.raises(ValueError) as ei:
        P.align_second_to_first(np.zeros((3, 3)), np.zeros((4, 3)), [])
    assert type(ei.value).__name__ == fx["align_pair_mismatch"]["raises"] and str(ei.value) == fx["align_pair_mismatch"]["message"]
