"""CPU tests of the wire formats / config precedence restated from the reference."""
import importlib

import numpy as np
import pytest

from pdb2reaction_amd import formats as F

U = importlib.import_module("pdb2reaction_amd.uma_pysis")


def test_trj_roundtrip_and_exact_text(tmp_path):
    syms = ["C", "H", "O"]
    imgs = [np.array([[0.0, 0.1, -0.2], [1.0, 1.123456789012345678, 2.0], [-3.5, 0.0, 1e-7]]), np.zeros((3, 3))]
    e = [-228.123456789012345, 1.5]
    p = tmp_path / "final_geometries.trj"
    F.write_trj_with_energy(syms, imgs, e, p)
    txt = p.read_text().splitlines()
    assert txt[0] == "3" and txt[1] == "-228.123456789012"            # f"{E:.12f}" (path_opt.py:283)
    assert txt[3] == "H 1.000000000000000 1.123456789012346 2.000000000000000"   # "{x:.15f}" (path_opt.py:285)
    assert txt[5] == "3" and txt[6] == "1.500000000000"
    assert F.read_energies_xyz(p) == [-228.123456789012, 1.5]
    s2, c2, comments = F.read_trj(p)
    assert s2 == syms and c2.shape == (2, 3, 3) and np.allclose(c2[0], imgs[0], atol=1e-15)
    assert comments[1] == "1.500000000000"


def test_reader_takes_first_decimal_number_only(tmp_path):
    p = tmp_path / "a.trj"
    p.write_text("1\nE = -1.25e-3 Hartree step 7\nH 0 0 0\n1\nenergy: 42\nH 0 0 1\n")
    assert F.read_energies_xyz(p) == [-1.25, 42.0]                    # exponents are not parsed (trj2fig.py:101)
    p.write_text("1\nno number here\nH 0 0 0\n")
    with pytest.raises(RuntimeError, match="Energy not found"):
        F.read_energies_xyz(p)
    p.write_text("not an xyz\n")
    with pytest.raises(RuntimeError, match="No energy data"):
        F.read_energies_xyz(p)


def test_hei_xyz(tmp_path):
    F.write_xyz(["N", "N"], [[0, 0, 0], [0, 0, 1.1]], tmp_path / "hei.xyz", energy_hartree=-109.5)
    assert (tmp_path / "hei.xyz").read_text().splitlines()[:2] == ["2", "-109.500000000000"]


def test_yaml_precedence_defaults_cli_yaml(tmp_path):
    calc_cfg = dict(U.CALC_KW)                                         # defaults
    calc_cfg["charge"] = -1                                            # <- CLI
    gs_cfg = {"max_nodes": 10, "climb": True, "nested": {"a": 1, "b": 2}}
    y = tmp_path / "cfg.yaml"
    y.write_text("calc:\n  charge: 2\n  spin: 3\ngs:\n  max_nodes: 14\n  nested:\n    b: 5\nsopt:\n  lbfgs:\n    thresh: gau\nlbfgs:\n  thresh: baker\n")
    cfg = F.load_yaml_dict(y)
    lbfgs = {"thresh": "gau_loose", "max_cycles": 100}
    F.apply_yaml_overrides(cfg, [(calc_cfg, (("calc",),)), (gs_cfg, (("gs",),)), (lbfgs, (("sopt", "lbfgs"), ("lbfgs",)))])
    assert calc_cfg["charge"] == 2 and calc_cfg["spin"] == 3 and calc_cfg["model"] == "uma-s-1p1"      # YAML wins, rest kept
    assert gs_cfg == {"max_nodes": 14, "climb": True, "nested": {"a": 1, "b": 5}}                        # deep merge
    assert lbfgs == {"thresh": "gau", "max_cycles": 100}                                                 # FIRST existing path wins
    assert F.load_yaml_dict(None) == {}
    y.write_text("- 1\n- 2\n")
    with pytest.raises(ValueError, match="mapping"):
        F.load_yaml_dict(y)
    # the merged calc section maps 1:1 onto the calculator constructor (path_opt.py:823)
    c = U.uma_pysis(**calc_cfg)
    assert c.charge == 2 and c.mult == 3
