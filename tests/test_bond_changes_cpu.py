"""CPU checks of the bond-change host logic and its oracle (no GPU: compute calls are in test_gpu_bond_changes.py)."""
import numpy as np

from oracle import bond_changes_oracle as O


def test_oracle_hand_case():
    # C-O 2.5 -> 1.36 A forms; C-H 1.09 -> 3.0 A breaks (radii C .76, O .66, H .31; thresholds 1.2*0.95*(ri+rj))
    cov = np.array([0.76, 0.66, 0.31])
    r1 = np.array([[0, 0, 0], [2.5, 0, 0], [0, 1.09, 0]], float)
    r2 = np.array([[0, 0, 0], [1.36, 0, 0], [0, 3.0, 0]], float)
    d1, d2, code = O.compare(r1, r2, cov)
    assert code[0, 1] == 1 and code[0, 2] == 2 and code.sum() == 3
    assert abs(d1[0, 1] - 2.5) < 1e-15 and abs(d2[0, 2] - 3.0) < 1e-15
    # a change smaller than delta_fraction*T is ignored even if it crosses the threshold
    thr = 1.2 * 0.95 * (0.76 + 0.66)
    r1b = np.array([[0, 0, 0], [thr + 0.01, 0, 0]], float)
    r2b = np.array([[0, 0, 0], [thr - 0.01, 0, 0]], float)
    assert O.compare(r1b, r2b, cov[:2])[2].sum() == 0


def test_radii_and_report_format():
    import importlib
    B = importlib.import_module("pdb2reaction_amd.bond_changes")
    elems, cov = B.element_radii(["c", "CL", "h"], unit_scale=1.0)
    assert elems == ["C", "Cl", "H"] and np.allclose(cov, [0.76, 1.02, 0.31])
    res = B.BondChangeResult({(0, 2)}, set(), None, None)
    import types
    g = types.SimpleNamespace(atoms=["c", "o", "h"])
    assert B.summarize_changes(g, res, one_based=False) == "Bond formed (1):\n  - C0-H2\nBond broken: None"
