"""GPU parity of umx_bond_changes (row f3) against the numpy restatement of bond_changes.py:142-187."""
import types

import numpy as np
import pytest

from oracle import bond_changes_oracle as O

pytestmark = pytest.mark.gpu


def _geoms(n, seed):
    from pdb2reaction_amd import synth
    from pdb2reaction_amd.uma_pysis import ANG2BOHR
    z, pos = synth.make_cluster(n, seed)
    prod = synth.make_product(pos, seed + 1)
    atoms = [synth.SYMBOLS[int(a)] for a in z]
    g1 = types.SimpleNamespace(atoms=atoms, coords3d=pos * ANG2BOHR)
    g2 = types.SimpleNamespace(atoms=atoms, coords3d=prod * ANG2BOHR)
    return g1, g2


@pytest.mark.parametrize("n", [2, 63, 64, 65, 500, 2000])
def test_matches_oracle(n):
    from pdb2reaction_amd import bond_changes as B
    g1, g2 = _geoms(max(n, 8), 5)
    g1.atoms, g2.atoms = g1.atoms[:n], g2.atoms[:n]
    g1.coords3d, g2.coords3d = g1.coords3d[:n], g2.coords3d[:n]
    res = B.compare_structures(g1, g2)
    _, cov = B.element_radii(g1.atoms)
    d1, d2, code = O.compare(g1.coords3d, g2.coords3d, cov)
    np.testing.assert_allclose(res.distances_1, d1, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(res.distances_2, d2, rtol=1e-13, atol=1e-13)
    assert res.formed_covalent == set(map(tuple, np.argwhere(code == 1).tolist()))
    assert res.broken_covalent == set(map(tuple, np.argwhere(code == 2).tolist()))
    assert all(i < j for i, j in res.formed_covalent | res.broken_covalent)


def test_known_formation_and_report():
    from pdb2reaction_amd import bond_changes as B
    from pdb2reaction_amd.uma_pysis import ANG2BOHR
    atoms = ["c", "O", "H"]                      # any case, as the reference capitalises
    r1 = np.array([[0, 0, 0], [2.5, 0, 0], [0, 1.09, 0]], float) * ANG2BOHR
    r2 = np.array([[0, 0, 0], [1.36, 0, 0], [0, 3.0, 0]], float) * ANG2BOHR
    g1 = types.SimpleNamespace(atoms=atoms, coords3d=r1)
    g2 = types.SimpleNamespace(atoms=atoms, coords3d=r2)
    res = B.compare_structures(g1, g2, device="cpu")      # device is advisory: always the GPU engine
    assert res.formed_covalent == {(0, 1)} and res.broken_covalent == {(0, 2)}
    txt = B.summarize_changes(g2, res)
    assert txt.splitlines() == ["Bond formed (1):", "  - C1-O2 : 2.500 Å --> 1.360 Å", "Bond broken (1):", "  - C1-H3 : 1.090 Å --> 3.000 Å"]
    same = B.compare_structures(g1, g1)
    assert B.summarize_changes(g1, same) == "Bond formed: None\nBond broken: None"


def test_errors():
    from pdb2reaction_amd import bond_changes as B
    g1 = types.SimpleNamespace(atoms=["C", "O"], coords3d=np.zeros((2, 3)))
    g2 = types.SimpleNamespace(atoms=["C", "N"], coords3d=np.zeros((2, 3)))
    with pytest.raises(AssertionError):
        B.compare_structures(g1, g2)
    g3 = types.SimpleNamespace(atoms=["C", "Qq"], coords3d=np.zeros((2, 3)))
    with pytest.raises(KeyError):
        B.compare_structures(g3, g3)
