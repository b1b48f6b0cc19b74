"""The consumer of the ``.trj`` files pinned to the REFERENCE's own functions (``trj2fig.py``).

``tests/golden/ref_energy_series.json`` holds what ``recompute_energies`` / ``transform_series`` / ``write_csv`` of
``pdb2reaction/trj2fig.py:112-205,287-303`` (``ast``-compiled by ``tools/make_reference_fixtures.py``) return: the re-scoring loop
with the reference's own ``uma_pysis.get_energy`` on ``tests/toy_core.ToyPairCore`` behind it.  Here the same files go through
``pdb2reaction_amd.formats``: ONE batched calculator call instead of a loop, and every number must agree bitwise."""
import importlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from toy_core import ToyPairCore
from pdb2reaction_amd import formats as F, synth

U = importlib.import_module("pdb2reaction_amd.uma_pysis")


@pytest.fixture(scope="module")
def fx():
    with open(os.path.join(GOLDEN, "ref_energy_series.json")) as f:
        return json.load(f)


def test_constants(fx):
    c = fx["constants"]
    assert (c["AU2KCALPERMOL"], c["ANG2BOHR"], c["EV2AU"]) == (F.AU2KCALPERMOL, U.ANG2BOHR, U.EV2AU)


def test_transform_series_matches_reference(fx):
    assert len(fx["transform_series"]) >= 150
    for c in fx["transform_series"]:
        what = f"{c['reference']!r} {c['unit']} reverse={c['reverse_x']} n={len(c['energies'])}"
        if "raises" in c:
            with pytest.raises(Exception) as ei:
                F.transform_series(c["energies"], c["reference"], c["unit"], c["reverse_x"])
            assert type(ei.value).__name__ == c["raises"] and str(ei.value) == c["message"], what
            continue
        v, lab, isd = F.transform_series(c["energies"], c["reference"], c["unit"], c["reverse_x"])
        assert v == c["values"] and all(isinstance(x, float) for x in v), what          # bitwise: same operation order
        assert lab == c["ylabel"] and isd is c["is_delta"], what


def test_write_energy_csv_matches_reference_bytes(fx, tmp_path):
    for i, c in enumerate(fx["write_csv"]):
        p = tmp_path / f"o{i}.csv"
        F.write_energy_csv(p, c["energies"], c["series"], c["unit"], c["is_delta"])
        assert p.read_bytes().decode("utf-8") == c["bytes"]


class _Recorder:
    """Stands where ``uma_pysis`` is looked up: records the constructor arguments, hands out a real calculator on the toy core."""

    real = U.uma_pysis

    def __init__(self, n_atoms, seed):
        self.n_atoms, self.seed, self.ctor, self.calc = n_atoms, seed, None, None

    def __call__(self, **kw):
        self.ctor = kw
        self.calc = self.real(**kw)
        self.calc._core = ToyPairCore(self.n_atoms, seed=self.seed)
        return self.calc


def test_recompute_energies_matches_reference_loop(fx, tmp_path, monkeypatch):
    for i, c in enumerate(fx["recompute_energies"]):
        p = tmp_path / f"t{i}.trj"
        p.write_text(c["text"])
        rec = _Recorder(c["n_atoms"], c["core_seed"])
        monkeypatch.setattr(U, "uma_pysis", rec)
        en = F.recompute_energies(p, c["charge"], c["multiplicity"])
        assert rec.ctor == c["ctor"]                                   # charge or 0, multiplicity or 1 -- as the reference builds it
        assert en == c["energies"] and all(isinstance(e, float) for e in en)
        assert rec.calc._core is None                                  # a calculator it made itself is closed again
        core = ToyPairCore(c["n_atoms"], seed=c["core_seed"])
        monkeypatch.undo()
        calc = U.uma_pysis()
        calc._core = core
        for mb in (None, 1, 2, 100):
            assert F.recompute_energies(p, None, None, calc=calc, max_batch=mb) == c["energies"]
        assert calc._core is core                                      # a caller's calculator stays open
        n_frames = len(c["energies"])
        assert core.calls == 4 * n_frames


def test_recompute_energies_rejects_what_the_reference_cannot_score(tmp_path):
    p = tmp_path / "e.trj"
    p.write_text("")
    with pytest.raises(RuntimeError):
        F.recompute_energies(p, 0, 1, calc=object())
    p.write_text("1\nx\nH 0 0 0\n1\ny\nC 0 0 0\n")                     # the atoms change between frames
    with pytest.raises(ValueError):
        F.recompute_energies(p, 0, 1, calc=object())


@pytest.mark.gpu
def test_recompute_energies_one_batched_call_on_the_engine(tmp_path):
    """On the HIP engine: all frames in ONE `umx_energy_forces` call, per frame bitwise what `get_energy` gives frame by frame (the
    reference's loop), and the profile derived from it."""
    z, imgs, _ = synth.make_images(60, 5, seed=21)
    sym = [synth.SYMBOLS[int(q)] for q in z]
    p = tmp_path / "path.trj"
    F.write_trj_with_energy(sym, [np.asarray(x, dtype=np.float64) for x in imgs], [0.0] * len(imgs), p)
    with U.uma_pysis(model="synthetic") as calc:
        en = F.recompute_energies(p, 0, 1, calc=calc)
        _, coords, _ = F.read_trj(p)
        one_by_one = [float(calc.get_energy(sym, (c * U.ANG2BOHR).reshape(-1))["energy"]) for c in coords]
        assert en == one_by_one
        assert F.recompute_energies(p, 0, 1, calc=calc, max_batch=2) == en
    vals, lab, isd = F.transform_series(en, "init", "kcal", False)
    assert isd and vals[0] == 0.0 and lab.endswith("(kcal/mol)") and len(vals) == 5
