"""CPU test of the path-opt composition (reference path_opt.py GSM branch: preopt -> align -> GSM -> final_geometries.trj -> hei.xyz)."""
import numpy as np

from pdb2reaction_amd import formats
from pdb2reaction_amd._calculator_base import ANG2BOHR
from pdb2reaction_amd.path_opt import optimize_path_gsm
from pdb2reaction_amd.string import select_hei_index


class DoubleWell:
    """N atoms on springs (a rigid-ish frame) + one coordinate of atom 0 in a double well: two minima joined by a barrier.
    Batched interface of uma_pysis (Bohr in, Hartree out)."""

    def __init__(self, ref_bohr):
        self.ref = np.asarray(ref_bohr, dtype=float)
        self.k, self.a, self.h = 0.4, 0.6, 0.02
        self.batches = []
        self.freeze_atoms = []

    def _one(self, x):
        q = x.reshape(-1, 3)
        d = q - self.ref
        e = 0.5 * self.k * np.sum(d[1:] ** 2) + 0.5 * self.k * np.sum(d[0, 1:] ** 2)
        g = np.zeros_like(q)
        g[1:] = self.k * d[1:]
        g[0, 1:] = self.k * d[0, 1:]
        u = d[0, 0]
        e += self.h * ((u / self.a) ** 2 - 1.0) ** 2
        g[0, 0] = self.h * 4.0 * ((u / self.a) ** 2 - 1.0) * u / self.a ** 2
        return e, -g.reshape(-1)

    def get_forces_batch(self, elem, coords):
        c = np.asarray(coords, dtype=float).reshape(len(coords), -1)
        self.batches.append(len(c))
        out = [self._one(x) for x in c]
        return {"energy": np.array([o[0] for o in out]), "forces": np.stack([o[1] for o in out])}

    def get_forces(self, elem, coords):
        e, f = self._one(np.asarray(coords, dtype=float).reshape(-1))
        return {"energy": e, "forces": f}

    def get_energy(self, elem, coords):
        return {"energy": self._one(np.asarray(coords, dtype=float).reshape(-1))[0]}


def test_path_opt_gsm_flow(tmp_path):
    rng = np.random.default_rng(0)
    n = 6
    ref_ang = rng.uniform(-2.0, 2.0, (n, 3))
    calc = DoubleWell(ref_ang * ANG2BOHR)
    elem = ["c", "H", "h", "O", "N", "H"]
    r_ang, p_ang = ref_ang.copy(), ref_ang.copy()
    r_ang[0, 0] -= calc.a / ANG2BOHR
    p_ang[0, 0] += calc.a / ANG2BOHR
    r_ang += 0.02 * rng.standard_normal((n, 3))                  # endpoints slightly off their minima: the pre-optimisation has work to do
    p_ang += 0.02 * rng.standard_normal((n, 3))
    logs = []
    res = optimize_path_gsm(elem, r_ang, p_ang, calc=calc, max_nodes=7, max_cycles=150, thresh="gau", preopt=True, sopt_kind="lbfgs",
                            sopt_cfg={"thresh": "gau_tight"}, fix_ends=True, out_dir=str(tmp_path / "out"), log=logs.append, align=False,
                            # (align=False: this toy surface is tied to the lab frame, a rigid fit of P onto R would leave its minimum;
                            #  the alignment step itself is covered by tests/test_prestep.py and, on the engine, tests/test_gpu_calculator.py)
                            gs_kw={"perp_thresh": 2e-2, "climb_rms": 5e-3})
    assert res["fully_grown"] and res["images_ang"].shape == (9, n, 3) and res["device"] == "cpu"
    assert len(res["preopt"]) == 2 and all(p["converged"] for p in res["preopt"])
    assert res["converged"], res["history"][-1]
    e = res["energies"]
    # endpoints sit in the two wells, the highest image on the barrier (h = 0.02 Hartree above the minima)
    assert abs(e[0]) < 1e-6 and abs(e[-1]) < 1e-6 and res["hei_index"] == select_hei_index(e) and 0 < res["hei_index"] < 8
    assert abs(e[res["hei_index"]] - calc.h) < 5e-4
    # the files are the reference's formats and describe the returned path
    syms, xyz, comments = formats.read_trj(res["files"]["final_geometries"])
    assert syms == ["C", "H", "H", "O", "N", "H"] and xyz.shape == (9, n, 3) and np.allclose(xyz, res["images_ang"], atol=1e-14)
    assert np.allclose(formats.read_energies_xyz(res["files"]["final_geometries"]), e, atol=5e-13)
    hs, hx, hc = formats.read_trj(res["files"]["hei"])
    assert np.allclose(hx[0], res["images_ang"][res["hei_index"]], atol=1e-14) and abs(float(hc[0]) - e[res["hei_index"]]) < 5e-13
    # energies belong to the returned geometries
    chk = calc.get_forces_batch(elem, res["images_ang"].reshape(9, -1) * ANG2BOHR)["energy"]
    assert np.allclose(chk, e, atol=1e-12)
    assert any("preopt" in s for s in logs)
    # defaults of the CLI: --fix-ends False moves the endpoints too, no preopt, no files
    res2 = optimize_path_gsm(elem, r_ang, p_ang, calc=DoubleWell(ref_ang * ANG2BOHR), max_nodes=3, max_cycles=6, align=False)
    assert res2["images_ang"].shape == (5, n, 3) and res2["files"] == {} and res2["preopt"] == []
    assert not np.allclose(res2["images_ang"][0], r_ang, atol=1e-6)
