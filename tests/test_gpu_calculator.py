"""GPU tests of the drop-in calculator (reference pdb2reaction/uma_pysis.py API) on the real engine."""
import importlib

import numpy as np
import pytest
import torch

from pdb2reaction_amd import synth

U = importlib.import_module("pdb2reaction_amd.uma_pysis")
pytestmark = pytest.mark.gpu

SYMS = {1: "H", 6: "C", 7: "N", 8: "O", 16: "S"}


@pytest.fixture(scope="module")
def setup(oracle):
    z, imgs, _ = synth.make_images(14, 4, seed=21)
    elem = [SYMS[int(v)].lower() for v in z]           # any case, like the reference (:266)
    return z, elem, imgs


@pytest.mark.parametrize("precision", [None, "split-bf16", "fp32"])
def test_get_forces_matches_oracle_in_atomic_units(oracle, setup, precision):
    z, elem, imgs = setup
    calc = U.uma_pysis(model="synthetic", freeze_atoms=[0, 5], **({"precision": precision} if precision else {}))
    x_bohr = (imgs[0] * U.ANG2BOHR).reshape(-1)
    r = calc.get_forces(elem, x_bohr)
    p32 = (x_bohr.reshape(-1, 3) * U.BOHR2ANG).astype(np.float32)
    e_ref, f_ref = oracle.energy_forces(z, p32.astype(np.float64))
    assert abs(r["energy"] - e_ref * U.EV2AU) <= 1e-4 * U.EV2AU
    f_ref[[0, 5]] = 0.0
    assert r["forces"].dtype == np.float64 and r["forces"].shape == (42,)
    assert np.abs(r["forces"] - (f_ref * U.F_EVAA_2_AU).reshape(-1)).max() <= 1e-3 * U.F_EVAA_2_AU
    assert np.all(r["forces"].reshape(-1, 3)[[0, 5]] == 0.0)
    assert calc.get_energy(elem, x_bohr)["energy"] == pytest.approx(r["energy"], abs=1e-9)
    assert calc._core.engine.precision == precision and not calc._core.engine.widened
    calc.close()


def test_batch_equals_serial_calls(setup):
    z, elem, imgs = setup
    calc = U.uma_pysis(model="synthetic")
    xb = imgs.reshape(4, -1) * U.ANG2BOHR
    rb = calc.get_forces_batch(elem, xb)
    for k in range(4):
        r = calc.get_forces(elem, xb[k])
        assert r["energy"] == rb["energy"][k] and np.array_equal(r["forces"], rb["forces"][k])


def test_fd_hessian_against_oracle_fd(oracle, setup):
    z, elem, imgs = setup
    calc = U.uma_pysis(model="synthetic", freeze_atoms=[3], out_hess_torch=True)
    x_bohr = (imgs[1] * U.ANG2BOHR).reshape(-1)
    r = calc.get_hessian(elem, x_bohr)
    h = r["hessian"]
    assert isinstance(h, torch.Tensor) and h.dtype == torch.float64 and h.shape == (42, 42) and h.is_cuda
    h = h.cpu().numpy()
    assert np.allclose(h, h.T)
    # oracle central differences with the same step on a few active columns
    p = (x_bohr.reshape(-1, 3) * U.BOHR2ANG).astype(np.float32).astype(np.float64)
    full = np.zeros((42, 42))
    cols = [0, 7, 20, 41]
    for k in cols:
        a, c = divmod(k, 3)
        pp, pm = p.copy(), p.copy()
        pp[a, c] += 1e-3
        pm[a, c] -= 1e-3
        fp = oracle.energy_forces(z, pp)[1].reshape(-1)
        fm = oracle.energy_forces(z, pm)[1].reshape(-1)
        full[:, k] = -(fp - fm) / 2e-3
    # compare un-symmetrised information: H_sym[i,k] = (H[i,k]+H[k,i])/2, check the diagonal block of those columns
    sub = np.ix_(cols, cols)
    ref = 0.5 * (full[sub] + full[sub].T) * U.H_EVAA_2_AU
    assert np.abs(h[sub] - ref).max() <= 5e-3 * U.H_EVAA_2_AU          # fp32 forces / 2e-3 A ~ 1e-3 eV/A^2 noise floor
    calc2 = U.uma_pysis(model="synthetic", freeze_atoms=[3], return_partial_hessian=True, out_hess_torch=False, hessian_double=False)
    h2 = calc2.get_hessian(elem, x_bohr)["hessian"]
    assert h2.shape == (39, 39) and h2.dtype == np.float32


def test_device_resident_fd_hessian_equals_the_host_entry(setup):
    """Round 6: ``get_hessian`` builds the displaced geometries on the GPU and keeps the forces there
    (``UMAcore.compute_batch_dev``, the engine's device-pointer entry) -- bit for bit the columns of the host entry, which copies 64 x N x 3 floats
    over PCIe each way per call."""
    from pdb2reaction_amd import hessian as H

    z, elem, imgs = setup
    calc = U.uma_pysis(model="synthetic", freeze_atoms=[3, 9])
    core = calc._ensure_core(elem)
    x_ang = imgs[2].astype(np.float64)
    seen = {"host": 0, "dev": 0}

    def host(c):
        seen["host"] += len(c)
        return core.compute_batch(c, forces=True)["forces"]

    def devf(p32):
        assert p32.is_cuda and p32.dtype == torch.float32
        seen["dev"] += len(p32)
        return core.compute_batch_dev(p32)

    kw = dict(device=core.device, double=True, partial=False, batch=8)
    h_host = H.fd_hessian(host, x_ang, calc.freeze_atoms, **kw)
    h_dev = H.fd_hessian(host, x_ang, calc.freeze_atoms, batch_forces_dev=devf, **kw)
    assert seen["dev"] == 2 * 3 * 12 and seen["host"] == seen["dev"]                   # the device run never touched the host entry
    assert torch.equal(h_host, h_dev)
    r = calc.get_hessian(elem, (x_ang * U.ANG2BOHR).reshape(-1))                       # the calculator takes the device entry by itself
    assert torch.equal(r["hessian"], H.hessian_to_au(h_dev, double=True, as_torch=True))
    calc.close()


def test_device_batch_entry_widens_like_the_host_entry(weights, monkeypatch):
    """``UMAcore.compute_batch_dev`` (the device-resident FD Hessian's force entry) goes through the asynchronous device-pointer entry, which cannot
    refuse its own result: an activation beyond the fast mode's fp16 operand range must be noticed there, the engine widened to bf16 forward planes
    and the batch repeated -- bitwise what an engine created in split-bf16 returns, as the host entry does (test_gpu_parity)."""
    from pdb2reaction_amd.engine import Engine

    big = dict(weights)
    key = "blocks.0.edge_wise.so2_conv_1.rad_func.fc3"
    big[key + ".weight"] = (np.asarray(weights[key + ".weight"]) * 3e4).astype(np.float32)
    z, imgs, _ = synth.make_images(40, 3, seed=2)
    p32 = np.asarray(imgs, dtype=np.float32)
    ref = Engine(0, precision="split-bf16")
    monkeypatch.setenv("UMX_PRECISION", "split")
    eng = Engine(0)
    try:
        ref.load_weights(big); ref.set_system(z)
        _, f0 = ref.energy_forces(p32)
        eng.load_weights(big); eng.set_system(z)
        core = U.UMAcore.__new__(U.UMAcore)
        core.engine = eng
        with pytest.warns(RuntimeWarning, match="split-bf16"):
            f = core.compute_batch_dev(torch.as_tensor(p32, device="cuda"))
        assert eng.widened and eng.precision_mode() == "split-bf16"
        assert f.is_cuda and np.array_equal(f.cpu().numpy(), f0)
        assert np.array_equal(core.compute_batch_dev(torch.as_tensor(p32, device="cuda")).cpu().numpy(), f0)      # stays widened
    finally:
        eng.close()
        ref.close()


def test_error_behaviour(setup):
    z, elem, imgs = setup
    with pytest.raises(RuntimeError, match="no CPU path"):
        U.uma_pysis(model="synthetic", device="cpu").get_energy(elem, imgs[0].reshape(-1))
    with pytest.raises(ValueError):
        U.uma_pysis(model="synthetic").get_energy(["Xx"] * 14, imgs[0].reshape(-1))
    c = U.uma_pysis(model="synthetic")
    c.get_energy(elem, imgs[0].reshape(-1))
    with pytest.raises(RuntimeError, match="Analytical Hessian is not available"):
        c._core.compute(imgs[0], forces=True, hessian=True)
    with pytest.raises(ValueError):
        c.get_energy(elem, np.zeros(9))                 # wrong atom count for the bound system


class FakeAtoms:
    """Minimal ASE-Atoms-like object (ASE is not installed here)."""

    def __init__(self, z, pos, charge=0, spin=1):
        self.numbers = np.asarray(z)
        self._p = np.asarray(pos, dtype=float)
        self.info = {"charge": charge, "spin": spin}
        self.calc = None

    def get_positions(self):
        return self._p

    def get_atomic_numbers(self):
        return self.numbers


def test_ase_style_calculator(oracle, setup):
    """Secondary boundary (reference path_opt.py:351-363,418-423): eV / eV/A, atoms.info charge & spin honoured."""
    from pdb2reaction_amd.ase_calculator import UMXCalculator

    z, elem, imgs = setup
    calc = UMXCalculator(model="synthetic", task_name="omol")
    images = [FakeAtoms(z, imgs[k].astype(np.float32)) for k in range(3)]
    for im in images:
        im.calc = calc
    e0 = calc.get_potential_energy(images[0])
    f0 = calc.get_forces(images[0])
    e_ref, f_ref = oracle.energy_forces(z, imgs[0].astype(np.float32).astype(np.float64))
    assert abs(e0 - e_ref) <= 1e-4 and np.abs(f0 - f_ref).max() <= 1e-3 and f0.dtype == np.float64
    eb, fb = calc.calculate_images(images)
    assert eb[0] == e0 and np.array_equal(fb[0], f0) and fb.shape == (3, 14, 3)
    charged = FakeAtoms(z, imgs[0].astype(np.float32), charge=-1, spin=2)
    e_c = calc.get_potential_energy(charged)
    e_cref, _ = oracle.energy_forces(z, imgs[0].astype(np.float32).astype(np.float64), charge=-1, spin=2, forces=False)
    assert abs(e_c - e_cref) <= 1e-4 and abs(e_c - e0) > 1e-3
    # energy then forces of an UNCHANGED image is one evaluation (torch_dmf asks for both, image by image); any change evaluates again
    calls = []
    orig = calc._engine.energy_forces
    calc._engine.energy_forces = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    im = FakeAtoms(z, imgs[1].astype(np.float32))
    e1 = calc.get_potential_energy(im)
    f1 = calc.get_forces(im)
    assert len(calls) == 1 and calc.get_potential_energy(im) == e1 and len(calls) == 1
    im._p = im._p + np.float32(0.01)
    f2 = calc.get_forces(im)
    assert len(calls) == 2 and not np.array_equal(f1, f2)
    calc.get_forces(charged)                                   # other charge / spin: re-bound, evaluated
    assert len(calls) == 3


def test_gsm_driver_on_the_engine(tmp_path, setup):
    """Row f1 end to end: growing string between two geometries, every cycle ONE batched engine call; outputs written
    in the reference's .trj format (row f2)."""
    from pdb2reaction_amd import formats
    from pdb2reaction_amd.gsm import GrowingStringDriver

    z, elem, imgs = setup
    calc = U.uma_pysis(model="synthetic", freeze_atoms=[0])
    r, p = (imgs[0] * U.ANG2BOHR).reshape(-1), (imgs[3] * U.ANG2BOHR).reshape(-1)
    drv = GrowingStringDriver(elem, r, p, calc, gs_kw={"max_nodes": 4, "perp_thresh": 1e3, "climb": False}, stopt_kw={"max_cycles": 12, "max_step": 0.05})
    res = drv.run()
    assert res.fully_grown and res.coords.shape == (6, 42) and np.isfinite(res.energies).all()
    assert res.force_evaluations >= 4 + 6 + 4 + 4
    # behaviour, not just finiteness:
    #  - the returned energies ARE the energies of the returned geometries (bitwise: one batched call either way)
    chk = calc.get_forces_batch(elem, res.coords)
    assert np.array_equal(chk["energy"][1:-1], res.energies[1:-1])
    assert calc.get_energy(elem, res.coords[0])["energy"] == pytest.approx(res.energies[0], abs=1e-9)
    #  - fixed endpoints are untouched
    assert np.array_equal(res.coords[0], r) and np.array_equal(res.coords[-1], p)
    #  - the optimiser does its job: the perpendicular force of the fully grown string falls from its first to its last cycle
    full = [h for h in res.history if h["images"] == 6]
    assert len(full) >= 6 and full[-1]["rms_fperp"] < 0.95 * full[0]["rms_fperp"]
    #  - "equi" reparametrisation keeps the nodes evenly spread along the path
    seg = np.linalg.norm(np.diff(res.coords, axis=0), axis=1)
    assert seg.max() / seg.min() < 1.5
    #  - forces of interior images are perpendicular-dominated no more than at the start: F_perp of the final string from a
    #    fresh evaluation agrees with the driver's last bookkeeping (same tangents, same forces)
    from pdb2reaction_amd.gsm import _tangents
    t = _tangents(res.coords)
    f = chk["forces"]
    fperp = f - (f * t).sum(1, keepdims=True) * t
    assert np.sqrt((fperp[1:-1] ** 2).mean()) < 1.05 * full[0]["rms_fperp"]
    path = tmp_path / "final_geometries.trj"
    formats.write_trj_with_energy([e.capitalize() for e in elem], res.coords.reshape(6, -1, 3) * U.BOHR2ANG, res.energies, path)
    assert np.allclose(formats.read_energies_xyz(path), res.energies, atol=1e-12)
    syms, xyz, _ = formats.read_trj(path)
    assert xyz.shape == (6, 14, 3)


def test_device_resident_gsm_driver_equals_the_host_path(setup):
    """Round 4: ``GrowingStringDriver.from_calculator`` keeps the string on the engine's GPU and evaluates it through
    the device-pointer entry (``parallel.EngineStringEvaluator``) -- per image the same numbers as ``get_forces_batch`` (same float32
    positions into the engine, same frozen rows, same unit factors), so the run follows the host-path run: same growth, same cycle
    count, coordinates to round-off (the float64 reductions of the string update run in another order on the GPU)."""
    import torch
    from pdb2reaction_amd.gsm import GrowingStringDriver
    from pdb2reaction_amd.parallel import EngineStringEvaluator

    z, elem, imgs = setup
    r, p = (imgs[0] * U.ANG2BOHR).reshape(-1), (imgs[3] * U.ANG2BOHR).reshape(-1)
    kw = dict(gs_kw={"max_nodes": 4, "perp_thresh": 1e3, "climb": False}, stopt_kw={"max_cycles": 12, "max_step": 0.05})
    calc = U.uma_pysis(model="synthetic", freeze_atoms=[0, 7])
    host = GrowingStringDriver(elem, r, p, calc, **kw).run()
    drv = GrowingStringDriver.from_calculator(elem, r, p, calc, **kw)
    assert drv.device.type == "cuda" and isinstance(drv._evaluate_device, EngineStringEvaluator)
    dev = drv.run()
    assert dev.cycles == host.cycles and dev.fully_grown and [h["images"] for h in dev.history] == [h["images"] for h in host.history]
    assert np.abs(dev.coords - host.coords).max() <= 1e-9 and np.abs(dev.energies - host.energies).max() <= 1e-9
    # the evaluator IS get_forces_batch on the device: bitwise per image
    x = torch.as_tensor(host.coords, dtype=torch.float64, device=drv.device)
    e_d, f_d = drv._evaluate_device(x)
    chk = calc.get_forces_batch(elem, host.coords)
    assert np.array_equal(e_d.cpu().numpy(), chk["energy"]) and np.array_equal(f_d.cpu().numpy(), chk["forces"])
    assert np.all(f_d.cpu().numpy().reshape(6, -1, 3)[:, [0, 7]] == 0.0)
    assert dev.timing["total_s"] > 0 and dev.timing["redo_steps"] >= 0
    # a stand-in calculator without an engine keeps the numpy path
    class Plain:
        def get_forces_batch(self, atoms, c):
            return calc.get_forces_batch(atoms, c)
    assert GrowingStringDriver.from_calculator(elem, r, p, Plain(), **kw).device.type == "cpu"


def test_rfo_single_structure_optimisation_on_the_engine(setup):
    """The RFO branch of ``_optimize_single`` (path_opt.py:464-518) on the engine: the initial Hessian is ``get_hessian`` (finite
    differences of BATCHED engine forces), every cycle one E+F; frozen atoms stay put and the energy falls cycle by cycle (the synthetic
    random-weight surface is rugged: twelve trust-radius-limited cycles lower the energy but do not yet shrink the largest force)."""
    from pdb2reaction_amd.rfo import optimize_single

    z, elem, imgs = setup
    calc = U.uma_pysis(model="synthetic", freeze_atoms=[0, 3], out_hess_torch=True)
    x0 = imgs[0] * U.ANG2BOHR
    f0 = calc.get_forces(elem, x0.reshape(-1))
    res = optimize_single(calc, elem, x0, "rfo", {"thresh": "gau_loose", "max_cycles": 12, "gdiis": False, "line_search": False, "adapt_step_func": False},
                          freeze=[0, 3])
    assert res["n_hessian_calls"] == 1 and res["n_force_calls"] == res["cycles"] + 1
    assert np.array_equal(res["coords"][[0, 3]], x0[[0, 3]])
    assert res["energy"] < f0["energy"] - 1e-4
    act = np.ones(len(z), bool); act[[0, 3]] = False
    assert np.isfinite(res["forces"]).all() and np.all(res["forces"].reshape(-1, 3)[~act] == 0.0)     # (frozen rows zeroed by the calculator)
    es = [h["energy"] for h in res["history"]]
    assert all(b <= a + 5e-6 for a, b in zip(es, es[1:]))           # restricted steps on a model Hessian: no energy rise beyond float32 noise
    lb = optimize_single(calc, elem, x0, "lbfgs", {"thresh": "gau_loose", "max_cycles": 12}, freeze=[0, 3])
    assert lb["energy"] < f0["energy"] - 1e-4 and np.array_equal(lb["coords"][[0, 3]], x0[[0, 3]])


def test_path_opt_gsm_flow_on_the_engine(tmp_path, setup):
    """The GSM branch of the reference's path-opt (path_opt.py:815-1048) as one call on the engine: shared calculator, pre-alignment
    (three frozen anchors -> rigid fit + staged anchor scan with batched L-BFGS), device-resident growing string, the reference's output
    files."""
    from pdb2reaction_amd import formats
    from pdb2reaction_amd.path_opt import optimize_path_gsm

    z, elem, imgs = setup
    rot = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    p_moved = imgs[3] @ rot.T + np.array([1.0, 2.0, -0.5])            # the product arrives in another frame: the alignment step has to undo it
    res = optimize_path_gsm(elem, imgs[0], p_moved, calc_kw={"model": "synthetic"}, freeze_atoms=[0, 5, 9], max_nodes=4, max_cycles=10,
                            gs_kw={"perp_thresh": 1e3, "climb": False}, stopt_kw={"max_step": 0.05}, out_dir=str(tmp_path / "po"),
                            align_kw={"per_step_cycles": 2, "final_cycles": 3})
    assert res["device"].startswith("cuda") and res["fully_grown"] and res["images_ang"].shape == (6, len(z), 3)
    assert np.isfinite(res["energies"]).all() and len(res["align"]) == 1 and "error" not in res["align"][0]
    # after the alignment the frozen anchors of the product coincide with the reactant's (and stay there: frozen atoms never move)
    assert np.abs(res["images_ang"][-1][[0, 5, 9]] - imgs[0][[0, 5, 9]]).max() < 1e-6
    e_file = formats.read_energies_xyz(res["files"]["final_geometries"])
    assert np.allclose(e_file, res["energies"], atol=5e-13)
    calc = U.uma_pysis(model="synthetic", freeze_atoms=[0, 5, 9])
    chk = calc.get_forces_batch(elem, res["images_ang"].reshape(6, -1) * U.ANG2BOHR)["energy"]
    assert np.abs(chk - res["energies"]).max() < 2e-9             # the written energies are the energies of the written geometries
    syms, xyz, _ = formats.read_trj(res["files"]["hei"])
    assert np.allclose(xyz[0], res["images_ang"][res["hei_index"]], atol=1e-14)


def test_staged_scan_with_batched_lbfgs_on_the_engine(setup):
    """Row f3 end to end: two mobile images are rigidly fitted onto a reference and their anchors dragged onto it while
    the rest relaxes -- every L-BFGS cycle is ONE batched engine call for both images."""
    from pdb2reaction_amd import prestep as PS

    z, elem, imgs = setup
    calc = U.uma_pysis(model="synthetic")
    ref = imgs[0] * U.ANG2BOHR
    anchors = [0, 5, 9]
    rot = PS._rodrigues(np.array([0.3, -1.0, 0.5]), 0.7)
    mobs = []
    for k in (1, 2):
        m = imgs[k] * U.ANG2BOHR
        m[anchors] += 0.25 * (k + 1) * np.array([0.0, 1.0, 0.0])       # anchors displaced by 0.26 / 0.40 A before the rigid move
        mobs.append(m @ rot.T + np.array([3.0, -1.0, 2.0]))
    out, res = PS.align_and_refine_sequence(calc, elem, [ref] + mobs, [anchors] * 3, step_A=0.1, per_step_cycles=3, final_cycles=6,
                                            thresh="gau_loose")
    assert [r["align"]["mode"] for r in res] == ["kabsch", "kabsch"] and all(r["scan"]["converged"] for r in res)
    assert all(2 <= r["scan"]["n_steps"] <= 8 for r in res)
    for o in out[1:]:
        np.testing.assert_array_equal(o[anchors], ref[anchors])
        assert np.isfinite(o).all() and PS.rmsd_ang(o, ref) < 3.0          # synthetic weights: no physical minimum nearby
    # relaxation lowers the energy relative to the same anchors-on-target geometry without relaxation
    start = PS.align_second_to_first(ref, mobs[0], anchors)[0]
    start[anchors] = ref[anchors]
    e = calc.get_energy_batch(elem, np.stack([start, out[1]]).reshape(2, -1))["energy"]
    assert e[1] < e[0]


def test_engine_first_then_torch_cuda_in_one_process():
    """Regression: the engine brought the GPU up through /opt/rocm's HIP runtime and a LATER torch.cuda use in the same
    process failed with "No HIP GPUs are available" (torch bundles its own runtime).  Fresh interpreter, engine first."""
    import os
    import subprocess
    import sys

    code = ("import numpy as np\n"
            "from pdb2reaction_amd.engine import Engine\n"
            "e = Engine(0)\n"
            "d1, d2, c = e.bond_changes(np.zeros((2, 3)), np.ones((2, 3)), np.ones(2))\n"
            "import torch\n"
            "print('OK', float(torch.ones(3, device='cuda').sum().item()))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK 3.0" in out.stdout, out.stderr[-2000:]


def test_real_checkpoint_path_end_to_end(tmp_path, monkeypatch, oracle, setup):
    """Rows a5 / f4: fairchem-style state dict (MoLE experts + routing net) -> checkpoint.convert_for_system -> a .umxw blob under
    $UMX_WEIGHTS_DIR -> ``uma_pysis(model="uma-s-1p1")`` loads THAT file (reference uma_pysis.py:246-250 loads its checkpoint by
    name), energies match the oracle run on the merged weights, and binding the blob to another system is refused."""
    import torch as T
    from test_checkpoint import _fake_state
    from pdb2reaction_amd import checkpoint as CK, weights as W

    z, elem, imgs = setup
    w = W.make_synthetic_weights(0)
    state, _, extra = _fake_state(w)
    rng = np.random.default_rng(5)
    n_exp = next(v.shape[0] for k, v in state.items() if k.endswith(".weights"))
    c = W.SPHERE_CHANNELS
    state["backbone.composition_embedding.weight"] = T.tensor(rng.standard_normal((W.MAX_NUM_ELEMENTS, c)))
    state["backbone.routing_mlp.0.weight"] = T.tensor(rng.standard_normal((32, 2 * c)) / np.sqrt(2 * c))
    state["backbone.routing_mlp.0.bias"] = T.tensor(0.1 * rng.standard_normal(32))
    state["backbone.routing_mlp.2.weight"] = T.tensor(rng.standard_normal((n_exp, 32)) / 6.0)
    state["backbone.routing_mlp.2.bias"] = T.tensor(0.1 * rng.standard_normal(n_exp))
    from test_checkpoint import UMA_S_CONFIG
    # the checkpoint says cutoff 5.0 A / 40 neighbours: the calculator must evaluate with THOSE (reference: backbone.cutoff /
    # backbone.max_neighbors, uma_pysis.py:301-309), and the oracle below is run with the same graph
    ckpt = {"config": {"model": {"backbone": {**UMA_S_CONFIG, "cutoff": 5.0, "max_neighbors": 40}}}, "state_dict": state}
    blob = CK.convert_checkpoint(ckpt, z, 0, 1, "omol", extra=extra)
    (tmp_path / "uma-s-1p1.umxw").write_bytes(blob)
    monkeypatch.setenv("UMX_WEIGHTS_DIR", str(tmp_path))
    monkeypatch.delenv("UMX_ALLOW_SYNTHETIC", raising=False)
    calc = U.uma_pysis()                                       # reference defaults: model="uma-s-1p1"
    x_bohr = (imgs[0] * U.ANG2BOHR).reshape(-1)
    r = calc.get_forces(elem, x_bohr)
    from oracle.escn_md_oracle import Oracle
    merged = W.unpack_blob(blob)
    assert merged.meta["model"]["cutoff"] == 5.0 and calc._core.model_record["max_neighbors"] == 40
    e_ref, f_ref = Oracle(merged, cutoff=5.0, max_neigh=40).energy_forces(z, imgs[0].astype(np.float32).astype(np.float64))
    e_6, _ = Oracle(merged).energy_forces(z, imgs[0].astype(np.float32).astype(np.float64))
    assert abs(e_6 - e_ref) > 1e-3                              # the default 6.0 A / 300 graph gives another energy: the blob's values matter
    assert abs(r["energy"] / U.EV2AU - e_ref) <= 1e-4
    assert np.abs(r["forces"].reshape(-1, 3) / U.F_EVAA_2_AU - f_ref).max() <= 1e-3
    # the merged experts differ from the un-routed synthetic set, so this really is the file's physics
    e_syn = U.uma_pysis(model="synthetic").get_energy(elem, x_bohr)["energy"]
    assert abs(e_syn - r["energy"]) > 1e-6
    with pytest.raises(ValueError, match="MoLE-merged for another system"):
        U.uma_pysis(charge=1).get_energy(elem, x_bohr)
    with pytest.raises(ValueError, match="composition"):
        U.uma_pysis().get_energy(elem[:-1], x_bohr[:-3])
