"""GPU parity at the BASELINE sizes the headline is quoted on.

tests/golden/c3c4_n2000.npz (tools/make_golden_c3.py) holds float64 oracle ENERGIES AND FORCES of 2000-atom images:
two images of the c3 string (16 images) and two of the c4 string (24 images), plus two oracle finite-difference Hessian
columns at c4 image 0.  The large-M code paths only exist at this size (2.3 M edges per iteration, 256 x 256 tiles, XCD
mapping, index products above 2^31), and in split mode the reverse pass uses 16-bit-significand products -- so the
forces are held to the north-star tolerance here, in BOTH precision modes, through the C ABI.

  c3: ~2000-atom cluster, GSM 16 images                      (BASELINE.json configs[2])
  c4: ~2000-atom cluster, DMF path_opt 24 images + freq Hessian (configs[3]; reference path_opt.py:399-426,
      uma_pysis.py:595-686)
"""
import importlib

import numpy as np
import pytest
import torch

from conftest import load_golden
from pdb2reaction_amd import synth, weights as W

U = importlib.import_module("pdb2reaction_amd.uma_pysis")

pytestmark = pytest.mark.gpu

TOL_E = 1e-4   # eV      (BASELINE.json north_star)


def energy_tol(n_atoms, mode="auto"):
    """include/umx.h: UMX_ENERGY_TOL_EV_N(n) = max(1e-4, 5e-9 n) eV in the default (bf16x3) and split-bf16 modes (the energy is the forward
    pass: split-bf16 == bf16x3 there) -- the north-star's FLAT 1e-4 eV at every BASELINE size, 20 000 atoms included (round 6: the last
    coherent term of the matrix cores, stage 1 of their adder, is gone); UMX_ENERGY_TOL_EV_FP32_N(n) = max(1e-4, 1e-8 n) in
    the fp32 mode (chains of IEEE FMAs: a wider scatter than the default mode's, worst -1.61e-4 eV of nine 20 000-atom cases);
    UMX_ENERGY_TOL_EV_FAST_N(n) = max(1e-4, 1.5e-7 n) in the fast mode -- 3e-4 eV at the headline size: the fast mode's error is a fixed
    -5.4e-8 ... +8.9e-8 eV per atom that depends on the weight set (two of eight sets beyond 1e-4 eV at 2000 atoms, profiles/r06_c3_weight_sets.txt;
    found at the end of round 6 -- until then the rule here was 1e-4 eV up to 2000 atoms, 6e-8 n beyond).  Ten 20 000-atom cases:
    bf16x3 -5.3e-5 ... +2.8e-5 eV, fp32 -1.61e-4 ... +5.3e-5, split -1.04e-3 ... +1.78e-3."""
    if mode in ("split", "split-f16"):
        return max(TOL_E, 1.5e-7 * n_atoms)
    return max(TOL_E, (1e-8 if mode == "fp32" else 5e-9) * n_atoms)
TOL_F = 1e-3   # eV/A


@pytest.fixture(scope="module")
def gold():
    return load_golden("c3c4_n2000")


@pytest.mark.parametrize("mode", ["split", "split-bf16", "bf16x3", "fp32"])
def test_c3_energy_and_forces_against_f64_oracle(weights, gold, mode, monkeypatch):
    """The headline configuration: E and F of 2000-atom images vs the float64 oracle, every precision mode (split = fp16 forward
    planes, the default; split-bf16 = three bf16 forward planes; fp32 = fp32 MFMA everywhere)."""
    from pdb2reaction_amd.engine import Engine

    monkeypatch.setenv("UMX_PRECISION", mode)
    eng = Engine(0)
    try:
        eng.load_weights(weights)
        eng.set_system(gold["z"])
        e, f = eng.energy_forces(gold["c3_pos"])
        ne, maxdeg = eng.graph_stats()
        assert ne > 2 * 140_000 and maxdeg <= 300
        de = np.abs(e - gold["c3_energy"])
        df = np.abs(f.astype(np.float64) - gold["c3_forces"])
        print(f"[c3 {mode}] |dE| = {de.max():.2e} eV, max|dF| = {df.max():.2e} eV/A, rms dF = {np.sqrt((df ** 2).mean()):.2e}")
        assert de.max() <= TOL_E, (mode, de)                            # the north-star's 1e-4 eV in EVERY mode at the headline size ON THIS WEIGHT SET
        assert energy_tol(2000, mode) == (TOL_E if mode != "split" else 3e-4)      # (the fast mode's own bound is wider: test_c3_fast_mode_... below)
        assert df.max() <= TOL_F, (mode, df.max())
        # the reverse pass is systematic-error free too: the net force error over 2000 atoms stays at round-off level
        assert np.abs((f.astype(np.float64) - gold["c3_forces"]).sum(axis=1)).max() <= 5e-4
    finally:
        eng.close()


class FakeAtoms:
    """The slice of ase.Atoms that torch_dmf / the reference's DMF path touches (path_opt.py:351-363,418-423)."""

    def __init__(self, z, pos, charge=0, spin=1):
        self.numbers = np.asarray(z)
        self._p = np.asarray(pos, dtype=np.float64)
        self.info = {"charge": charge, "spin": spin}
        self.calc = None

    def get_positions(self):
        return self._p

    def get_atomic_numbers(self):
        return self.numbers


def test_c4_dmf_24_images_one_batched_call(gold):
    """c4's string: ONE calculate_images call for the 24 DMF images of a 2000-atom cluster (ASE protocol, eV and eV/A)."""
    from pdb2reaction_amd.ase_calculator import UMXCalculator

    z, imgs, _ = synth.make_images(2000, 24)
    p32 = imgs.astype(np.float32)
    i0, i1 = (int(i) for i in gold["c4_index"])
    assert np.array_equal(p32[[i0, i1]], gold["c4_pos"]) and np.array_equal(z, gold["z"])
    calc = UMXCalculator(model="synthetic", task_name="omol")
    images = [FakeAtoms(z, p32[k]) for k in range(24)]
    for im in images:
        im.calc = calc
    e, f = calc.calculate_images(images)
    assert e.shape == (24,) and f.shape == (24, 2000, 3) and f.dtype == np.float64
    assert np.isfinite(e).all() and np.isfinite(f).all()
    for j, k in enumerate((i0, i1)):
        assert abs(e[k] - gold["c4_energy"][j]) <= TOL_E
        assert np.abs(f[k] - gold["c4_forces"][j]).max() <= TOL_F
    # the image-by-image protocol torch_dmf uses gives bit-identical numbers (batch size / chunking never changes a result)
    for k in (i1, 23):
        assert calc.get_potential_energy(images[k]) == e[k]
        assert np.array_equal(calc.get_forces(images[k]), f[k])
    assert np.abs(f.sum(axis=1)).max() <= 5e-4                                   # Newton's third law, every image
    calc.close()                                                                 # hand the HBM workspace back (24 images of 2000 atoms)


def test_c4_fd_hessian_2000_atoms(gold):
    """c4's 'freq Hessian (3N force batches)': get_hessian on a 2000-atom image with 66 active DOF (the rest frozen),
    compared with central differences of the float64 oracle forces on the two golden columns -- full 6000-long columns,
    frozen rows included -- plus the return_partial_hessian / dtype / container variants (uma_pysis.py:515-551,595-686)."""
    z = gold["z"]
    i0 = int(gold["c4_index"][0])
    assert i0 == 0
    p64 = gold["c4_pos"][0].astype(np.float64)
    order = np.argsort(np.einsum("ij,ij->i", p64, p64))
    k_gold = [int(k) for k in gold["hess_dof"]]
    active_atoms = sorted(set(int(a) for a in order[0:41:2]) | {int(order[1])} | {k // 3 for k in k_gold})
    assert len(active_atoms) == 22
    frozen = [a for a in range(2000) if a not in set(active_atoms)]
    elem = [synth.SYMBOLS[int(q)] for q in z]
    x_bohr = (p64 * U.ANG2BOHR).reshape(-1)

    calc = U.uma_pysis(model="synthetic", freeze_atoms=frozen, out_hess_torch=False)
    r = calc.get_hessian(elem, x_bohr)
    h = r["hessian"]
    assert isinstance(h, np.ndarray) and h.dtype == np.float64 and h.shape == (6000, 6000)
    assert np.array_equal(h, h.T)
    assert abs(r["energy"] / U.EV2AU - gold["c4_energy"][0]) <= TOL_E
    fz = r["forces"].reshape(2000, 3)
    assert np.all(fz[frozen] == 0.0)                                              # uma_pysis.py:561-567
    assert np.abs(fz[active_atoms] / U.F_EVAA_2_AU - gold["c4_forces"][0][active_atoms]).max() <= TOL_F
    act_dof = np.array([3 * a + c for a in active_atoms for c in range(3)])
    frz_dof = np.array([3 * a + c for a in frozen for c in range(3)])
    assert np.all(h[np.ix_(frz_dof, frz_dof)] == 0.0)                             # columns of frozen DOF are never built
    tol_h = 1.5e-2          # eV/A^2: float32 forces (round-off ~1e-5 eV/A) / (2 * 1e-3 A); entries reach 0.68 eV/A^2
    cols = gold["hess_cols"]                                                      # (2, 6000) eV/A^2, oracle FD
    for j, k in enumerate(k_gold):
        # frozen rows: H_sym[i,k] = (H[i,k] + 0) / 2 -- the whole column against the oracle
        got = h[frz_dof, k] / U.H_EVAA_2_AU
        assert np.abs(got - 0.5 * cols[j][frz_dof]).max() <= tol_h
        assert np.abs(cols[j][frz_dof]).max() > 0.05                              # ... and it is not a comparison of zeros
    sub = np.ix_(k_gold, k_gold)
    ref = 0.5 * (cols[:, k_gold] + cols[:, k_gold].T)                             # symmetrised 2x2 block of the golden columns
    assert np.abs(h[sub] / U.H_EVAA_2_AU - ref.T).max() <= tol_h
    assert np.abs(np.diag(ref)).min() > 0.03                                      # (synthetic weights: curvatures are O(0.1) eV/A^2)

    # partial Hessian: active block only, float32, torch on the device
    calc2 = U.uma_pysis(model="synthetic", freeze_atoms=frozen, return_partial_hessian=True, hessian_double=False, out_hess_torch=True)
    h2 = calc2.get_hessian(elem, x_bohr)["hessian"]
    assert isinstance(h2, torch.Tensor) and h2.is_cuda and h2.dtype == torch.float32 and tuple(h2.shape) == (66, 66)
    h2 = h2.cpu().numpy().astype(np.float64)
    assert np.abs(h2 - h[np.ix_(act_dof, act_dof)]).max() <= 1e-6 * np.abs(h2).max() + 1e-7
    calc.close(); calc2.close()


@pytest.mark.parametrize("n_atoms,n_img", [(500, 12), (2000, 16)])
def test_c2_c3_gsm_driver_at_baseline_sizes(n_atoms, n_img, gold):
    """BASELINE configs[1] and [2]: ~500-atom cluster with 12 images / ~2000-atom cluster with 16 images (max_nodes = images - 2,
    path_opt.py:58,171) on one GPU: the batched driver grows the string to its full size with ONE engine call per cycle, and the
    endpoint it started from matches the golden energy / forces of that configuration."""
    from pdb2reaction_amd.gsm import GrowingStringDriver

    z, imgs, frozen = synth.make_images(n_atoms, n_img)
    if n_atoms == 500:
        g = load_golden("c2_n500_k2")
        assert np.array_equal(imgs[[0, 6]].astype(np.float32), g["pos"])
    else:
        g = {"energy": gold["c3_energy"], "forces": gold["c3_forces"]}
        assert np.array_equal(imgs[0].astype(np.float32), gold["c3_pos"][0])
    elem = [synth.SYMBOLS[int(q)] for q in z]
    calc = U.uma_pysis(model="synthetic", freeze_atoms=frozen)
    calls = []
    inner = calc.get_forces_batch

    def counted(el, c):
        calls.append(len(c))
        return inner(el, c)

    calc.get_forces_batch = counted
    r, p = (imgs[0] * U.ANG2BOHR).reshape(-1), (imgs[n_img - 1] * U.ANG2BOHR).reshape(-1)
    drv = GrowingStringDriver(elem, r, p, calc, gs_kw={"max_nodes": n_img - 2, "perp_thresh": 1e3, "climb": False},
                              stopt_kw={"max_cycles": n_img // 2 + 2, "max_step": 0.05})
    res = drv.run()
    assert res.fully_grown and res.coords.shape == (n_img, 3 * n_atoms) and np.isfinite(res.energies).all()
    assert len(calls) <= res.cycles + 1 and max(calls) == n_img and res.force_evaluations == sum(calls)
    # the DEVICE-RESIDENT form of the same run (round 4: the string never leaves the GPU, one read of 2K+8 doubles per cycle) follows it:
    # same growth sequence, same number of evaluations, coordinates and energies to round-off
    n_host = len(calls)
    dev = GrowingStringDriver.from_calculator(elem, r, p, calc, gs_kw={"max_nodes": n_img - 2, "perp_thresh": 1e3, "climb": False},
                                              stopt_kw={"max_cycles": n_img // 2 + 2, "max_step": 0.05})
    assert dev.device.type == "cuda"
    rd = dev.run()
    assert len(calls) == n_host                                  # (the device path does not go through get_forces_batch)
    assert rd.cycles == res.cycles and rd.force_evaluations == res.force_evaluations and [h["images"] for h in rd.history] == [h["images"] for h in res.history]
    assert np.abs(rd.coords - res.coords).max() <= 1e-8 and np.abs(rd.energies - res.energies).max() <= 1e-8
    # endpoint energy of the driver == golden energy of c2 image 0 (Hartree vs eV), forces of the frozen atoms are zero
    assert abs(res.energies[0] / U.EV2AU - g["energy"][0]) <= TOL_E
    f0 = inner(elem, res.coords[:1])["forces"].reshape(n_atoms, 3)
    assert np.all(f0[frozen] == 0.0)
    act = np.setdiff1d(np.arange(n_atoms), frozen)
    assert np.abs(f0[act] / U.F_EVAA_2_AU - g["forces"][0][act]).max() <= TOL_F
    full = [h for h in res.history if h["images"] == n_img]
    assert full and full[-1]["rms_fperp"] <= full[0]["rms_fperp"]
    calc.close()


@pytest.mark.parametrize("mode", ["auto", "split", "fp32"])
@pytest.mark.parametrize("name", ["c5_n20000", "c5_n20000_g1", "c5_n20000_w1"])
def test_c5_energy_and_forces_against_f64_oracle(name, mode, monkeypatch):
    """BASELINE configs[4]: 20 000-atom images (1.6 M directed edges; one image needs more workspace than the default cap -- the engine
    then takes the full budget).  E and F against the float64 oracle (tools/make_golden_c5.py) on THREE cases: the BASELINE image, another
    cluster (`g1`), and the BASELINE image with another synthetic weight set (`w1`, seed 1).

    What bounds the energy of a float32-accumulating pipeline against exact arithmetic is not noise (sqrt N: 3e-5 eV here) but SYSTEMATIC
    terms -- errors coherent over the atoms or edges add up ~ N.  Rounds 3-4 removed the ones with a cause that could be found on ONE weight
    set (float32 copies of shared constants, `var + eps`, fp32-MFMA node linears, the one-sided adder of the 16-bit MFMAs -> sign-alternating
    operand rows) and reached +6.9e-5 eV at this size -- which round 5 showed to be a CANCELLATION specific to that weight set: with other
    weights the same build gave +1.4e-3 eV (tools/gpu_energy_bias.py, profiles/r05_energy_bias.txt).  tools/gpu_energy_cuts.py (linear
    response of the energy along cuts of the network, float64 oracle gradients) then located coherent terms of +-1...3e-8 eV per atom in
    every float32 RMS norm (cured: the norms, the edge -> node sums and the readout run in double since) and of +-1e-8 eV per atom and layer
    in every float32-ACCUMULATED GEMM stage, with TWO causes, both cured: (i) the bias was added to the finished float32 sum -- a value on
    the float32 grid plus a constant has ONE rounding error for all rows of a binade (tools/cpu_fp32_coherence.py reproduces the GPU's
    number on the CPU; the accumulators now start from the bias: fp32 mode -3.6e-8 ... +2.1e-8 -> <= 2e-9 eV per atom); (ii) the 16-bit
    matrix cores cut the 2^-16-order plane products against a large accumulator, with a part that follows the product's sign -- coherent
    where the operand columns are one-signed (tools/gpu_fc3_error_form.py; those products now accumulate apart: bf16x3 +5e-8 -> <= 8e-9 eV
    per atom); (iii) the same trap as (i) in the radial fc1 -- float32 sum + element-table constant -- whose rounding no longer reaches the
    LayerNorm (tools/gpu_fc1_table_form.py); (iv, round 6) stage 1 of a 16-bit MFMA pass cuts each of its 8 products TOWARD ZERO at 2^-24 of
    the largest before adding -- found with a bit-exact model of the adder fitted on raw hardware results (tools/mfma_emul.c,
    tests/test_mfma_model.py, tests/test_gpu_mfma_model.py) -- coherent where an activation column is one-signed and consistently small; the
    leading planes are now quantised to their pass group (UMX_ALIGN_PLANES): the weight set that kept -1.63e-4 eV (8e-9 eV per atom) is at
    +4e-7 eV.  The bound is the north-star's flat 1e-4 eV through 20 000 atoms (UMX_ENERGY_TOL_EV_N, include/umx.h; fast mode: 5e-8 eV per
    atom beyond the headline size) -- against 1.2e-7 eV per atom for a plain float32 evaluation in the reference's op style.  Forces keep the
    absolute 1e-3 eV/A at every size (measured 9e-7)."""
    from pdb2reaction_amd.engine import Engine

    g = load_golden(name)
    monkeypatch.setenv("UMX_PRECISION", mode)
    eng = Engine(0)
    try:
        eng.load_weights(W.make_synthetic_weights(int(g["weights_seed"]) if "weights_seed" in g else 0))
        eng.set_system(g["z"])
        assert eng.precision_mode() == {"auto": "bf16x3", "split": "split-f16", "fp32": "fp32"}[mode]
        e, f = eng.energy_forces(g["pos"][None])
        ne, maxdeg = eng.graph_stats()
        assert ne > 1_500_000 and maxdeg <= 300
        de = e[0] - g["energy"][0]
        df = np.abs(f[0].astype(np.float64) - g["forces"][0])
        print(f"[{name} {mode}] dE = {de:+.2e} eV ({de / 20000:+.1e} eV/atom), max|dF| = {df.max():.2e} eV/A, rms dF = {np.sqrt((df ** 2).mean()):.2e}")
        assert abs(de) <= energy_tol(20000, mode), (name, mode, de)
        if mode == "auto":
            assert abs(de) <= TOL_E, (name, mode, de)                    # flat 1e-4 eV through 20 000 atoms in the default mode
        assert df.max() <= TOL_F, (name, mode, df.max())
        assert not eng.widened
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["c5_n20000_w2", "c5_n20000_w3", "c5_n20000_w4", "c5_n20000_w5", "c5_n20000_w6", "c5_n20000_w7"])
def test_c5_energy_with_more_geometries_and_weight_sets(name):
    """Round 6: six 20 000-atom goldens made AFTER the aligned planes went in (other clusters, weights seeds 2 ... 7; tools/make_golden_c5.py;
    w4 ... w7 were added to this list before the engine had ever been run on them) -- the flat 1e-4 eV of the north-star must hold on cases the
    fix was never looked at on (round 4's claim at this size had been a cancellation on the one weight set it was measured with).  Default mode."""
    from pdb2reaction_amd.engine import Engine

    g = load_golden(name)
    eng = Engine(0)
    try:
        eng.load_weights(W.make_synthetic_weights(int(g["weights_seed"])))
        eng.set_system(g["z"])
        e, f = eng.energy_forces(g["pos"][None])
        de = e[0] - g["energy"][0]
        df = np.abs(f[0].astype(np.float64) - g["forces"][0]).max()
        print(f"[{name}] dE = {de:+.2e} eV ({de / 20000:+.1e} eV/atom), max|dF| = {df:.2e} eV/A")
        assert abs(de) <= TOL_E and df <= TOL_F, (name, de, df)
    finally:
        eng.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7])
def test_c3_energy_with_other_weight_sets(seed):
    """The headline size (2000 atoms) with seven OTHER synthetic weight sets (tools/make_golden_c3_weights.py; seeds 4-7 joined at the end of round 6,
    before the engine had run on them): the 1e-4 eV of the north-star must not depend on the one weight set the kernels were tuned on.  Default mode."""
    from pdb2reaction_amd.engine import Engine

    g = load_golden(f"c3_n2000_w{seed}")
    eng = Engine(0)
    try:
        eng.load_weights(W.make_synthetic_weights(seed))
        eng.set_system(g["z"])
        e, f = eng.energy_forces(g["pos"])
        de = e[0] - g["energy"][0]
        df = np.abs(f[0].astype(np.float64) - g["forces"][0]).max()
        print(f"[c3 weights seed {seed}] dE = {de:+.2e} eV ({de / 2000:+.1e} eV/atom), max|dF| = {df:.2e} eV/A")
        assert abs(de) <= TOL_E and df <= TOL_F, (seed, de, df)
    finally:
        eng.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7])
def test_c3_fast_mode_error_is_a_fixed_amount_per_atom_that_depends_on_the_weights(seed, monkeypatch):
    """The opt-in fast mode (split: fp16 forward planes, one accumulator) at the headline size with the seven other weight sets: its energy error is
    coherent -- -5.4e-8 ... +8.9e-8 eV per atom, fixed by the weight set (seed 6: +1.77e-4 eV at 2000 atoms, +1.78e-3 at 20 000) -- so it does NOT keep
    the north-star's 1e-4 eV on every weight set (seeds 2 and 6 are beyond it; profiles/r06_c3_weight_sets.txt).  Its own bound, include/umx.h:
    UMX_ENERGY_TOL_EV_FAST_N(n) = max(1e-4, 1.5e-7 n).  Forces stay 150x inside theirs."""
    from pdb2reaction_amd.engine import Engine

    g = load_golden(f"c3_n2000_w{seed}")
    monkeypatch.setenv("UMX_PRECISION", "split")
    eng = Engine(0)
    try:
        eng.load_weights(W.make_synthetic_weights(seed))
        eng.set_system(g["z"])
        e, f = eng.energy_forces(g["pos"])
        de = e[0] - g["energy"][0]
        df = np.abs(f[0].astype(np.float64) - g["forces"][0]).max()
        print(f"[c3 fast mode, weights seed {seed}] dE = {de:+.2e} eV ({de / 2000:+.1e} eV/atom), max|dF| = {df:.2e} eV/A")
        assert eng.precision_mode() == "split-f16" and not eng.widened
        assert abs(de) <= energy_tol(2000, "split") and df <= TOL_F, (seed, de, df)
    finally:
        eng.close()


@pytest.mark.parametrize("name", [f"c3_n2000_grid_w{i}" for i in range(7)] + [f"c5_n20000_grid_w{i}" for i in range(7)])
def test_grid_feed_forward_variant_at_the_baseline_sizes(name):
    """The GRID feed-forward form of the model (`ff_type = grid | spectral (unsure)` -- it may be the form the real checkpoint has) at
    the sizes where 1e-4 eV is 5e-8 / 5e-9 eV per atom: the variant tests stop at 500 atoms, and a coherent per-atom error in the node-level grid
    GEMMs would only show here.  Goldens: tools/make_golden_variant_sizes.py (float64 chunked oracle), made at the end of round 6 and asserted
    before the engine had run on them.  Default mode, the north-star's tolerances."""
    from pdb2reaction_amd.engine import Engine

    g = load_golden(name)
    w = W.make_synthetic_weights(int(g["weights_seed"]), ff_type="grid")
    eng = Engine(0)
    try:
        eng.load_weights(w)
        assert "ff=grid" in eng.model_variant()
        eng.set_system(g["z"])
        pos = g["pos"] if g["pos"].ndim == 3 else g["pos"][None]
        e, f = eng.energy_forces(pos)
        de = e[0] - g["energy"][0]
        df = np.abs(f[0].astype(np.float64) - g["forces"][0]).max()
        print(f"[{name}] dE = {de:+.2e} eV ({de / len(g['z']):+.1e} eV/atom), max|dF| = {df:.2e} eV/A")
        assert abs(de) <= TOL_E and df <= TOL_F, (name, de, df)
    finally:
        eng.close()


def test_c5_energy_with_the_atom_order_permuted(weights):
    """The same 20 000-atom image with its atoms in a random order: every edge gets another index, i.e. the parity that decides which operand
    rows are stored negated is re-dealt -- the cancellation of the matrix cores' one-sided rounding must not depend on the order the structure
    happens to be listed in.  Energy against the float64 golden within the north-star's 1e-4 eV, forces (un-permuted) within 1e-3 eV/A, and
    the two orders agree with each other to 1e-4 eV (measured 2.3e-5: the systematic part is order-independent)."""
    from pdb2reaction_amd.engine import Engine

    g = load_golden("c5_n20000")
    perm = np.random.default_rng(5).permutation(len(g["z"]))
    eng = Engine(0)
    try:
        eng.load_weights(weights)
        eng.set_system(g["z"])
        e0, _ = eng.energy_forces(g["pos"][None], forces=False)
        eng.set_system(g["z"][perm])
        e, f = eng.energy_forces(g["pos"][perm][None])
        de = e[0] - g["energy"][0]
        fb = np.empty_like(f[0])
        fb[perm] = f[0]
        df = np.abs(fb.astype(np.float64) - g["forces"][0]).max()
        print(f"[c5 permuted] dE = {de:+.2e} eV (listed order: {e0[0] - g['energy'][0]:+.2e} eV), max|dF| = {df:.2e} eV/A")
        assert abs(de) <= TOL_E and df <= TOL_F
        assert abs(e[0] - e0[0]) <= TOL_E
    finally:
        eng.close()
