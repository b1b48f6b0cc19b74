"""Randomised parity sweep of the default precision mode against the float64 oracle: sizes 2-160, elements up to Z = 53, compressed and
dilute clusters (down to 0.75 A contacts), charge / spin / task variants, several weight seeds -- and (round 5) the model variants: the
grid feed-forward, pos_emb / lin_emb charge-spin embeddings, a dataset list in another order.  Prints the worst errors; exits 1 when a
case leaves the north-star tolerances (1e-4 eV, 1e-3 eV/A) or when the engine had to widen its operands.

    python3 tools/gpu_fuzz_parity.py [seed] [cases]"""
import sys, numpy as np
sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W, synth
from pdb2reaction_amd.engine import Engine
from oracle.escn_md_oracle import Oracle

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 36
worst_e = worst_f = 0.0
bad = 0
VARIANTS = [(0, {}), (3, {}), (0, dict(ff_type="grid")), (1, dict(ff_type="grid", chg_spin_emb_type="pos_emb", dataset_list=("omol", "omat", "oc20", "odac"))),
            (2, dict(chg_spin_emb_type="lin_emb"))]
for wseed, vkw in VARIANTS:
    w = W.make_synthetic_weights(wseed, **vkw)
    orc = Oracle(w)
    eng = Engine(0); eng.load_weights(w)
    tasks = list(eng.dataset_list)
    print(f"# weights seed {wseed}, variant {eng.model_variant()}", flush=True)
    for case in range(max(ncase // len(VARIANTS), 1)):
        n = int(rng.integers(2, 161))
        z, pos = synth.make_cluster(n, seed=int(rng.integers(1, 10**6)))
        z = rng.choice(np.array([1, 5, 6, 7, 8, 9, 11, 12, 15, 16, 17, 20, 26, 29, 30, 35, 53], dtype=np.int32), size=n)
        pos = pos * rng.uniform(0.7, 1.6)                       # compressed ... dilute (isolated atoms, ragged graphs)
        charge, spin, task = int(rng.integers(-2, 3)), int(rng.integers(0 if vkw.get("chg_spin_emb_type") else 1, 4)), tasks[int(rng.integers(0, len(tasks)))]
        p32 = pos.astype(np.float32)
        eng.set_system(z, charge=charge, spin=spin, task=task)
        e, f = eng.energy_forces(p32[None])
        e_ref, f_ref = orc.energy_forces(z, p32.astype(np.float64), charge=charge, spin=spin, task=task)
        de, df = abs(e[0] - e_ref), np.abs(f[0] - f_ref).max()
        dmin = np.sort(np.linalg.norm(p32[:, None] - p32[None], axis=-1) + 10 * np.eye(n), axis=None)[0]
        flag = "" if (de <= 1e-4 and df <= 1e-3) else "   <-- OUT OF TOLERANCE"
        if flag:
            # The reference detaches the edge frame where x_y is "numerically 1" (torch.isclose: |x_y - 1| <= 1e-8 + 1e-5), and detached / not detached
            # differ by ~1e-3 eV/A on such an edge.  An edge within float32 rounding of that threshold can fall on either side depending on how x_y
            # was rounded (the engine's float32, the reference's float32, this oracle's float64): not an arithmetic error.  Evaluate the oracle with
            # the threshold nudged to either side and accept the case if one of them is the engine's decision (tools/gpu_pole_threshold_case.py).
            import oracle.escn_md_oracle as OM
            real = OM.torch.isclose
            try:
                for rtol in (0.98e-5, 1.02e-5):
                    OM.torch.isclose = lambda a, b, _r=rtol, **kw: real(a, b, rtol=_r, atol=1e-8)
                    e2, f2 = Oracle(w).energy_forces(z, p32.astype(np.float64), charge=charge, spin=spin, task=task)
                    if abs(e[0] - e2) <= 1e-4 and np.abs(f[0] - f2).max() <= 1e-3:
                        flag = f"   (pole-threshold ambiguity: an edge within float32 rounding of |x_y - 1| = 1.001e-5; oracle threshold {rtol:.2e}: max|dF| = {np.abs(f[0] - f2).max():.2e})"
                        de, df = abs(e[0] - e2), np.abs(f[0] - f2).max()
                        break
            finally:
                OM.torch.isclose = real
            if "ambiguity" in flag:
                bad -= 1
        bad += 1 if flag else 0
        worst_e, worst_f = max(worst_e, de), max(worst_f, df)
        print(f"w{wseed} N={n:3d} q={charge:+d} s={spin} {task:5s} dmin={dmin:.2f} A  edges={eng.graph_stats()[0]:6d}  |dE|={de:.2e} eV  max|dF|={df:.2e} eV/A  max|F|={np.abs(f_ref).max():.2f}{flag}", flush=True)
    if eng.widened:
        print("engine widened its operands (fp16 range exceeded)"); bad += 1
    eng.close()
print(f"worst |dE| {worst_e:.2e} eV, worst max|dF| {worst_f:.2e} eV/A over {ncase} cases; failures: {bad}")
sys.exit(1 if bad else 0)
