"""A fuzz case (seed 47, weights seed 2 / lin_emb, 117 atoms) left the force tolerance by 1.7e-3 eV/A: four of its edges lie within 1e-5 of the
+y pole, where the reference DETACHES the frame angles (torch.isclose(x_y, 1): |x_y - 1| <= 1e-8 + 1e-5) -- and one of them sits within float32
rounding of that threshold, so a float32 evaluation of x_y (the engine's, and the reference's own) and the float64 oracle's can decide differently.
This script evaluates the oracle with the threshold moved slightly to either side and shows which decision the engine took.

    python tools/gpu_pole_threshold_case.py        (MI355X)"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W, synth  # noqa: E402
from pdb2reaction_amd.engine import Engine  # noqa: E402
import oracle.escn_md_oracle as O  # noqa: E402

rng = np.random.default_rng(47)
VARIANTS = [(0, {}), (3, {}), (0, dict(ff_type="grid")), (1, dict(ff_type="grid", chg_spin_emb_type="pos_emb", dataset_list=("omol", "omat", "oc20", "odac"))),
            (2, dict(chg_spin_emb_type="lin_emb"))]
found = None
for wseed, vkw in VARIANTS:                      # replay tools/gpu_fuzz_parity.py 47 60 up to the case
    tasks = list(vkw.get("dataset_list", W.DATASET_LIST))
    for case in range(12):
        n = int(rng.integers(2, 161))
        z, pos = synth.make_cluster(n, seed=int(rng.integers(1, 10**6)))
        z = rng.choice(np.array([1, 5, 6, 7, 8, 9, 11, 12, 15, 16, 17, 20, 26, 29, 30, 35, 53], dtype=np.int32), size=n)
        pos = pos * rng.uniform(0.7, 1.6)
        charge, spin, task = int(rng.integers(-2, 3)), int(rng.integers(0 if vkw.get("chg_spin_emb_type") else 1, 4)), tasks[int(rng.integers(0, len(tasks)))]
        if wseed == 2 and n == 117:
            found = (wseed, vkw, z, pos.astype(np.float32), charge, spin, task)
wseed, vkw, z, p32, charge, spin, task = found
w = W.make_synthetic_weights(wseed, **vkw)
eng = Engine(0); eng.load_weights(w); eng.set_system(z, charge=charge, spin=spin, task=task)
e, f = eng.energy_forces(p32[None])
d = p32[:, None, :].astype(np.float64) - p32[None, :, :]
r = np.linalg.norm(d, axis=-1) + np.eye(len(z)) * 100
ny = (d[..., 1] / r)[r <= 6.0]
ny32 = ((p32[:, None, 1] - p32[None, :, 1]) / np.sqrt(((p32[:, None, :] - p32[None, :, :]) ** 2).sum(-1) + np.eye(len(z), dtype=np.float32) * 100).astype(np.float32))[r <= 6.0]
near = np.argsort(-ny)[:4]
print("edges nearest the +y pole: 1 - x_y (float64) =", [f"{1 - ny[i]:.4e}" for i in near], " (float32) =", [f"{np.float32(1) - ny32[i]:.4e}" for i in near], " threshold 1e-8 + 1e-5 = 1.001e-5")
real_isclose = torch.isclose
for rtol in (1e-5, 0.99e-5, 1.01e-5):
    O.torch.isclose = lambda a, b, _r=rtol, **kw: real_isclose(a, b, rtol=_r, atol=1e-8)
    orc = O.Oracle(w)
    e_ref, f_ref = orc.energy_forces(z, p32.astype(np.float64), charge=charge, spin=spin, task=task)
    print(f"oracle pole threshold rtol = {rtol:.2e}: |dE| = {abs(e[0] - e_ref):.2e} eV, max|dF| = {np.abs(f[0] - f_ref).max():.2e} eV/A")
O.torch.isclose = real_isclose
