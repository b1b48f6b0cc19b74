"""Energy error at the headline size (one 2000-atom image) for the weight sets tests/golden/c3_n2000_w<seed>.npz, in every precision mode.

    python3 tools/gpu_c3_weight_sets.py [seeds ...]        (default 1 ... 7)"""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W  # noqa: E402
from pdb2reaction_amd.engine import Engine  # noqa: E402

seeds = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 5, 6, 7]
for seed in seeds:
    g = np.load(os.path.join("tests", "golden", f"c3_n2000_w{seed}.npz"))
    w = W.make_synthetic_weights(seed)
    line = [f"weights seed {seed}:"]
    for mode in ("bf16x3", "fp32", "split"):
        os.environ["UMX_PRECISION"] = mode
        eng = Engine(0)
        eng.load_weights(w)
        eng.set_system(g["z"])
        e, f = eng.energy_forces(g["pos"])
        de = e[0] - g["energy"][0]
        df = np.abs(f[0].astype(np.float64) - g["forces"][0]).max()
        line.append(f"{eng.precision_mode()} dE = {de:+.2e} eV ({de / 2000:+.1e} /atom) max|dF| = {df:.1e}")
        eng.close()
    print("   ".join(line), flush=True)
