"""Energy / force error of the GRID feed-forward variant at 2000 and 20 000 atoms against tests/golden/c{3_n2000,5_n20000}_grid_w<seed>.npz, per mode.

    python3 tools/gpu_grid_variant_sizes.py [golden names ...]"""
import glob
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W  # noqa: E402
from pdb2reaction_amd.engine import Engine  # noqa: E402

names = sys.argv[1:] or sorted(os.path.basename(p)[:-4] for p in glob.glob("tests/golden/c*_grid_w*.npz"))
for name in names:
    g = np.load(os.path.join("tests", "golden", name + ".npz"))
    w = W.make_synthetic_weights(int(g["weights_seed"]), ff_type="grid")
    pos = g["pos"] if g["pos"].ndim == 3 else g["pos"][None]
    line = [f"{name}:"]
    for mode, extra in (("bf16x3", {}), ("bf16x3", {"UMX_GRID_F64": "0"}), ("fp32", {}), ("split", {})):
        os.environ["UMX_PRECISION"] = mode
        os.environ.pop("UMX_GRID_F64", None)
        os.environ.update(extra)
        eng = Engine(0)
        eng.load_weights(w)
        eng.set_system(g["z"])
        e, f = eng.energy_forces(pos)
        de = e[0] - g["energy"][0]
        df = np.abs(f[0].astype(np.float64) - g["forces"][0]).max()
        line.append(f"{eng.precision_mode()}{' GRID_F64=0' if extra else ''} dE = {de:+.2e} ({de / len(g['z']):+.1e} /atom) dF {df:.1e}")
        eng.close()
    os.environ.pop("UMX_GRID_F64", None)
    print("   ".join(line), flush=True)
