"""Energy error against the float64 goldens at c3 (4 x 2000 atoms) and c5 (20 000 atoms) for precision modes x dev switches.

    python3 tools/gpu_energy_bias.py [c3] [c5]
Each variant runs in a fresh engine with the given environment (the switches are read at umx_create / umx_load_weights)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W  # noqa: E402
from pdb2reaction_amd.engine import Engine  # noqa: E402

GOLD = os.path.join("tests", "golden")
VARIANTS = [
    {"UMX_PRECISION": "bf16x3"}, {"UMX_PRECISION": "bf16x3", "UMX_ALT_ROWS": "0"}, {"UMX_PRECISION": "split"}, {"UMX_PRECISION": "split", "UMX_ALT_ROWS": "0"},
    {"UMX_PRECISION": "fp32"}, {"UMX_PRECISION": "f16x2b8"}, {"UMX_PRECISION": "f16x2b8", "UMX_ALT_ROWS": "0"},
]
VARIANTS += [{"UMX_PRECISION": "split", "UMX_F16_PRODUCTS": "3"}]       # two-plane fp16 weights: what a rounding of the WEIGHTS does to the energy
if os.environ.get("BIAS_ONLY"):          # e.g. BIAS_ONLY=f16x2b8: only that mode's variants
    VARIANTS = [v for v in VARIANTS if v["UMX_PRECISION"] in os.environ["BIAS_ONLY"].split(",")]
which = sys.argv[1:] or ["c3", "c5"]
w = W.make_synthetic_weights(0)
for name in which:
    g = np.load(os.path.join(GOLD, "c5_n20000.npz" if name == "c5" else "c3c4_n2000.npz"))
    z = g["z"]
    pos = g["pos"][None] if name == "c5" else g["c3_pos"]
    e_ref = g["energy"] if name == "c5" else g["c3_energy"]
    f_ref = g["forces"][None] if name == "c5" else g["c3_forces"]
    for env in VARIANTS:
        for k in ("UMX_PRECISION", "UMX_NODE_F64", "UMX_DEG_SPLIT", "UMX_ALT_ROWS", "UMX_F16_PRODUCTS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        eng = Engine(0)
        eng.load_weights(w)
        eng.set_system(z)
        eng.reserve_images(len(pos))
        eng.energy_forces(pos)
        t = time.perf_counter()
        e, f = eng.energy_forces(pos)
        dt = time.perf_counter() - t
        de = e - e_ref
        df = np.abs(f.astype(np.float64) - f_ref).max()
        print(f"{name} {eng.precision_mode():10s} {str(env):90s} dE = {' '.join(f'{x:+.2e}' for x in de)} eV ({np.abs(de).max() / len(z):.1e} eV/atom)  "
              f"max|dF| = {df:.1e}  {dt * 1e3:.0f} ms", flush=True)
        eng.close()
