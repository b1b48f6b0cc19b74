"""Energy error against the float64 goldens at c3 (4 x 2000 atoms) and c5 (20 000 atoms: the BASELINE image, another cluster `g1`, another
weight set `w1`, and the BASELINE image with its atoms permuted `perm`) for precision modes x switches.

    python3 tools/gpu_energy_bias.py [c3] [c5] [g1] [w1] [perm]
Each variant runs in a fresh engine with the given environment (the switches are read at umx_create / umx_load_weights).  UMX_LIBRARY may point
at a library built with other flags (e.g. -DUMX_EXP_DOUBLE: every exp / sigmoid / SiLU correctly rounded from double) to measure what those
functions contribute."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W  # noqa: E402
from pdb2reaction_amd.engine import Engine  # noqa: E402

GOLD = os.path.join("tests", "golden")
VARIANTS = [{"UMX_PRECISION": "bf16x3"}, {"UMX_PRECISION": "bf16x3", "UMX_ALT_ROWS": "0"}, {"UMX_PRECISION": "split"}, {"UMX_PRECISION": "split", "UMX_ALT_ROWS": "0"},
            {"UMX_PRECISION": "fp32"}, {"UMX_PRECISION": "bf16x3", "UMX_NODE_F64": "0"}]
if os.environ.get("BIAS_LOW_SEP"):      # the bf16x3 forward products with their 2^-16-order plane products chained from zero: none / fc3 / all
    VARIANTS = [{"UMX_PRECISION": "bf16x3", "UMX_LOW_SEP": v} for v in os.environ["BIAS_LOW_SEP"].split(",")]
if os.environ.get("BIAS_ONLY"):          # e.g. BIAS_ONLY=bf16x3: only that mode's variants
    VARIANTS = [v for v in VARIANTS if v["UMX_PRECISION"] in os.environ["BIAS_ONLY"].split(",")]
if os.environ.get("BIAS_ENVS"):          # any list of switch settings: BIAS_ENVS='[{"UMX_PRECISION": "bf16x3", "UMX_ALIGN_PLANES": "0"}, ...]'
    import json
    VARIANTS = [{"UMX_PRECISION": "bf16x3", **{str(k): str(v) for k, v in e.items()}} for e in json.loads(os.environ["BIAS_ENVS"])]
SWITCHES = sorted({k for v in VARIANTS for k in v} | {"UMX_PRECISION", "UMX_NODE_F64", "UMX_ALT_ROWS", "UMX_LOW_SEP", "UMX_ALIGN_PLANES"})
FILES = {"c3": "c3c4_n2000.npz", "c5": "c5_n20000.npz", "g1": "c5_n20000_g1.npz", "w1": "c5_n20000_w1.npz", "perm": "c5_n20000.npz",
         "w2": "c5_n20000_w2.npz", "w3": "c5_n20000_w3.npz",         # (round 6: two more geometries with weights seeds 2 / 3)
         "w4": "c5_n20000_w4.npz", "w5": "c5_n20000_w5.npz", "w6": "c5_n20000_w6.npz", "w7": "c5_n20000_w7.npz"}          # (end of round 6: two more, cluster seeds 20260630 / 20260730, weights seeds 4 / 5)
which = sys.argv[1:] or ["c3", "c5"]
for name in which:
    g = np.load(os.path.join(GOLD, FILES[name]))
    w = W.make_synthetic_weights(int(g["weights_seed"]) if "weights_seed" in g.files else 0)
    z = g["z"]
    pos = g["c3_pos"] if name == "c3" else g["pos"][None]
    e_ref = g["c3_energy"] if name == "c3" else g["energy"]
    f_ref = g["c3_forces"] if name == "c3" else g["forces"][None]
    if name == "perm":
        perm = np.random.default_rng(5).permutation(len(z))
        z, pos, f_ref = z[perm], pos[:, perm], f_ref.reshape(1, len(z), 3)[:, perm]
    for env in VARIANTS:
        for k in SWITCHES:
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
        print(f"{name} {eng.precision_mode():10s} {str(env):70s} dE = {' '.join(f'{x:+.2e}' for x in de)} eV ({np.abs(de).max() / len(z):.1e} eV/atom)  "
              f"max|dF| = {df:.1e}  {dt * 1e3:.0f} ms", flush=True)
        eng.close()
