"""Energy / force error vs the float64 oracle fixtures (c3: 2000 atoms, c5: 20000 atoms) for each forward operand format of the
split path: three bf16 planes (6 products) and two fp16 planes (4 or 3 products)."""
import os, sys, time, numpy as np
sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W
from pdb2reaction_amd.engine import Engine
g3 = np.load("tests/golden/c3c4_n2000.npz")
g5 = np.load("tests/golden/c5_n20000.npz")
print("c5 keys", list(g5.keys()))
w = W.make_synthetic_weights(0)
for fwd, prod in (("bf16", "4"), ("f16", "4"), ("f16", "3")):
    os.environ["UMX_PRECISION"] = "split"; os.environ["UMX_FWD"] = fwd; os.environ["UMX_F16_PRODUCTS"] = prod
    eng = Engine(0); eng.load_weights(w); eng.set_system(g3["z"])
    e, f = eng.energy_forces(g3["c3_pos"])
    t = time.time(); e, f = eng.energy_forces(g3["c3_pos"]); dt = time.time() - t
    df = np.abs(f.astype(np.float64) - g3["c3_forces"])
    print(f"{fwd} prod={prod}: c3 dE {e - g3['c3_energy']} eV  max|dF| {df.max():.2e} rms {np.sqrt((df**2).mean()):.2e} eV/A  ({dt*1e3:.1f} ms)", flush=True)
    eng.close()
    eng = Engine(0); eng.load_weights(w); eng.set_system(g5["z"])
    e, f = eng.energy_forces(g5["pos"][None] if g5["pos"].ndim == 2 else g5["pos"])
    df = np.abs(f.astype(np.float64) - g5["forces"])
    print(f"{fwd} prod={prod}: c5 dE {e - g5['energy']} eV  max|dF| {df.max():.2e} rms {np.sqrt((df**2).mean()):.2e} eV/A", flush=True)
    eng.close()
