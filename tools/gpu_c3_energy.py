import os, sys, numpy as np
sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W
from pdb2reaction_amd.engine import Engine
g = np.load("tests/golden/c3_n2000_energy.npz")
w = W.make_synthetic_weights(0)
for mode in ("fp32", "split"):
    os.environ["UMX_PRECISION"] = mode
    eng = Engine(0); eng.load_weights(w); eng.set_system(g["z"])
    e, _ = eng.energy_forces(g["pos"], forces=False)
    print(mode, "c3 N=2000 dE vs f64 oracle:", e - g["energy"], "per atom", (e - g["energy"]) / 2000 / 1.5)
    eng.close()
