"""c3 energy error of the engine vs the float64 oracle fixture, per precision mode and MFMA-shape choice."""
import os, sys, numpy as np
sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W
from pdb2reaction_amd.engine import Engine
g = np.load("tests/golden/c3_n2000_energy.npz")
w = W.make_synthetic_weights(0)
for mode, m16 in (("fp32", "1"), ("split", "0"), ("split", "1"), ("split", "2")):
    os.environ["UMX_PRECISION"] = mode
    os.environ["UMX_MFMA16"] = m16
    eng = Engine(0); eng.load_weights(w); eng.set_system(g["z"])
    e, _ = eng.energy_forces(g["pos"], forces=False)
    print(f"{mode:5s} UMX_MFMA16={m16}: c3 N=2000 dE vs f64 oracle: {e - g['energy']} eV; per atom {(e - g['energy']) / 2000}")
    eng.close()
