"""c3 energy AND force error of the engine vs the float64 oracle fixture, per precision mode and MFMA-shape choice."""
import os, sys, numpy as np
sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W
from pdb2reaction_amd.engine import Engine
g = np.load("tests/golden/c3c4_n2000.npz")
w = W.make_synthetic_weights(0)
for mode, m16 in (("fp32", "1"), ("split", "0"), ("split", "1"), ("split", "2")):
    os.environ["UMX_PRECISION"] = mode
    os.environ["UMX_MFMA16"] = m16
    eng = Engine(0); eng.load_weights(w); eng.set_system(g["z"])
    e, f = eng.energy_forces(g["c3_pos"])
    df = np.abs(f.astype(np.float64) - g["c3_forces"])
    print(f"{mode:5s} UMX_MFMA16={m16}: c3 N=2000 dE vs f64 oracle: {e - g['c3_energy']} eV; max|dF| {df.max():.2e} rms {np.sqrt((df**2).mean()):.2e} eV/A", flush=True)
    eng.close()
