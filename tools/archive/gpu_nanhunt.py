import os, sys
import numpy as np
sys.path.insert(0, ".")
os.environ["UMX_NO_WIDEN"] = "1"
from pdb2reaction_amd import synth, weights as W
from pdb2reaction_amd.engine import Engine, UmxError
w = W.make_synthetic_weights(0)
z, pos = synth.make_cluster(26, seed=4)
eng = Engine(0); eng.load_weights(w); eng.set_system(z); eng.debug_keep(True)
print("mode", eng.precision_mode())
try:
    eng.energy_forces(pos.astype(np.float32))
    print("ok")
except UmxError as e:
    print("ERR", e)
names = ["x0", "rad.deg", "e_node"]
for i in range(4):
    names += [f"{s}.{i}" for s in ("xn", "rad", "hg", "msg", "xmid", "xn2", "gspre", "ffh", "x")]
for nm in names:
    try:
        a = eng.debug_fetch(nm)
        print(f"{nm:10s} size {a.size:8d} nonfinite {np.count_nonzero(~np.isfinite(a)):8d} absmax {np.nanmax(np.abs(a[np.isfinite(a)])) if np.isfinite(a).any() else float('nan'):.3e}")
    except Exception as ex:
        print(nm, "fetch failed", ex)
