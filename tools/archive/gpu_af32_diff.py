"""Dev: which stage differs between UMX_A_F32=1 (float32 operand blocks) and UMX_A_F32=0 (pre-split planes)?"""
import os, sys
import numpy as np
sys.path.insert(0, ".")
from pdb2reaction_amd import synth, weights as W
from pdb2reaction_amd.engine import Engine
w = W.make_synthetic_weights(0)
z, pos = synth.make_cluster(int(sys.argv[1]) if len(sys.argv) > 1 else 300, seed=4)
p32 = pos.astype(np.float32)
names = ["rad.deg", "x0"] + [f"{s}.{i}" for i in range(4) for s in ("xn", "rad", "hg", "msg", "x")] + ["e_node", "g_xfinal"] + [f"{s}.{i}" for i in (3, 2, 1, 0) for s in ("g_xmid", "g_hid", "g_xn", "g_xin")] + ["dedd"]
outs = []
for a in ("1", "0"):
    os.environ["UMX_A_F32"] = a
    eng = Engine(0); eng.load_weights(w); eng.set_system(z); eng.debug_keep(True)
    e, f = eng.energy_forces(p32)
    d = {}
    for nm in names:
        try:
            d[nm] = eng.debug_fetch(nm)
        except Exception:
            pass
    outs.append((e, f, d)); eng.close()
print("mode energies", outs[0][0], outs[1][0], "forces equal", np.array_equal(outs[0][1], outs[1][1]))
for nm in names:
    if nm in outs[0][2] and nm in outs[1][2]:
        a, b = outs[0][2][nm], outs[1][2][nm]
        nd = int(np.count_nonzero(a != b))
        print(f"{nm:10s} size {a.size:9d} differing {nd:9d}  max|d| {np.abs(a - b).max():.3e}" + ("   <-- first difference" if nd else ""))
        if nd:
            idx = np.flatnonzero(a != b)[:5]
            print("    at", idx, a[idx], b[idx])
            pass
