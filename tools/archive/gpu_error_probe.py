"""Localise energy-error sources: per-stage signed error statistics (GPU fp32 vs f64 staged oracle)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W, synth
from pdb2reaction_amd.engine import Engine
from oracle.staged import Staged

torch.set_num_threads(16)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 250
w = W.make_synthetic_weights(0)
z, pos = synth.make_cluster(n)
p32 = pos.astype(np.float32)
st = Staged(w); st.forward(z, p32.astype(np.float64))
T = {k: v.numpy() for k, v in st.t.items() if torch.is_tensor(v)}
for mode in ("fp32", "split"):
    os.environ["UMX_PRECISION"] = mode
    eng = Engine(0); eng.load_weights(w); eng.set_system(z); eng.debug_keep(True)
    e, _ = eng.energy_forces(p32, forces=False)
    print(f"== {mode}: N={n}")
    def stat(name, ref):
        a = eng.debug_fetch(name).astype(np.float64); r = np.asarray(ref, np.float64).reshape(-1)
        d = a - r
        print(f"{name:10s} mean {d.mean():+.3e}  std {d.std():.3e}  max {np.abs(d).max():.3e}  |ref|rms {np.sqrt((r**2).mean()):.3e}   sum(d) {d.sum():+.3e}")
    stat("frame", np.concatenate([T["rm"].reshape(len(T["src"]), 9), np.zeros((len(T["src"]), 27))], 1)) if False else None
    fr = eng.debug_fetch("frame").reshape(-1, 36).astype(np.float64)
    for nm, col, ref in (("env", 34, T["env"]), ("denv", 35, T["denv"])):
        d = fr[:, col] - ref; print(f"{nm:10s} mean {d.mean():+.3e} std {d.std():.3e} max {np.abs(d).max():.3e}")
    d = fr[:, :9] - T["rm"].reshape(-1, 9); print(f"R          mean {d.mean():+.3e} std {d.std():.3e} max {np.abs(d).max():.3e}")
    ev = eng.debug_fetch("evec").reshape(-1, 4).astype(np.float64)
    d = ev[:, 3] - T["dist"]; print(f"dist       mean {d.mean():+.3e} std {d.std():.3e} max {np.abs(d).max():.3e}")
    stat("rad.deg", T["rad.deg"]); stat("x0", T["x0"])
    for i in range(4):
        stat(f"xn.{i}", T[f"xn.{i}"]); stat(f"rad.{i}", T[f"rad.{i}"]); stat(f"msg.{i}", T[f"msg.{i}"]); stat(f"xmid.{i}", T[f"xmid.{i}"]); stat(f"x.{i}", T[f"x.{i}"])
    stat("pre1", T["pre1"]); stat("pre2", T["pre2"]); stat("e_node", T["e_node"])
    eng.close()
