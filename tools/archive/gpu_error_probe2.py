"""Per-column systematic error: ratio of |mean over rows of (gpu - ref)| to the noise floor std/sqrt(rows)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W, synth
from pdb2reaction_amd.engine import Engine
from oracle.staged import Staged
torch.set_num_threads(16)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 250
w = W.make_synthetic_weights(0)
z, pos = synth.make_cluster(n); p32 = pos.astype(np.float32)
st = Staged(w); st.forward(z, p32.astype(np.float64))
T = {k: v.numpy() for k, v in st.t.items() if torch.is_tensor(v)}
ne = len(T["src"])
os.environ["UMX_PRECISION"] = "fp32"
eng = Engine(0); eng.load_weights(w); eng.set_system(z); eng.debug_keep(True)
eng.energy_forces(p32, forces=False)
def stat(name, ref, rows):
    a = eng.debug_fetch(name).astype(np.float64).reshape(rows, -1); r = np.asarray(ref, np.float64).reshape(rows, -1)
    d = a - r
    cm = d.mean(0); floor = d.std(0) / np.sqrt(rows) + 1e-30
    ratio = np.abs(cm) / floor
    print(f"{name:10s} cols {d.shape[1]:5d}  rms(colmean) {np.sqrt((cm**2).mean()):.3e}  noise floor {np.sqrt((floor**2).mean()):.3e}  median ratio {np.median(ratio):.2f}  max ratio {ratio.max():.1f}  overall std {d.std():.3e}")
h1 = None
stat("rad.deg", T["rad.deg"], ne); stat("x0", T["x0"], n)
for i in range(4):
    stat(f"xn.{i}", T[f"xn.{i}"], n); stat(f"xrot.{i}", T[f"xrot.{i}"], ne); stat(f"rad.{i}", T[f"rad.{i}"], ne)
    hg = np.concatenate([T[f"gate.{i}"], T[f"hpre.{i}"].reshape(ne, -1)], 1); stat(f"hg.{i}", hg, ne)
    stat(f"hid.{i}", T[f"hid.{i}"], ne); stat(f"msg.{i}", T[f"msg.{i}"], ne); stat(f"xmid.{i}", T[f"xmid.{i}"], n)
    stat(f"xn2.{i}", T[f"xn2.{i}"], n); stat(f"gspre.{i}", T[f"gspre.{i}"], n); stat(f"ffh.{i}", T[f"ffh.{i}"], n); stat(f"x.{i}", T[f"x.{i}"], n)
stat("pre1", T["pre1"], n); stat("pre2", T["pre2"], n); stat("e_node", T["e_node"], n)
