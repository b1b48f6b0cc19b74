"""GPU bring-up: compare every named intermediate of the HIP engine with the staged float64 oracle."""
import os
os.environ.setdefault("UMX_FUSE_MODROT", "0")   # expose g_xrot for the stage comparison
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W, synth
from pdb2reaction_amd.engine import Engine
from oracle.staged import Staged

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
w = W.make_synthetic_weights(0)
z, pos = synth.make_cluster(n)
pos32 = pos.astype(np.float32)
st = Staged(w)
em = st.forward(z, pos32.astype(np.float64)); g = st.backward()
rmsd = float(w["normalizer.rmsd"][0])
e_ref = float(em) * rmsd + float(np.asarray(w["element_refs"], np.float64)[z].sum())
f_ref = (-g * rmsd).numpy()
T = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in st.t.items()}

eng = Engine(0)
eng.load_weights(w)
eng.set_system(z)
eng.debug_keep(True)
e, f = eng.energy_forces(pos32)
print("E hip %.6f ref %.6f diff %.3e" % (e[0], e_ref, e[0] - e_ref))
print("F max|diff| %.3e  max|F| %.3e" % (np.abs(f[0] - f_ref).max(), np.abs(f_ref).max()))

ne = len(T["src"])
print("edges hip", eng.graph_stats(), "ref", ne)
src = eng.debug_fetch("src", np.int32); dst = eng.debug_fetch("dst", np.int32)
print("graph equal:", np.array_equal(src, T["src"]), np.array_equal(dst, T["dst"]))

def cmp(name, ref, shape=None):
    try:
        a = eng.debug_fetch(name)
    except Exception as ex:
        print(f"{name:12s} MISSING ({ex})"); return
    r = np.asarray(ref, dtype=np.float64).reshape(-1)
    if a.size != r.size:
        print(f"{name:12s} SIZE {a.size} vs {r.size}"); return
    d = np.abs(a - r).max(); s = np.abs(r).max()
    print(f"{name:12s} max|diff| {d:.3e}  max|ref| {s:.3e}  rel {d/(s+1e-30):.2e}")

fr = eng.debug_fetch("frame").reshape(ne, 36)
print("frame R   ", np.abs(fr[:, :9].reshape(ne, 3, 3) - T["rm"]).max())
wig = T["wig"]  # m-primary rows x l-primary cols
inv = np.argsort(np.array(W.TO_M))
wl = wig[:, inv, :]
print("frame D2  ", np.abs(fr[:, 9:34].reshape(ne, 5, 5) - wl[:, 4:9, 4:9]).max())
print("frame env ", np.abs(fr[:, 34] - T["env"]).max(), np.abs(fr[:, 35] - T["denv"]).max())
cmp("rad.deg", T["rad.deg"]); cmp("x0", T["x0"])
for i in range(4):
    cmp(f"xn.{i}", T[f"xn.{i}"]); cmp(f"xrot.{i}", T[f"xrot.{i}"]); cmp(f"rad.{i}", T[f"rad.{i}"])
    hg = np.concatenate([T[f"gate.{i}"], T[f"hpre.{i}"].reshape(ne, -1)], axis=1)
    cmp(f"hg.{i}", hg); cmp(f"hid.{i}", T[f"hid.{i}"]); cmp(f"msg.{i}", T[f"msg.{i}"]); cmp(f"xmid.{i}", T[f"xmid.{i}"])
    cmp(f"xn2.{i}", T[f"xn2.{i}"]); cmp(f"gspre.{i}", T[f"gspre.{i}"]); cmp(f"ffh.{i}", T[f"ffh.{i}"]); cmp(f"x.{i}", T[f"x.{i}"])
cmp("pre1", T["pre1"]); cmp("pre2", T["pre2"]); cmp("e_node", T["e_node"])
cmp("g_xfinal", T["g_xfinal"])
for i in reversed(range(4)):
    cmp(f"g_xmid.{i}", T[f"g_xmid.{i}"]); cmp(f"g_msg.{i}", T[f"g_msg.{i}"]); cmp(f"g_hid.{i}", T[f"g_hid.{i}"])
    ghg = np.concatenate([T[f"g_gate.{i}"], T[f"g_hpre.{i}"].reshape(ne, -1)], axis=1)
    cmp(f"g_hg.{i}", ghg); cmp(f"g_xrot.{i}", T[f"g_xrot.{i}"]); cmp(f"g_rad.{i}", T[f"g_rad.{i}"])
    cmp(f"g_xn.{i}", T[f"g_xn.{i}"]); cmp(f"g_xin.{i}", T[f"g_xin.{i}"])
cmp("dedd", T["dedd"])
tau = eng.debug_fetch("tau").reshape(ne, 4)[:, :3]
print("tau        max|diff| %.3e max|ref| %.3e  tau_y(gauge) %.3e" % (np.abs(tau - T["tau"]).max(), np.abs(T["tau"]).max(), np.abs(tau[:, 1]).max()))
gv = eng.debug_fetch("gvec").reshape(ne, 4)[:, :3]
print("gvec       max|diff| %.3e max|ref| %.3e" % (np.abs(gv - T["gvec"]).max(), np.abs(T["gvec"]).max()))
# batched: 3 images, chunked to 1 image/chunk vs all-in-one
imgs = np.stack([pos32, pos32 + 0.01, pos32[::-1].copy()])
eng.debug_keep(False)
e3, f3 = eng.energy_forces(imgs)
print("batch consistency: dE %.3e dF %.3e" % (abs(e3[0] - e[0]), np.abs(f3[0] - f[0]).max()), " translated dE %.3e" % (e3[1] - e3[0]))
t = time.time(); e3, f3 = eng.energy_forces(imgs); print("3 images time %.1f ms" % ((time.time() - t) * 1e3))
