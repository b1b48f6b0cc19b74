import os, sys
import numpy as np
sys.path.insert(0, ".")
from pdb2reaction_amd import synth, weights as W
from pdb2reaction_amd.engine import Engine
w = W.make_synthetic_weights(0)
z, pos = synth.make_cluster(100, seed=4)
p32 = pos.astype(np.float32)
raw = {}
for a in ("1", "0"):
    os.environ["UMX_A_F32"] = a
    eng = Engine(0); eng.load_weights(w); eng.set_system(z); eng.debug_keep(True)
    eng.energy_forces(p32)
    raw[a] = eng.debug_fetch("a2raw.deg", np.uint8); ne = eng.graph_stats()[0]; eng.close()
ne4 = (ne + 3) // 4 * 4
f = raw["1"].view(np.float32).reshape(ne4 // 4, 128 // 16, 4, 16)          # [row group][block][row][k]
vals = f.transpose(0, 2, 1, 3).reshape(ne4, 128)
pl = raw["0"].view(np.uint16).reshape(ne4 // 4, 128 // 16, 4, 3, 16)      # [row group][block][row][plane][k]
planes = pl.transpose(0, 2, 3, 1, 4).reshape(ne4, 3, 128)
def bf(u16): return (u16.astype(np.uint32) << 16).view(np.float32)
s = bf(planes[:, 0]) .astype(np.float64) + bf(planes[:, 1]).astype(np.float64) + bf(planes[:, 2]).astype(np.float64)
print("edges", ne, "rows compared", ne)
d = np.abs(s[:ne] - vals[:ne].astype(np.float64))
print("max |sum of planes - float32 value|", d.max(), " nonzero:", np.count_nonzero(d))
# RN split of the float32 values on the host
x = vals[:ne].copy()
def rn_bf16(x):
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    return r
p0 = rn_bf16(x); r1 = x - bf(p0); p1 = rn_bf16(r1.astype(np.float32)); r2 = r1.astype(np.float32) - bf(p1); p2 = rn_bf16(r2.astype(np.float32))
for q, pq in enumerate((p0, p1, p2)):
    print("plane", q, "host RN vs device planes differing:", np.count_nonzero(pq != planes[:ne, q]))
idx = np.argwhere(p2 != planes[:ne, 2])[:8]
for i, k in idx:
    xv = x[i, k]
    print(f"x={xv!r:>14} bits={xv.view(np.uint32):08x} dev planes {planes[i,0,k]:04x} {planes[i,1,k]:04x} {planes[i,2,k]:04x} = {bf(planes[i,0,k:k+1])[0]!r} {bf(planes[i,1,k:k+1])[0]!r} {bf(planes[i,2,k:k+1])[0]!r}  host p2 {p2[i,k]:04x} = {bf(p2[i,k:k+1])[0]!r}   r2 = {r2[i,k]!r}")
