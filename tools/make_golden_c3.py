"""f64 oracle ENERGY of c3 images (2000 atoms): forward only under no_grad (forces at this size need ~200 GB of autograd state)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W, synth
from oracle.escn_md_oracle import Oracle
torch.set_num_threads(8)
w = W.make_synthetic_weights(0)
orc = Oracle(w)
z, imgs, _ = synth.make_images(2000, 16)
ks = [0, 9]
p32 = imgs[ks].astype(np.float32)
es = []
for i, k in enumerate(ks):
    t = time.time()
    with torch.no_grad():
        e, _ = orc.energy_forces(z, p32[i].astype(np.float64), forces=False)
    es.append(e); print(k, repr(e), f"{time.time()-t:.0f}s", flush=True)
np.savez_compressed("tests/golden/c3_n2000_energy.npz", z=z.astype(np.int32), pos=p32, energy=np.array(es), image_index=np.array(ks),
                    charge=0, spin=1, task="omol", weights_seed=0)
