"""Generate tests/golden/*.npz: inputs and float64-oracle outputs (E eV, F eV/A) for parity tests.

The reference (fairchem UMA behind pdb2reaction/uma_pysis.py) cannot be imported here and ships no
fixtures, so these vectors come from the repo's own CPU restatement -- "parity
unpinned" -- with deterministic synthetic weights (seed 0).  Positions are the float32-rounded
values the engine receives.  Usage: python tools/make_golden.py [c1] [c2] [small]
"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W, synth
from oracle.escn_md_oracle import Oracle

torch.set_num_threads(8)
w = W.make_synthetic_weights(0)
orc = Oracle(w)
which = sys.argv[1:] or ["small", "c1", "c2"]

def run(name, z, imgs, charge=0, spin=1, task="omol"):
    p32 = imgs.astype(np.float32)
    es, fs = [], []
    for k in range(len(p32)):
        t = time.time()
        e, f = orc.energy_forces(z, p32[k].astype(np.float64), charge=charge, spin=spin, task=task)
        es.append(e); fs.append(f)
        print(name, k, e, f"{time.time()-t:.1f}s", flush=True)
    np.savez_compressed(f"tests/golden/{name}.npz", z=z.astype(np.int32), pos=p32, energy=np.array(es), forces=np.stack(fs),
                        charge=charge, spin=spin, task=task, weights_seed=0)

if "small" in which:
    z, imgs, _ = synth.make_images(20, 3, seed=7)
    run("small_n20_k3", z, imgs)
    run("small_n20_charged", z, imgs[:1], charge=-1, spin=2, task="omat")
if "c1" in which:
    z, imgs, _ = synth.make_images(50, 8)
    run("c1_n50_k8", z, imgs)
if "c2" in which:
    z, imgs, _ = synth.make_images(500, 12)
    run("c2_n500_k2", z, imgs[[0, 6]])
