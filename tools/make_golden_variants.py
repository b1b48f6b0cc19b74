"""tests/golden/*_grid*.npz: float64-oracle outputs (E eV, F eV/A) for the MODEL VARIANTS (grid feed-forward, pos_emb charge / spin
embedding, a checkpoint-ordered dataset list) -- same recipe as tools/make_golden.py: synthetic weights of the variant (seed 0), float32-
rounded positions, the repo's own CPU restatement ("parity unpinned").  The variant keywords are stored in the fixture (``variant_*``).

    python tools/make_golden_variants.py [small c2 | c3]
"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W, synth
from oracle.escn_md_oracle import Oracle
from oracle.chunked import ChunkedForces

torch.set_num_threads(8)


def run(name, variant, z, imgs, charge=0, spin=1, task="omol", chunked=False):
    w = W.make_synthetic_weights(0, **variant)
    orc = ChunkedForces(w) if chunked else Oracle(w)
    p32 = imgs.astype(np.float32)
    es, fs = [], []
    for k in range(len(p32)):
        t = time.time()
        e, f = orc.energy_forces(z, p32[k].astype(np.float64), charge=charge, spin=spin, task=task)
        es.append(e); fs.append(f)
        print(name, k, e, f"{time.time()-t:.1f}s", flush=True)
    extra = {f"variant_{k}": (np.array(list(v)) if isinstance(v, (tuple, list)) else np.array(v)) for k, v in variant.items()}
    np.savez_compressed(f"tests/golden/{name}.npz", z=z.astype(np.int32), pos=p32, energy=np.array(es), forces=np.stack(fs),
                        charge=charge, spin=spin, task=task, weights_seed=0, **extra)


which = sys.argv[1:] or ["small", "c2"]
if "c3" in which:            # the headline size with the grid feed-forward: one image of synth.make_images(2000, 16), ~4 min on 8 cores
    z, imgs, _ = synth.make_images(2000, 16)
    run("c3_n2000_k1_grid", dict(ff_type="grid"), z, imgs[[5]], chunked=True)
    sys.exit(0)
z, imgs, _ = synth.make_images(20, 3, seed=7)
run("small_n20_k3_grid", dict(ff_type="grid"), z, imgs)
run("small_n20_charged_grid_pos_emb", dict(ff_type="grid", chg_spin_emb_type="pos_emb", dataset_list=("omol", "omat", "oc20")), z, imgs[:1], charge=-1, spin=2, task="omat")
z, imgs, _ = synth.make_images(500, 12)
run("c2_n500_k1_grid", dict(ff_type="grid"), z, imgs[[6]], chunked=True)
