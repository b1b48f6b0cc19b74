"""Soak run: growing-string optimisation of a 500-atom synthetic cluster on the engine (c2-like), watching device memory
and wall time per cycle.  usage: python tools/gpu_soak.py [atoms] [max_nodes] [cycles] [lanczos]
"lanczos" (round 6): the climbing thresholds are forced so that the reference's default climbing phase (climb_lanczos, path_opt.py:179-182) runs
from the first fully grown cycle on -- the warm-started Lanczos recursions over hundreds of cycles."""
import importlib
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from pdb2reaction_amd import synth  # noqa: E402
from pdb2reaction_amd.gsm import GrowingStringDriver  # noqa: E402

U = importlib.import_module("pdb2reaction_amd.uma_pysis")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
nodes = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 25
z, imgs, frozen = synth.make_images(n, 2)
elem = [synth.SYMBOLS[int(a)] for a in z]
calc = U.uma_pysis(model="synthetic", freeze_atoms=list(frozen))
r, p = (imgs[0] * U.ANG2BOHR).reshape(-1), (imgs[1] * U.ANG2BOHR).reshape(-1)
free0 = torch.cuda.mem_get_info()[0]
t0 = time.perf_counter()
force_lz = len(sys.argv) > 4 and sys.argv[4] == "lanczos"
gs_kw = {"max_nodes": nodes, "climb": True, **({"climb_rms": 1e9, "climb_lanczos_rms": 1e9} if force_lz else {})}
drv = GrowingStringDriver.from_calculator(elem, r, p, calc, gs_kw=gs_kw, stopt_kw={"max_cycles": cycles})
print("driver device:", drv.device)
res = drv.run()
dt = time.perf_counter() - t0
free1 = torch.cuda.mem_get_info()[0]
print(f"atoms {n} images {len(res.coords)} cycles {res.cycles} force evaluations {res.force_evaluations} "
      f"fully_grown {res.fully_grown} converged {res.converged}")
print(f"wall {dt:.2f} s = {dt / max(res.cycles, 1) * 1e3:.1f} ms/cycle; image E+F per s {res.force_evaluations / dt:.1f}")
print(f"timing: total {res.timing['total_s']:.2f} s, optimistic steps recomputed {int(res.timing['redo_steps'])}, Lanczos evaluations {drv.lanczos_evals} in {drv.lanczos_calls} recursions ({drv.lanczos_warm_calls} warm-started and kept, {drv.lanczos_warm_rejected} warm results rejected)")
print(f"device memory in use by the run: {(free0 - free1) / 2**30:.2f} GiB (workspace is allocated once and kept)")
print("energies (Hartree, rel. to first):", np.round(res.energies - res.energies[0], 5))
assert np.isfinite(res.energies).all() and np.isfinite(res.coords).all()
# second run on the same calculator must not grow the footprint
drv2 = GrowingStringDriver.from_calculator(elem, r, p, calc, gs_kw={"max_nodes": nodes, "climb": False}, stopt_kw={"max_cycles": 5})
drv2.run()
free2 = torch.cuda.mem_get_info()[0]
print(f"after a second run: {(free1 - free2) / 2**20:.1f} MiB more")
assert free1 - free2 < 64 * 2**20
