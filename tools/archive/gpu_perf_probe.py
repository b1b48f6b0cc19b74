"""Quick timing probe: N atoms x K images, wall time per batched E+F call and GEMM share."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from pdb2reaction_amd import weights as W, synth
from pdb2reaction_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ks = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
w = W.make_synthetic_weights(0)
eng = Engine(0)
eng.load_weights(w)
for k in ks:
    z, imgs, frozen = synth.make_images(n, k)
    eng.set_system(z)
    eng.reserve_images(k)
    p = imgs.astype(np.float32)
    e, f = eng.energy_forces(p)          # warm-up + allocation
    ne, md = eng.graph_stats()
    t = time.time(); reps = 3
    for _ in range(reps):
        e, f = eng.energy_forces(p)
    dt = (time.time() - t) / reps
    eng.profile_enable(True); eng.profile_read(True)
    e, f = eng.energy_forces(p)
    pr = eng.profile_read(True); eng.profile_enable(False)
    flops = 30.98e6 * ne
    print(f"N={n} K={k} edges={ne} maxdeg={md}: {dt*1e3:.1f} ms/call  {k*n/dt:.3e} atom-img/s  "
          f"alg {flops/dt/1e12:.1f} TFLOP/s | gemm {pr['gemm_ms']:.1f} ms over {pr['gemm_launches']} launches, "
          f"{pr['gemm_flops']/max(pr['gemm_ms'],1e-9)/1e9:.1f} TFLOP/s in-kernel | sumF {np.abs(f.sum(1)).max():.2e}", flush=True)
    t = time.time(); e2, _ = eng.energy_forces(p, forces=False); print(f"   energy-only {1e3*(time.time()-t):.1f} ms")
