"""Two independent PROCESSES evaluating on the same GPU at the same time: is each one's result still bitwise reproducible?  (dev)
Separates 'a kernel is disturbed by whatever else runs on the chip' from 'the two-lane executor shares something it should not'."""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, ".")
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    from pdb2reaction_amd import synth, weights as W
    from pdb2reaction_amd.engine import Engine

    seed, reps = int(sys.argv[2]), int(sys.argv[3])
    z, imgs, _ = synth.make_images(260, 3, seed=seed)
    eng = Engine(0)
    eng.load_weights(W.make_synthetic_weights(0))
    eng.set_system(z)
    names = ["g_xfinal"] + [f"{n}.{i}" for i in (3, 2, 1, 0) for n in ("g_xmid", "g_hid", "g_xn", "g_xin")] + ["dedd", "tau", "gvec"]
    trace = os.environ.get("CORUN_TRACE") == "1"
    if trace:
        eng.debug_keep(True)
    e0, f0 = eng.energy_forces(imgs)
    ref = {n: eng.debug_fetch(n).copy() for n in names} if trace else {}
    bad = 0
    for r in range(reps):
        e, f = eng.energy_forces(imgs)
        if not (np.array_equal(e, e0) and np.array_equal(f, f0)):
            bad += 1
            if trace:
                diff = [n for n in names if not np.array_equal(eng.debug_fetch(n), ref[n])]
                print(f"worker {seed} rep {r}: first differing captures: {diff[:4]}", flush=True)
            print(f"worker {seed} rep {r}: dE {np.abs(e - e0).max():.1e} max|dF| {np.abs(f - f0).max():.2e} images {sorted(set(np.argwhere(f != f0)[:, 0].tolist()))}", flush=True)
    print(f"worker {seed}: {bad} of {reps} evaluations differ from the first", flush=True)
    sys.exit(0)
reps = sys.argv[1] if len(sys.argv) > 1 else "300"
ps = [subprocess.Popen([sys.executable, __file__, "worker", str(21 + i), reps]) for i in range(2)]
sys.exit(max(p.wait() for p in ps))
