"""Repeat the body of test_two_lane_execution_is_bitwise_identical in one process and report any mismatch (dev)."""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from pdb2reaction_amd import synth, weights as W  # noqa: E402
from pdb2reaction_amd.engine import Engine  # noqa: E402

w = W.make_synthetic_weights(0)
z, imgs, _ = synth.make_images(260, 5, seed=21)
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    res = {}
    for lanes, cap in (("1", "512"), ("2", "512"), ("2", "64"), ("2", "0")):
        os.environ["UMX_STREAMS"], os.environ["UMX_STREAM_BLOCKS"] = lanes, cap
        eng = Engine(0)
        eng.load_weights(w)
        eng.set_system(z)
        res[(lanes, cap)] = eng.energy_forces(imgs)
        eng.close()
    e0, f0 = res[("1", "512")]
    for key, (e, f) in res.items():
        if not (np.array_equal(e, e0) and np.array_equal(f, f0)):
            bad += 1
            df = np.abs(f - f0)
            print(f"rep {rep} {key}: dE = {e - e0}, max|dF| = {np.nanmax(df):.3e} in image(s) {sorted(set(np.argwhere(df > 0)[:, 0].tolist()))}, "
                  f"atoms differing {int((df.max(-1) > 0).sum())}, nan {int(np.isnan(f).sum())}", flush=True)
print("mismatches:", bad)
