#!/bin/bash
# round 6: VERDICT r5 item 3a -- the persistent tile loop of the LS forward kernels (UMX_PERSIST=1), RUN: bitwise identity, then the c3 step
set -e
mkdir -p gpurun_out/r6p
python - <<'PY' > gpurun_out/r6p/bitwise.txt 2>&1
import os, sys, numpy as np
sys.path.insert(0, ".")
from pdb2reaction_amd import synth, weights as W
from pdb2reaction_amd.engine import Engine
out = {}
for n, k in ((2000, 3), (150, 2), (50, 8)):
    z, imgs, _ = synth.make_images(n, k)
    p32 = np.asarray(imgs, dtype=np.float32)
    for pers in ("0", "1"):
        os.environ["UMX_PERSIST"] = pers
        eng = Engine(0); eng.load_weights(W.make_synthetic_weights(0)); eng.set_system(z)
        out[(n, pers)] = eng.energy_forces(p32); eng.close()
    e0, f0 = out[(n, "0")]; e1, f1 = out[(n, "1")]
    print(f"N = {n} x {k}: UMX_PERSIST=1 bitwise equal to 0: energies {np.array_equal(e0, e1)}, forces {np.array_equal(f0, f1)}  (max |dE| {np.abs(e0 - e1).max():.1e}, max |dF| {np.abs(f0 - f1).max():.1e})")
PY
cat gpurun_out/r6p/bitwise.txt
for rep in 1 2; do
for pers in 0 1; do
  UMX_PERSIST=$pers python bench.py --no-shard --no-serial --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-fast-mode --driver string > gpurun_out/r6p/bench_p${pers}_$rep.json 2> gpurun_out/r6p/bench_p${pers}_$rep.err || true
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r6p/bench_p${pers}_$rep.json").read().strip().splitlines()[-1]); print("UMX_PERSIST=$pers ms_per_step", round(d["ms_per_step"],1), "gemm family ms", round(d["roofline"]["ms_per_step"],1))
except Exception as e: print("bench parse failed", e)
PY
done
done
UMX_PERSIST=1 UMX_PROFILE_DUMP=gpurun_out/r6p/gemm_p1.csv python bench.py --no-shard --no-serial --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-fast-mode --driver string > /dev/null 2>&1 || true
UMX_PERSIST=0 UMX_PROFILE_DUMP=gpurun_out/r6p/gemm_p0.csv python bench.py --no-shard --no-serial --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-fast-mode --driver string > /dev/null 2>&1 || true
ls gpurun_out/r6p
