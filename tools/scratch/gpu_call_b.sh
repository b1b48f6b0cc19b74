#!/bin/bash
# GPU call B: two-lane token executor -- correctness (bitwise vs one lane) and bench variants
set -o pipefail
mkdir -p gpurun_out/b
echo "== lanes bitwise" && timeout -k 10 300 python - <<'PY' &&
import os, numpy as np
from pdb2reaction_amd import synth, weights as W
from pdb2reaction_amd.engine import Engine
w = W.make_synthetic_weights(0)
z, imgs, _ = synth.make_images(300, 5)
res = {}
for lanes in ("1", "2"):
    os.environ["UMX_STREAMS"] = lanes
    eng = Engine(0); eng.load_weights(w); eng.set_system(z)
    res[lanes] = eng.energy_forces(imgs)
    eng.close()
print("bitwise equal:", np.array_equal(res["1"][0], res["2"][0]) and np.array_equal(res["1"][1], res["2"][1]))
assert np.array_equal(res["1"][0], res["2"][0]) and np.array_equal(res["1"][1], res["2"][1])
PY
for cfg in "UMX_STREAMS=1" "UMX_STREAMS=2" "UMX_STREAMS=2 UMX_Q3WIDE=0 UMX_WIDE=0" "UMX_STREAMS=2 UMX_Q3WIDE=0" "UMX_STREAMS=2 UMX_WIDE=0" "UMX_STREAMS=2 UMX_MAX_CHUNK_IMAGES=4" "UMX_STREAMS=2 UMX_MAX_CHUNK_IMAGES=2"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  echo "== bench $cfg" && env $cfg timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode > gpurun_out/b/bench_$tag.log 2>&1 &&
  python - "$tag" <<'PY'
import json,sys
tag=sys.argv[1]
d=json.loads([l for l in open(f"gpurun_out/b/bench_{tag}.log") if l.startswith("{")][-1])
r=d["roofline"]
print(f"   {tag}: {d['ms_per_step']:.1f} ms/step, GEMM {r['ms_per_step']:.1f} ms, other-gemm {r['other_gemm_family']['ms_per_step']:.1f}, rest {r['hbm_regime']['ms_per_step']:.1f}")
PY
done
