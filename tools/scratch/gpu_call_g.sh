#!/bin/bash
# GPU call G: grid-stride kernels + throttled two-lane mode: correctness, then a sweep of the block cap
set -o pipefail
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/g
echo "== lanes/cap bitwise" && timeout -k 10 300 python - <<'PY' &&
import os, numpy as np
from pdb2reaction_amd import synth, weights as W
from pdb2reaction_amd.engine import Engine
w = W.make_synthetic_weights(0)
z, imgs, _ = synth.make_images(300, 5)
res = {}
for lanes, cap in (("1", "512"), ("2", "512"), ("2", "64"), ("2", "0")):
    os.environ["UMX_STREAMS"] = lanes; os.environ["UMX_STREAM_BLOCKS"] = cap
    eng = Engine(0); eng.load_weights(w); eng.set_system(z)
    res[(lanes, cap)] = eng.energy_forces(imgs)
    eng.close()
ref = res[("1", "512")]
for k, v in res.items():
    ok = np.array_equal(ref[0], v[0]) and np.array_equal(ref[1], v[1])
    print(k, "bitwise equal:", ok); assert ok
PY
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "engine_matches or precision_modes or stage_by_stage or c3_energy or batch" 2>&1 | tail -2 &&
for cfg in "UMX_STREAMS=1" "UMX_STREAMS=2 UMX_STREAM_BLOCKS=0" "UMX_STREAMS=2 UMX_STREAM_BLOCKS=256" "UMX_STREAMS=2 UMX_STREAM_BLOCKS=512" "UMX_STREAMS=2 UMX_STREAM_BLOCKS=1024" "UMX_STREAMS=2 UMX_STREAM_BLOCKS=2048" "UMX_STREAMS=2 UMX_STREAM_BLOCKS=512 UMX_MAX_CHUNK_IMAGES=4"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  echo "== bench $cfg" && env $cfg timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode > $R/gpurun_out/g/bench_$tag.log 2>&1 &&
  python - "$tag" <<'PY'
import json,sys
tag=sys.argv[1]
d=json.loads([l for l in open(f"gpurun_out/g/bench_{tag}.log") if l.startswith("{")][-1])
r=d["roofline"]
print(f"   {tag}: {d['ms_per_step']:.1f} ms/step, GEMM {r['ms_per_step']:.1f} ms, other-gemm {r['other_gemm_family']['ms_per_step']:.1f}, rest {r['hbm_regime']['ms_per_step']:.1f}")
PY
done
