#!/bin/bash
# round 6: aligned leading planes -- (1) the engine's fc3 GEMM against the bit-exact matrix-core model, (2) energy error at 20 000 atoms with the
# planes aligned / plain, (3) what it costs at c3
set -e
mkdir -p gpurun_out/r6a
python -m pytest tests/test_gpu_mfma_model.py -x -q -s > gpurun_out/r6a/model_test.log 2>&1 || { tail -30 gpurun_out/r6a/model_test.log; exit 1; }
tail -8 gpurun_out/r6a/model_test.log
BIAS_ENVS='[{"UMX_ALIGN_PLANES":"1"},{"UMX_ALIGN_PLANES":"0"}]' python tools/gpu_energy_bias.py w1 c5 g1 perm c3 > gpurun_out/r6a/energy_bias.txt 2>&1
cat gpurun_out/r6a/energy_bias.txt
for al in 1 0 1 0; do
  UMX_ALIGN_PLANES=$al python bench.py --no-shard --no-serial --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-fast-mode --driver string > gpurun_out/r6a/bench_al$al.json 2> gpurun_out/r6a/bench_al$al.err || true
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r6a/bench_al$al.json").read().strip().splitlines()[-1]); print("UMX_ALIGN_PLANES=$al ms_per_step", d["ms_per_step"])
except Exception as e: print("bench parse failed", e)
PY
done
