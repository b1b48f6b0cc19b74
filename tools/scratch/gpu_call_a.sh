#!/bin/bash
# GPU call A (round 2): full GPU test suite on the new boundary/tests, bench line, stream/tile variants, force-error probe.
set -o pipefail
mkdir -p gpurun_out/a
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
echo "== pytest -m gpu" && timeout -k 10 900 python -m pytest tests -m gpu -x -q -s 2>&1 | tee gpurun_out/a/pytest.log | tail -15 &&
echo "== bench default" && timeout -k 10 400 python bench.py --steps 3 --warmup 1 > gpurun_out/a/bench_default.log 2>&1 && tail -1 gpurun_out/a/bench_default.log | cut -c1-1500 &&
for cfg in "UMX_STREAMS=2" "UMX_Q3WIDE=0 UMX_WIDE=0" "UMX_STREAMS=2 UMX_Q3WIDE=0 UMX_WIDE=0" "UMX_STREAMS=2 UMX_Q3WIDE=0 UMX_WIDE=0 UMX_MAX_CHUNK_IMAGES=4" "UMX_MAX_CHUNK_IMAGES=16"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  echo "== bench $cfg" && env $cfg timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode > gpurun_out/a/bench_$tag.log 2>&1 &&
  python - "$tag" <<'PY'
import json,sys
tag=sys.argv[1]
d=json.loads([l for l in open(f"gpurun_out/a/bench_{tag}.log") if l.startswith("{")][-1])
r=d["roofline"]
print(f"   {tag}: {d['ms_per_step']:.1f} ms/step, GEMM {r['ms_per_step']:.1f} ms, other-gemm {r['other_gemm_family']['ms_per_step']:.1f}, rest {r['hbm_regime']['ms_per_step']:.1f}")
PY
done &&
echo "== c3 error probe" && timeout -k 10 300 python tools/gpu_c3_energy.py 2>&1 | tee gpurun_out/a/c3_err.log
