#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/h
for cfg in "UMX_Q3S=2" "UMX_Q3S=3" "UMX_Q3S=3 UMX_MFMA16=2" "UMX_Q3S=2 UMX_MFMA16=2" "UMX_Q3S=2 UMX_MFMA16=0" "UMX_Q3S=2"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  echo "== bench $cfg" && env $cfg timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fp32-mode > $R/gpurun_out/h/bench_$tag.log 2>&1 &&
  python - "$tag" <<'PY'
import json,sys
tag=sys.argv[1]
d=json.loads([l for l in open(f"gpurun_out/h/bench_{tag}.log") if l.startswith("{")][-1])
r=d["roofline"]
print(f"   {tag}: {d['ms_per_step']:.1f} ms/step, GEMM {r['ms_per_step']:.1f} ms, other-gemm {r['other_gemm_family']['ms_per_step']:.1f}, rest {r['hbm_regime']['ms_per_step']:.1f}")
PY
done
