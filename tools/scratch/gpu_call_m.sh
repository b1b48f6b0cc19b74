#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/m
for f in 0 2 1 0 2; do
  echo "== UMX_RADIAL_FAST=$f" &&
  UMX_RADIAL_FAST=$f timeout -k 10 300 python tools/gpu_c3_energy.py 2>&1 | grep -E "fp32  |split UMX_MFMA16=1" &&
  UMX_RADIAL_FAST=$f timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fp32-mode 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print(f\"   {d['ms_per_step']:.1f} ms/step, GEMM {r['ms_per_step']:.1f}, other-gemm {r['other_gemm_family']['ms_per_step']:.1f}, rest {r['hbm_regime']['ms_per_step']:.1f}\")"
done
