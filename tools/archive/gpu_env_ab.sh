#!/bin/bash
# dev: c3 bench with environment variants given as "VAR=val" words (one run each, twice, alternating with the default)
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/env
for rep in 1 2; do
for kv in default "$@"; do
  ( [ "$kv" = default ] || export "$kv"
  timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode > gpurun_out/env/b.log 2>&1 || { tail -3 gpurun_out/env/b.log; exit 0; }
  python3 - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/env/b.log") if l.startswith("{")][-1]
r=d["roofline"]
print("%-28s ms_per_step %.2f  split-gemm %.2f  node-gemm %.2f  radial %.2f  edge %.2f" % ("$kv", d["ms_per_step"], r["ms_per_step"], r["other_gemm_family"]["ms_per_step"], r["hbm_regime"]["radial"]["ms_per_step"], r["hbm_regime"]["ms_per_step"]))
PY
  )
done; done
