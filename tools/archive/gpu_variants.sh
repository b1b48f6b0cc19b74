#!/bin/bash
# dev: c3 bench of the default build and of each library given on the command line (paths under the repo), twice, alternating
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/var
for rep in 1 2; do
for lib in default "$@"; do
  if [ $lib = default ]; then unset UMX_LIBRARY UMX_ALLOW_STALE; else export UMX_LIBRARY=$GRAFT_REPO_ROOT/$lib UMX_ALLOW_STALE=1; fi
  timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode > gpurun_out/var/b.log 2>&1 || { tail -3 gpurun_out/var/b.log; continue; }
  python3 - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/var/b.log") if l.startswith("{")][-1]
r=d["roofline"]
print("%-28s ms_per_step %.2f  split-gemm %.2f  node-gemm %.2f  radial %.2f  edge %.2f" % ("$lib", d["ms_per_step"], r["ms_per_step"], r["other_gemm_family"]["ms_per_step"], r["hbm_regime"]["radial"]["ms_per_step"], r["hbm_regime"]["ms_per_step"]))
PY
done; done
