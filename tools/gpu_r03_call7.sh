#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c7
mkdir -p $O
cd $R
for sd in 1 0 1 0; do
  UMX_SIDE=$sd timeout -k 10 300 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fp32-mode > $O/bench_side$sd.log 2>&1 || { tail -5 $O/bench_side$sd.log; exit 1; }
  python3 - <<PY
import json
d=[json.loads(l) for l in open("$O/bench_side$sd.log") if l.startswith("{")][-1]
r=d["roofline"]
print("UMX_SIDE=$sd ms_per_step %.2f  split-gemm %.2f  fp32-gemm %.2f  radial %.2f  rest-minus-radial %.2f" % (d["ms_per_step"], r["ms_per_step"], r["other_gemm_family"]["ms_per_step"], r["hbm_regime"]["radial"]["ms_per_step"], r["hbm_regime"]["ms_per_step"]))
PY
done
true; rc=0

exit $rc
