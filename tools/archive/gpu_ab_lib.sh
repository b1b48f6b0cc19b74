#!/bin/bash
# A/B of two builds of the library on ONE box: default libumx.so against $1 (a path under build/), c3 bench, alternating runs
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab
mkdir -p $O
cd $R
ALT=$1
for v in A B A B; do
  if [ $v = A ]; then unset UMX_LIBRARY UMX_ALLOW_STALE; else export UMX_LIBRARY=$R/$ALT UMX_ALLOW_STALE=1; fi
  timeout -k 10 300 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fp32-mode > $O/bench_$v.log 2>&1 || { tail -5 $O/bench_$v.log; exit 1; }
  python3 - <<PY
import json
d=[json.loads(l) for l in open("$O/bench_$v.log") if l.startswith("{")][-1]
r=d["roofline"]
print("$v (${UMX_LIBRARY:-default}) ms_per_step %.2f  split-gemm %.2f  node-gemm %.2f  radial %.2f  edge %.2f" % (d["ms_per_step"], r["ms_per_step"], r["other_gemm_family"]["ms_per_step"], r["hbm_regime"]["radial"]["ms_per_step"], r["hbm_regime"]["ms_per_step"]))
PY
done
