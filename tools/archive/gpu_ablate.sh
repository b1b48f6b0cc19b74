#!/bin/bash
# dev: radial-head ablation builds (build/libumx_abl<mask>.so), c3 bench each, prints the fused-radial family time
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abl
for m in 0 1 2 4 8 16 31 0; do
  if [ $m = 0 ]; then unset UMX_LIBRARY UMX_ALLOW_STALE; else export UMX_LIBRARY=$GRAFT_REPO_ROOT/build/libumx_abl$m.so UMX_ALLOW_STALE=1; fi
  timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode > gpurun_out/abl/b$m.log 2>&1 || { tail -3 gpurun_out/abl/b$m.log; continue; }
  python3 - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/abl/b$m.log") if l.startswith("{")][-1]
r=d["roofline"]
print("ABL=%-2s ms_per_step %.2f  radial(head+tail) %.2f  edge %.2f" % ("$m", d["ms_per_step"], r["hbm_regime"]["radial"]["ms_per_step"], r["hbm_regime"]["ms_per_step"]))
PY
done
