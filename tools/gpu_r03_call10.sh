#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c10
export UMX_LIBRARY=$GRAFT_REPO_ROOT/build/libumx_f1.so UMX_ALLOW_STALE=1
for f in 0 1 2 0 1 2; do
UMX_RADIAL_FAST=$f timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode > gpurun_out/c10/b.log 2>&1 || { tail -3 gpurun_out/c10/b.log; exit 1; }
python3 - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/c10/b.log") if l.startswith("{")][-1]
print("FAST=$f ms_per_step %.2f radial %.2f" % (d["ms_per_step"], d["roofline"]["hbm_regime"]["radial"]["ms_per_step"]))
PY
done
