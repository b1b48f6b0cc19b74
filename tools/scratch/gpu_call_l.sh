#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/l
cd /tmp && export TMPDIR=/tmp && cd $R
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/l/kt -o kt -f csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-mode > $R/gpurun_out/l/kt.log 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/l/kt/kt_kernel_stats.csv")))
for r in rows[:24]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):5d} {float(r['TotalDurationNs'])/3e6:8.2f} ms/it  avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
