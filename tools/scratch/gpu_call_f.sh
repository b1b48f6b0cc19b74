#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/f
cd /tmp && export TMPDIR=/tmp
cd $R
for L in 1 2; do
  UMX_STREAMS=$L rocprofv3 --kernel-trace -d $R/gpurun_out/f/t$L -o t -f csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-mode > $R/gpurun_out/f/t$L.log 2>&1 || exit 1
  tail -1 $R/gpurun_out/f/t$L.log | cut -c1-200
done
ls -la $R/gpurun_out/f/t1 $R/gpurun_out/f/t2
