#!/bin/bash
# round 3, call 1: baseline of the round-2 build at the sizes GSM really runs at (c1, c2, the 2-image shard) + kernel traces
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
python3 tools/gpu_eval_config.py c1 50 > $O/c1.log 2>&1 && cat $O/c1.log &&
python3 tools/gpu_eval_config.py c2 20 > $O/c2.log 2>&1 && cat $O/c2.log &&
python3 tools/gpu_eval_config.py c3-shard 10 > $O/c3s.log 2>&1 && cat $O/c3s.log &&
rocprofv3 --kernel-trace --stats -d $O/kt_c2 -o kt -f csv -- python3 tools/gpu_eval_config.py c2 5 > $O/kt_c2.log 2>&1 &&
rocprofv3 --kernel-trace --stats -d $O/kt_c1 -o kt -f csv -- python3 tools/gpu_eval_config.py c1 20 > $O/kt_c1.log 2>&1 &&
head -30 $O/kt_c2/kt_kernel_stats.csv && echo done
