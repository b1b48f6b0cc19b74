#!/bin/bash
# round 3, call 2: the GPU test suite on the ABI v7 build (range flag, auto precision, graph-parallel vs oracle, 1-rank RCCL)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c2
mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q -s > $O/gpu_tests.log 2>&1; rc=$?
tail -25 $O/gpu_tests.log
exit $rc
