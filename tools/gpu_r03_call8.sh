#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c8
mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?
tail -4 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python3 bench.py > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }
grep '^{' $O/bench.log | tail -1 | cut -c1-1500
