#!/bin/bash
# quick parity gate + A/B against build/libumx_prev.so
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/q
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "engine_matches_oracle or stage_by_stage or golden or precision_modes" > gpurun_out/q/parity.log 2>&1; rc=$?
tail -3 gpurun_out/q/parity.log
[ $rc -eq 0 ] || exit $rc
bash tools/gpu_ab_lib.sh ${1:-build/libumx_prev.so}
