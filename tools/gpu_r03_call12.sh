#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c12
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "engine_matches_oracle or stage_by_stage or golden or precision_modes or documented" > gpurun_out/c12/parity.log 2>&1; rc=$?
tail -3 gpurun_out/c12/parity.log
[ $rc -eq 0 ] || exit $rc
bash tools/gpu_env_ab.sh UMX_RADIAL_TR=2 UMX_RADIAL_TR=1
