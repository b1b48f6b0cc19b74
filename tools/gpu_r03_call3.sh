#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c3
mkdir -p $O
cd $R
timeout -k 10 900 python3 tools/gpu_energy_bias.py c3 c5 > $O/bias.log 2>&1; rc=$?
grep -v amdgpu.ids $O/bias.log | tail -30
exit $rc
