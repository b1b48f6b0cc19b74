#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c9
timeout -k 10 900 python3 tools/gpu_energy_bias.py c3 c5 > gpurun_out/c9/bias.log 2>&1 || { tail gpurun_out/c9/bias.log; exit 1; }
grep -v amdgpu.ids gpurun_out/c9/bias.log | grep "NODE_F64': '1'}" | cut -c1-200
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/c9/parity.log 2>&1; rc=$?; tail -3 gpurun_out/c9/parity.log; exit $rc
