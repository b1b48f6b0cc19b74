#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c11
timeout -k 10 900 python3 tools/gpu_energy_bias.py c3 c5 > gpurun_out/c11/bias.log 2>&1 || { tail gpurun_out/c11/bias.log; exit 1; }
grep -v amdgpu.ids gpurun_out/c11/bias.log | grep "NODE_F64': '1'}" | cut -c1-200
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/c11/gpu_tests.log 2>&1; rc=$?; tail -4 gpurun_out/c11/gpu_tests.log; exit $rc
