#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c13
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/c13/gpu_tests.log 2>&1; rc=$?; tail -4 gpurun_out/c13/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python3 tools/gpu_fuzz_parity.py > gpurun_out/c13/fuzz.log 2>&1; rc=$?; grep -v amdgpu.ids gpurun_out/c13/fuzz.log | tail -6; exit $rc
