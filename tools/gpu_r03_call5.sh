#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c5
mkdir -p $O
cd $R
echo "=== UMX_PRECISION=fp32 UMX_NODE_F64=1" > $O/stage.log
UMX_PRECISION=fp32 UMX_NODE_F64=1 timeout -k 10 600 python3 tools/gpu_stage_bias.py 400 >> $O/stage.log 2>&1 || exit 1
echo "=== UMX_PRECISION=split UMX_NODE_F64=1" >> $O/stage.log
UMX_PRECISION=split UMX_NODE_F64=1 timeout -k 10 600 python3 tools/gpu_stage_bias.py 400 >> $O/stage.log 2>&1 || exit 1
timeout -k 10 900 python3 tools/gpu_energy_bias.py c3 c5 > $O/bias.log 2>&1 || exit 1
grep -v amdgpu.ids $O/bias.log | cut -c1-200
